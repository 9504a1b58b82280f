// mas.hip -- Monotonic Alignment Search on the GPU (gfx950), replacing the reference's only native code:
//   Grad-TTS/model/monotonic_align/core.pyx:9-35 (maximum_path_each), :40-45 (batch loop),
//   wrapper __init__.py:8-23 (value * mask, t_x/t_y from the mask).
//
// One workgroup per batch item.  The DP runs column by column (y); inside a column every row x of the band
// [max(0, t_x+y-t_y), min(t_x, y+1)) is independent, so rows are spread over the lanes and the previous column
// lives in LDS (double-buffered).  Each cell does exactly the reference's arithmetic -- one fp32 max and one fp32
// add -- so DP values, and therefore the path, are bit-identical to the CPU result.  The backtrack only ever
// compares value[index][y-1] < value[index-1][y-1]; that predicate is recorded per cell during the forward
// sweep (1 byte in caller-provided scratch), so the DP matrix itself is never written back and `value` stays
// read-only.
#include <cstdlib>
#include "common.h"
#include <atomic>
#include "kernels.h"

namespace gtts {

// t_x > t_y (more tokens than frames; nothing the alignment of tts.py:116-127 produces): the band of core.pyx:18 is empty in EVERY
// column (lo = t_x - t_y + y > y = hi - 1), the reference's in-place `value` stays the raw value * mask, and its backtrack (:32-35) walks
// those.  One thread restates that walk (t_y dependent loads; the read at y == 0 never reaches `path`).
__device__ static void mas_degenerate(const float *__restrict__ val, const float *__restrict__ msk, int *__restrict__ path, size_t base,
                                      int t_x, int t_y, int ty) {
    int index = t_x - 1;
    for (int y = t_y - 1; y >= 0; --y) {
        path[base + (size_t)index * ty + y] = 1;
        if (index != 0) {
            bool dec = index == y;
            if (!dec && y > 0) {
                const size_t o1 = (size_t)index * ty + (y - 1), o0 = (size_t)(index - 1) * ty + (y - 1);
                const float a = msk ? __fmul_rn(val[o1], msk[o1]) : val[o1];
                const float c = msk ? __fmul_rn(val[o0], msk[o0]) : val[o0];
                dec = a < c;
            }
            if (dec) index -= 1;
        }
    }
}

__global__ __launch_bounds__(256) void mas_kernel(const float *__restrict__ value, const float *__restrict__ mask,
                                                  const int *__restrict__ t_xs, const int *__restrict__ t_ys,
                                                  int *__restrict__ path, unsigned char *__restrict__ flags, int tx,
                                                  int ty) {
    extern __shared__ float sm[];          // [2][tx]: DP values of the previous / current column
    const int b = blockIdx.x, tid = threadIdx.x;
    const int t_x = t_xs[b], t_y = t_ys[b];
    if (t_x <= 0 || t_y <= 0 || t_x > tx || t_y > ty) return;   // reference behaviour undefined: all-zero path
    const float NEG = -1e9f;
    const size_t base = (size_t)b * tx * ty;
    const float *val = value + base;
    const float *msk = mask ? mask + base : nullptr;
    unsigned char *flg = flags + base;
    float *prev = sm, *cur = sm + tx;
    if (t_x > t_y) {
        if (tid == 0) mas_degenerate(val, msk, path, base, t_x, t_y, ty);
        return;
    }

    for (int y = 0; y < t_y; ++y) {
        const int lo = max(0, t_x + y - t_y), hi = min(t_x, y + 1);
        for (int x = lo + tid; x < hi; x += 256) {
            const float pc = (y > 0 && x != y) ? prev[x] : NEG;                 // value[x][y-1]   (core.pyx:18-21)
            float pp;
            if (x == 0) pp = (y == 0) ? 0.f : NEG;                              // core.pyx:22-26
            else pp = prev[x - 1];                                              // value[x-1][y-1] (core.pyx:28)
            const size_t o = (size_t)x * ty + y;
            float v = val[o];
            if (msk) v = __fmul_rn(v, msk[o]);                                  // value * mask (__init__.py:13)
            const float mx = (pc > pp) ? pc : pp;
            cur[x] = __fadd_rn(mx, v);                                          // core.pyx:30
            flg[o] = (x > 0 && x != y && y > 0 && prev[x] < prev[x - 1]) ? 1 : 0;   // backtrack predicate (:34)
        }
        __syncthreads();
        float *t = prev; prev = cur; cur = t;
    }
    __threadfence_block();
    __syncthreads();
    if (tid == 0) {
        int index = t_x - 1;
        for (int y = t_y - 1; y >= 0; --y) {                                    // core.pyx:32-35
            path[base + (size_t)index * ty + y] = 1;
            if (index != 0 && (index == y || (y > 0 && flg[(size_t)index * ty + y]))) index -= 1;
        }
    }
}

// ---- round 4: one DP wave per sample, columns staged through LDS ------------------------------------------------------------
// The kernel above pays one workgroup barrier and one round trip of stride-t_y (uncoalesced) global loads per column, and its
// backtrack is 1 k dependent global loads by one thread: 0.8 ms for 16 x 200 x 1024.  Here:
//   * waves 1-3 stage tiles of TY columns (coalesced 128-byte row segments, value * mask applied) transposed into LDS, one tile ahead;
//   * wave 0 owns the whole column: lane l holds rows l*R .. l*R+R-1 of the previous column in registers (row x-1 of its first row
//     comes from lane l-1 by one shuffle), so a DP step is R LDS reads + R max/add, no barrier (one barrier per TILE);
//   * the backtrack predicates of a column are R bits per lane: one coalesced 128-byte store per column ([b][t_y][64] u16 in the
//     caller's scratch), read back through LDS in chunks for the backtrack, which walks them with LDS-latency dependent reads.
// Same arithmetic per cell as the reference (one fp32 max, one fp32 add, strict '<'): bit-identical paths.
template <int R, int TY>
__global__ __launch_bounds__(256) void mas_wave_kernel(const float *__restrict__ value, const float *__restrict__ mask,
                                                       const int *__restrict__ t_xs, const int *__restrict__ t_ys,
                                                       int *__restrict__ path, unsigned short *__restrict__ flags, int tx, int ty) {
    constexpr int TXP = 64 * R + 4;                    // row pitch of a staged column (a multiple of 4 floats: a lane's R rows are 16-byte reads)
    constexpr int CH = 256;                            // columns per backtrack chunk (CH * 64 u16 = 32 KB, inside the tile buffers)
    static_assert(2 * TY * TXP * 4 >= CH * 64 * 2, "the backtrack chunk reuses the tile buffers");
    extern __shared__ float sm[];                      // [2][TY][TXP]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int t_x = t_xs[b], t_y = t_ys[b];
    if (t_x <= 0 || t_y <= 0 || t_x > tx || t_y > ty) return;   // reference behaviour undefined: all-zero path
    const float NEG = -1e9f;
    const size_t base = (size_t)b * tx * ty;
    const float *val = value + base;
    const float *msk = mask ? mask + base : nullptr;
    unsigned short *flg = flags + (size_t)b * ty * 64;
    const int ntile = (t_y + TY - 1) / TY;
    if (t_x > t_y) {
        if (tid == 0) mas_degenerate(val, msk, path, base, t_x, t_y, ty);
        return;
    }

    // stage tile t into buffer t & 1: lane -> (row parity group, column), TY columns per row segment
    auto stage = [&](int t, int first_wave, int nwaves) {
        float *dst = sm + (size_t)(t & 1) * TY * TXP;
        const int y0 = t * TY;
        constexpr int RPI = 64 / TY;                   // rows per wave instruction
        constexpr int U = 8;                           // row groups in flight per lane (a tile is latency-, not bandwidth-bound)
        const int col = lane % TY, sub = lane / TY;
        const int y = y0 + col;
        const bool yok = y < t_y;
        const int step = nwaves * RPI;
        for (int xb = (wave - first_wave) * RPI + sub; xb < t_x; xb += step * U) {
            float v[U], m[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int x = min(xb + u * step, t_x - 1);                      // (clamped: the extra loads are not stored)
                const size_t o = (size_t)x * ty + (yok ? y : 0);
                v[u] = val[o];
                m[u] = msk ? msk[o] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int x = xb + u * step;
                if (x < t_x) dst[col * TXP + x] = yok ? (msk ? __fmul_rn(v[u], m[u]) : v[u]) : 0.f;     // value * mask (__init__.py:13)
            }
        }
    };
    stage(0, 0, 4);
    __syncthreads();
    float prevv[R];
#pragma unroll
    for (int r = 0; r < R; ++r) prevv[r] = NEG;
    const int x0 = lane * R;
    for (int t = 0; t < ntile; ++t) {
        if (wave != 0) {
            if (t + 1 < ntile) stage(t + 1, 1, 3);
        } else {
            const float *src = sm + (size_t)(t & 1) * TY * TXP;
            const int y0 = t * TY, yn = min(TY, t_y - y0);
            typedef __attribute__((ext_vector_type(4))) float f32x4_t;
            f32x4_t vin[R / 4], vnext[R / 4];
#pragma unroll
            for (int q = 0; q < R / 4; ++q) vnext[q] = *reinterpret_cast<const f32x4_t *>(src + x0 + 4 * q);
            for (int j = 0; j < yn; ++j) {
                const int y = y0 + j;
                const int lo = max(0, t_x + y - t_y), hi = min(t_x, y + 1);
                const unsigned span = (unsigned)(hi - lo);
#pragma unroll
                for (int q = 0; q < R / 4; ++q) vin[q] = vnext[q];
                if (j + 1 < yn) {                                                 // next column's rows: in flight during this step's arithmetic
#pragma unroll
                    for (int q = 0; q < R / 4; ++q) vnext[q] = *reinterpret_cast<const f32x4_t *>(src + (j + 1) * TXP + x0 + 4 * q);
                }
                // value[x0-1][y-1]: the last row of lane l-1 (DPP wave shift, no LDS round trip); lane 0 has no row above: the
                // reference's 0 (first column) / max_neg (core.pyx:22-26)
                float left = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, prevv[R - 1]), 0x138, 0xf, 0xf, false));
                if (lane == 0) left = y == 0 ? 0.f : NEG;
                float nv[R];
                unsigned bits = 0;
                const unsigned xrel = (unsigned)(x0 - lo);                        // row x is inside the band iff xrel + r < span (unsigned)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float pm1 = r == 0 ? left : prevv[r - 1];               // value[x-1][y-1] (core.pyx:22-28)
                    // value[x][y-1] (core.pyx:18-21): the diagonal cell x == y was outside the previous column's band, and cells
                    // outside the band hold max_neg here -- the reference's `if x == y: v_cur = max_neg` without a compare
                    const float pc = prevv[r];
                    const float mx = fmaxf(pc, pm1);                              // (pc > pp ? pc : pp: no NaNs, no signed zeros that matter)
                    const float sum = __fadd_rn(mx, vin[r / 4][r % 4]);           // core.pyx:30 (computed for every lane: branch-free)
                    const bool in = xrel + (unsigned)r < span;
                    nv[r] = in ? sum : NEG;
                    // backtrack predicate (:34).  Bits of cells with x == 0, x == y or y == 0 are never consulted (the backtrack
                    // tests index != 0, index == y and y > 0 itself)
                    const bool lt = pc < pm1;
                    bits |= (unsigned)(in & lt) << r;
                }
#pragma unroll
                for (int r = 0; r < R; ++r) prevv[r] = nv[r];
                flg[(size_t)y * 64 + lane] = (unsigned short)bits;
            }
        }
        __syncthreads();
    }
    // ---- backtrack (core.pyx:32-35) in chunks of CH columns staged into LDS
    __threadfence_block();
    unsigned short *s_f = reinterpret_cast<unsigned short *>(sm);
    int index = t_x - 1;
    for (int c1 = t_y; c1 > 0; c1 -= CH) {
        const int c0 = max(0, c1 - CH);
        __syncthreads();
        for (int i = tid; i < (c1 - c0) * 64; i += 256) s_f[i] = flg[(size_t)c0 * 64 + i];
        __syncthreads();
        if (tid == 0) {
            for (int y = c1 - 1; y >= c0; --y) {
                path[base + (size_t)index * ty + y] = 1;
                if (index != 0) {
                    const unsigned w = s_f[(y - c0) * 64 + index / R];
                    if (index == y || (y > 0 && ((w >> (index % R)) & 1u))) index -= 1;
                }
            }
        }
    }
}

template <int R, int TY>
static hipError_t launch_mas_wave(const float *value, const float *mask, const int *t_x, const int *t_y, int *path,
                                  unsigned char *scratch, int b, int tx, int ty, hipStream_t st) {
    const size_t smem = (size_t)2 * TY * (64 * R + 4) * sizeof(float);
    static bool attr_done[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_done[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&mas_wave_kernel<R, TY>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) { (void)hipGetLastError(); return hipErrorNotSupported; }      // the ONE failure the caller may answer with the other kernel
        attr_done[dev] = true;
    }
    hipLaunchKernelGGL((mas_wave_kernel<R, TY>), dim3(b), dim3(256), smem, st, value, mask, t_x, t_y, path,
                       reinterpret_cast<unsigned short *>(scratch), tx, ty);
    return hipGetLastError();
}

// test hook (tests/test_gpu_parity.py compares the two kernels on the same input): process-wide, results are identical either way
static std::atomic<int> g_mas_force_sweep{0};
extern "C" void gtts_debug_mas_force_sweep(int on) { g_mas_force_sweep.store(on ? 1 : 0, std::memory_order_relaxed); }

hipError_t launch_mas(const float *value, const float *mask, const int *t_x, const int *t_y, int *path,
                      unsigned char *scratch, int b, int tx, int ty, hipStream_t st) {
    hipError_t e = hipMemsetAsync(path, 0, (size_t)b * tx * ty * sizeof(int), st);
    if (e != hipSuccess) return e;
    // one DP wave per sample (66-132 KB of LDS); if the device refuses that much dynamic LDS the column-sweep kernel below
    // (2 tx floats) computes the same path -- both restate core.pyx:9-35 cell for cell
    // (the column-sweep kernel runs when the device refuses the wave kernel's dynamic LDS, for tx > 1024, or when a test asked for
    // it through gtts_debug_mas_force_sweep; any OTHER failure of the wave kernel's launch is reported, not papered over)
    e = hipErrorNotSupported;
    if (g_mas_force_sweep.load(std::memory_order_relaxed)) { /* fall through to the column-sweep kernel */ }
    else if (tx <= 256) e = launch_mas_wave<4, 32>(value, mask, t_x, t_y, path, scratch, b, tx, ty, st);
    else if (tx <= 512) e = launch_mas_wave<8, 32>(value, mask, t_x, t_y, path, scratch, b, tx, ty, st);
    else if (tx <= 1024) e = launch_mas_wave<16, 16>(value, mask, t_x, t_y, path, scratch, b, tx, ty, st);
    if (e != hipErrorNotSupported) return e;
    const size_t smem = (size_t)2 * tx * sizeof(float);
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    if (smem > 48 * 1024) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&mas_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(mas_kernel, dim3(b), dim3(256), smem, st, value, mask, t_x, t_y, path, scratch, tx, ty);
    return hipGetLastError();
}

}  // namespace gtts
