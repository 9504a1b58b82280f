// train_wgrad.hip -- weight gradient of Block's 3x3 convolution (training hot path):
//     dW[co][ci][ky][kx] = sum_{b,y,x} dy[b,co,y,x] * (x * mask)[b,ci,y+ky-1,x+kx-1]          (autograd of diffusion.py:56-57)
// as an LDS-tiled MFMA reduction over PIXELS.  GEMM view per workgroup: D[co (64)][ci (64), tap (9)] += A[co][pixel] B[pixel][ci,tap],
// the contraction index is the frame axis (contiguous in NCHW).  A chunk is 2 image rows x 32 columns; per chunk the workgroup
// stages dy[64 co][2 rows][32 px] and (x * mask)[64 ci][4 rows][32 px + one edge column each side] into LDS, split ONCE into
// bf16 hi / lo planes ([channel][row][8-pixel block], a 16-byte slot per block; channel strides are odd numbers of slots:
// conflict-free ds_read_b128).  The lane's 8 k-values are the 8 pixels of one block: the A fragment is one aligned read; the
// B fragment of tap (ky, kx) is block (row + ky) for kx = 1 and, for kx = 0 / 2, the same block shifted by one PIXEL across the
// packed pairs with v_alignbit_b32 (5 per plane give both shifted fragments) using the neighbouring blocks' edge dwords (rows are
// padded with one slot each side for the tile's edge columns, so those dwords sit at fixed offsets from every block) --
// no unaligned LDS access and no second copy of the tile.  Precision: split-bf16, 3 MFMAs per product, fp32 accumulate.
// Parallelism over pixels: `nslice` workgroups per (co tile, ci tile) each own a contiguous range of chunks and write their
// 64 x 64 x 9 partial tile to the workspace; a second kernel adds the slices in a fixed order (deterministic, no atomics)
// and writes the reference layout [cout][cin][3][3].  The bias gradient falls out of the dy staging (every staged value is summed
// by the thread that stages it; the ci-tile-0 workgroups publish per-slice partials, reduced in the same second kernel).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdarg>
#include <cstdio>

#include "../../include/gradtts_abi.h"
#include "common.h"
#include "kernels.h"

namespace gtts {

struct Wgrad2Args {
    const float *x;        // [B][cin][H][W], or the first c0 channels of a concatenated input ...
    const float *x1;       // ... whose other cin - c0 channels are [B][cin - c0][H][W] here (nullptr: one source); c0 % 64 == 0
    int c0;
    const float *mask;     // [B][W]
    const float *dy;       // [B][cout][H][W]
    float *part;           // [nslice][tiles][9][64 co][64 ci]
    float *dbpart;         // [nslice][cout] partial bias gradients (written by the ci-tile-0 workgroups), or nullptr
    int B, cin, cout, H, W;
    int ncx, ncy;          // chunks per row (ceil(W / 32)) and per column (ceil(H / 2))
    int nchunk;            // B * ncy * ncx
    int nslice;
};

constexpr int DY_STRIDE = 9;     // 16-byte slots per co: 2 rows x 4 blocks + 1 pad (odd)
constexpr int X_ROW = 6;         // 16-byte slots per (ci, row): a left pad slot, 4 blocks, a right pad slot -- the pads hold the
                                 // edge columns x0 - 1 (last dword, high half) and x0 + 32 (first dword, low half), so that the
                                 // neighbour dwords of EVERY block sit at fixed offsets from it (one address register per lane)
constexpr int X_STRIDE = 25;     // 16-byte slots per ci: 4 rows x X_ROW + 1 pad (odd)

__device__ __forceinline__ void split8(const float (&v)[8], u32x4 &hi, u32x4 &lo) {
    bf16x8 vh, vl;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        __bf16 h, l;
        split_bf16(v[i], h, l);
        vh[i] = h;
        vl[i] = l;
    }
    hi = __builtin_bit_cast(u32x4, vh);
    lo = __builtin_bit_cast(u32x4, vl);
}

// Wave-specialised: waves 0-3 are CONSUMERS (one 32 co x 32 ci x 9 tap accumulator block each: fragment reads, the
// alignbit shifts and MFMAs only), waves 4-7 are PRODUCERS (global loads, mask, hi / lo split, LDS writes, bias sums).  The LDS
// tile is double-buffered and one barrier per chunk hands a buffer over: the producers stage chunk c + 1 (and have the loads
// of chunk c + 2 in flight across the barrier) while the consumers multiply chunk c.  (The uniform-wave form of round 2 --
// every wave loading, splitting, then multiplying between two barriers -- exposed the load latency and the split VALU work
// once per chunk: 190 us per launch where the MFMAs need 45.)
__global__ __launch_bounds__(512, 1) void conv3x3_wgrad2_kernel(const Wgrad2Args a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wg_smem[];
    constexpr int BUF16 = 2 * 64 * DY_STRIDE + 2 * 64 * X_STRIDE;      // 16-byte slots of one buffer
    u32x4 *s_base = reinterpret_cast<u32x4 *>(wg_smem);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ncit = a.cin / 64;
    const int tiles = ncit * (a.cout / 64);
    const int tile = blockIdx.x % tiles, slice = blockIdx.x / tiles;
    const int co0 = (tile / ncit) * 64, ci0 = (tile % ncit) * 64;
    const size_t HW = (size_t)a.H * a.W;
    const int per = (a.nchunk + a.nslice - 1) / a.nslice;
    const int c_begin = slice * per, c_end = min(a.nchunk, c_begin + per);

    if (wave < 4) {
        // =============================================================================== CONSUMERS
        const int l31 = lane & 31, kg = lane >> 5;
        const int wm = wave >> 1, wn = wave & 1;
        f32x16 acc[3][3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ky][kx][r] = 0.f;
        // (bottom-tested by hand: with the exit test at the top hipcc copies all 144 accumulator registers around the loop)
        if (c_begin < c_end) {
            int ch = c_begin;
            do {
                __syncthreads();                    // chunk ch is staged; the producers may overwrite the other buffer
                const u32x4 *s_dyh = s_base + ((ch - c_begin) & 1) * BUF16, *s_dyl = s_dyh + 64 * DY_STRIDE;
                const u32x4 *s_xh = s_dyl + 64 * DY_STRIDE, *s_xl = s_xh + 64 * X_STRIDE;
                // ---- 4 k-steps of 16 pixels: (row r, column half cb); lane's block = 2 cb + kg
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int r = ks >> 1, blk = (ks & 1) * 2 + kg;
                    const int ai = (wm * 32 + l31) * DY_STRIDE + r * 4 + blk;
                    const bf16x8 Ah = __builtin_bit_cast(bf16x8, s_dyh[ai]), Al = __builtin_bit_cast(bf16x8, s_dyl[ai]);
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const int rr = r + ky, ci = wn * 32 + l31;
                        const int bi = ci * X_STRIDE + rr * X_ROW + 1 + blk;
                        const unsigned *xh32 = reinterpret_cast<const unsigned *>(s_xh), *xl32 = reinterpret_cast<const unsigned *>(s_xl);
                        const u32x4 dh = s_xh[bi], dl = s_xl[bi];
                        // P: dword holding the pixel left of the block in its HIGH half; N: dword holding the pixel right of it in its LOW half
                        const unsigned Ph = xh32[bi * 4 - 1], Pl = xl32[bi * 4 - 1], Nh = xh32[bi * 4 + 4], Nl = xl32[bi * 4 + 4];
                        // shifted fragments: (e[-1], e0) (e1, e2) (e3, e4) (e5, e6)   and   (e1, e2) (e3, e4) (e5, e6) (e7, e[8])
                        const unsigned m1h = __builtin_amdgcn_alignbit(dh[1], dh[0], 16), m2h = __builtin_amdgcn_alignbit(dh[2], dh[1], 16),
                                       m3h = __builtin_amdgcn_alignbit(dh[3], dh[2], 16);
                        const unsigned m1l = __builtin_amdgcn_alignbit(dl[1], dl[0], 16), m2l = __builtin_amdgcn_alignbit(dl[2], dl[1], 16),
                                       m3l = __builtin_amdgcn_alignbit(dl[3], dl[2], 16);
                        u32x4 b0h, b0l, b2h, b2l;
                        b0h[0] = __builtin_amdgcn_alignbit(dh[0], Ph, 16); b0h[1] = m1h; b0h[2] = m2h; b0h[3] = m3h;
                        b0l[0] = __builtin_amdgcn_alignbit(dl[0], Pl, 16); b0l[1] = m1l; b0l[2] = m2l; b0l[3] = m3l;
                        b2h[0] = m1h; b2h[1] = m2h; b2h[2] = m3h; b2h[3] = __builtin_amdgcn_alignbit(Nh, dh[3], 16);
                        b2l[0] = m1l; b2l[1] = m2l; b2l[2] = m3l; b2l[3] = __builtin_amdgcn_alignbit(Nl, dl[3], 16);
                        const bf16x8 Bh[3] = {__builtin_bit_cast(bf16x8, b0h), __builtin_bit_cast(bf16x8, dh), __builtin_bit_cast(bf16x8, b2h)};
                        const bf16x8 Bl[3] = {__builtin_bit_cast(bf16x8, b0l), __builtin_bit_cast(bf16x8, dl), __builtin_bit_cast(bf16x8, b2l)};
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            acc[ky][kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh[kx], acc[ky][kx], 0, 0, 0);
                            acc[ky][kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl[kx], acc[ky][kx], 0, 0, 0);
                            acc[ky][kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh[kx], acc[ky][kx], 0, 0, 0);
                        }
                    }
                }
            } while (++ch < c_end);
        }
        // ---- partial tile: D[m = co][n = ci]; lane (l31 = ci, kg) holds rows (rg&3) + 8 (rg>>2) + 4 kg
        float *out = a.part + ((size_t)slice * tiles + tile) * (9 * 64 * 64);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const int co = wm * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * kg;
                    out[((ky * 3 + kx) * 64 + co) * 64 + wn * 32 + l31] = acc[ky][kx][rg];
                }
        return;
    }
    // =================================================================================== PRODUCERS
    const int ptid = tid - 256;
    float bsum[2] = {0.f, 0.f};            // bias gradient: this thread's dy items belong to co = (ptid >> 3) + 32 k
    // Global loads: 16-byte buffer loads at dword-aligned offsets (rows of an NCHW plane start anywhere).  Columns past the
    // row end alias the next row and are zeroed by selects; rows outside the image and whole items past the tensor read
    // offset 0 and are zeroed the same way.
    // the workgroup's 64 input channels lie in one source (torch.cat of the up path read in place: diffusion.py:166)
    const bool second = a.x1 != nullptr && ci0 >= a.c0;
    const float *xsrc = second ? a.x1 : a.x;
    const int xc = a.x1 == nullptr ? a.cin : (second ? a.cin - a.c0 : a.c0), xc0 = second ? ci0 - a.c0 : ci0;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xsrc), 0, (int)((size_t)a.B * xc * HW * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.dy), 0, (int)((size_t)a.B * a.cout * HW * 4), 0x00020000);
    u32x4 xr[4][2], dr[2][2];
    float er[2], mk[8];
    auto issue = [&](int ch) {
        const int cx = ch % a.ncx, cy = (ch / a.ncx) % a.ncy, b = ch / (a.ncx * a.ncy);
        const int x0 = cx * 32, y0 = cy * 2;
#pragma unroll
        for (int i = 0; i < 8; ++i) mk[i] = a.mask[(size_t)b * a.W + min(x0 + 8 * (ptid & 3) + i, a.W - 1)];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int item = ptid + 256 * k, blk = item & 3, rr = (item >> 2) & 3, ci = item >> 4;
            const int y = min(max(y0 - 1 + rr, 0), a.H - 1), px = min(x0 + 8 * blk, a.W - 1);
            const int off = (int)((((size_t)b * xc + xc0 + ci) * HW + (size_t)y * a.W + px) * 4);
            xr[k][0] = __builtin_amdgcn_raw_buffer_load_b128(rsx, off, 0, 0);
            xr[k][1] = __builtin_amdgcn_raw_buffer_load_b128(rsx, off + 16, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int item = ptid + 256 * k, blk = item & 3, r = (item >> 2) & 1, co = item >> 3;
            const int y = min(y0 + r, a.H - 1), px = min(x0 + 8 * blk, a.W - 1);
            const int off = (int)((((size_t)b * a.cout + co0 + co) * HW + (size_t)y * a.W + px) * 4);
            dr[k][0] = __builtin_amdgcn_raw_buffer_load_b128(rsd, off, 0, 0);
            dr[k][1] = __builtin_amdgcn_raw_buffer_load_b128(rsd, off + 16, 0, 0);
        }
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int item = ptid + 256 * k, side = item & 1, rr = (item >> 1) & 3, ci = item >> 3;
            const int y = y0 - 1 + rr, px = side ? x0 + 32 : x0 - 1;
            const bool ok = y >= 0 && y < a.H && px >= 0 && px < a.W;
            const int pc = min(max(px, 0), a.W - 1);
            const float t = xsrc[((size_t)b * xc + xc0 + ci) * HW + (size_t)(ok ? y : 0) * a.W + pc] * a.mask[(size_t)b * a.W + pc];
            er[k] = ok ? t : 0.f;
        }
    };
    if (c_begin < c_end) issue(c_begin);
    for (int ch = c_begin; ch < c_end; ++ch) {
        const int cx = ch % a.ncx, cy = (ch / a.ncx) % a.ncy;
        const int x0 = cx * 32, y0 = cy * 2;
        u32x4 *s_dyh = s_base + ((ch - c_begin) & 1) * BUF16, *s_dyl = s_dyh + 64 * DY_STRIDE;
        u32x4 *s_xh = s_dyl + 64 * DY_STRIDE, *s_xl = s_xh + 64 * X_STRIDE;
        // ---- x * mask: 1024 main items (ci, row 0..3 = image rows y0-1..y0+2, block) -> bf16 hi / lo
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int item = ptid + 256 * k, blk = item & 3, rr = (item >> 2) & 3, ci = item >> 4;
            const int y = y0 - 1 + rr, px = x0 + 8 * blk;
            const bool rowok = y >= 0 && y < a.H;
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned w = xr[k][i >> 2][i & 3];      // (copy first: __builtin_bit_cast of a vector ELEMENT reads element 0)
                const float t = __builtin_bit_cast(float, w);
                v[i] = (rowok && px + i < a.W) ? t * mk[i] : 0.f;
            }
            u32x4 hi, lo;
            split8(v, hi, lo);
            s_xh[ci * X_STRIDE + rr * X_ROW + 1 + blk] = hi;
            s_xl[ci * X_STRIDE + rr * X_ROW + 1 + blk] = lo;
        }
        // ---- dy: 512 items (co, row, block)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int item = ptid + 256 * k, blk = item & 3, r = (item >> 2) & 1, co = item >> 3;
            const int y = y0 + r, px = x0 + 8 * blk;
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned w = dr[k][i >> 2][i & 3];
                const float t = __builtin_bit_cast(float, w);
                v[i] = (y < a.H && px + i < a.W) ? t : 0.f;
                bsum[k] += v[i];
            }
            u32x4 hi, lo;
            split8(v, hi, lo);
            s_dyh[co * DY_STRIDE + r * 4 + blk] = hi;
            s_dyl[co * DY_STRIDE + r * 4 + blk] = lo;
        }
        // ---- the two edge columns (x0 - 1 -> high half, x0 + 32 -> low half of the stored dword): 512 items
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int item = ptid + 256 * k, side = item & 1, rr = (item >> 1) & 3, ci = item >> 3;
            __bf16 h, l;
            split_bf16(er[k], h, l);
            const unsigned hb = (unsigned)__builtin_bit_cast(unsigned short, h), lb = (unsigned)__builtin_bit_cast(unsigned short, l);
            const int di = (ci * X_STRIDE + rr * X_ROW + (side ? 5 : 0)) * 4 + (side ? 0 : 3);      // dword inside the pad slot
            reinterpret_cast<unsigned *>(s_xh)[di] = side ? hb : hb << 16;
            reinterpret_cast<unsigned *>(s_xl)[di] = side ? lb : lb << 16;
        }
        if (ch + 1 < c_end) issue(ch + 1);      // in flight across the barrier wait
        __syncthreads();                        // chunk ch handed over; the consumers are done with the buffer of chunk ch - 1
    }
    if (a.dbpart && ci0 == 0) {
        // the 8 threads (row, block) of a co are consecutive lanes: fixed-order butterfly, lane 0 of each octet publishes
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float v = bsum[k];
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 4, 64);
            if ((ptid & 7) == 0) a.dbpart[(size_t)slice * a.cout + co0 + (ptid >> 3) + 32 * k] = v;
        }
    }
}

// dW[co][ci][3][3] = sum over slices (fixed order).  A workgroup owns 32 consecutive partial-tile elements (tile, tap, co, ci;
// ci fastest: 128-byte rows) x 8 slice groups: every thread adds its group's slices with eight loads in flight, the groups are
// combined through LDS in a fixed order.  The last workgroups of the grid reduce the bias partials the same way.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ part, const float *__restrict__ dbpart,
                                                           float *__restrict__ dw, float *__restrict__ db, int cin, int cout,
                                                           int nslice, int taps) {
    __shared__ float s_red[8][32];
    const int ncit = cin / 64, tiles = ncit * (cout / 64);
    const size_t tile_elems = (size_t)taps * 64 * 64, total = (size_t)tiles * tile_elems;
    const int e32 = threadIdx.x & 31, sg = threadIdx.x >> 5;
    const size_t idx = (size_t)blockIdx.x * 32 + e32;
    const bool is_w = idx < total;                       // (total is a multiple of 32: a workgroup is all-weights or all-bias)
    const size_t bco = idx - total;
    const float *p = is_w ? part + idx : dbpart + (bco < (size_t)cout ? bco : 0);
    const size_t stride = is_w ? total : (size_t)cout;
    const int per = (nslice + 7) / 8, s0 = sg * per, s1 = min(nslice, s0 + per);
    float s = 0.f;
    if (is_w || (db && dbpart && bco < (size_t)cout)) {
        int sl = s0;
        for (; sl + 8 <= s1; sl += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(sl + u) * stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; sl < s1; ++sl) s += p[(size_t)sl * stride];
    }
    s_red[sg][e32] = s;
    __syncthreads();
    if (sg == 0) {
        float t = s_red[0][e32];
#pragma unroll
        for (int g = 1; g < 8; ++g) t += s_red[g][e32];
        if (is_w) {
            const int tile = (int)(idx / tile_elems), e = (int)(idx % tile_elems);
            const int tap = e / 4096, co = (tile / ncit) * 64 + (e >> 6) % 64, ci = (tile % ncit) * 64 + (e & 63);
            dw[((size_t)co * cin + ci) * taps + tap] = t;
        } else if (db && dbpart && bco < (size_t)cout) {
            db[bco] = t;
        }
    }
}

// ---------------------------------------------------------------------------------------------------- 1x1 convolutions
// dW[co][ci] = sum_{b,p} dy[b,co,p] * (x * mask)[b,ci,p]  (res_conv, to_qkv, to_out: diffusion.py:70,87-88) -- the same
// pixel-contraction GEMM without taps: a chunk is 64 consecutive pixels of one sample's flattened H x W plane, the workgroup
// tile 64 co x 64 ci (four waves, one 32 x 32 accumulator each), split bf16 hi / lo planes in LDS ([channel][8-pixel block],
// odd channel stride), the loads of chunk c + 1 in flight across the MFMAs of chunk c, slices over pixels reduced by
// wgrad_reduce_kernel in a fixed order.
struct Wgrad1Args {
    const float *x;        // [B][cin][HW]
    const float *mask;     // [B][W] or nullptr
    const float *dy;       // [B][cout][HW]
    float *part;           // [nslice][tiles][64 co][64 ci]
    float *dbpart;         // [nslice][cout] or nullptr
    int B, cin, cout, HW, W;
    int cps;               // chunks per sample: ceil(HW / 64)
    int nchunk, nslice;
};

__global__ __launch_bounds__(256, 2) void conv1x1_wgrad_kernel(const Wgrad1Args a) {
    __shared__ __attribute__((aligned(16))) u32x4 s_dy[2][64 * DY_STRIDE], s_x[2][64 * DY_STRIDE];     // [hi | lo][channel][8 blocks + pad]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kg = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int ncit = a.cin / 64;
    const int tiles = ncit * (a.cout / 64);
    const int tile = blockIdx.x % tiles, slice = blockIdx.x / tiles;
    const int co0 = (tile / ncit) * 64, ci0 = (tile % ncit) * 64;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float bsum[2] = {0.f, 0.f};
    const int per = (a.nchunk + a.nslice - 1) / a.nslice;
    const int c_begin = slice * per, c_end = min(a.nchunk, c_begin + per);
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.x), 0, (int)((size_t)a.B * a.cin * a.HW * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsd = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.dy), 0, (int)((size_t)a.B * a.cout * a.HW * 4), 0x00020000);
    u32x4 xr[2][2], dr[2][2];
    auto issue = [&](int ch) {
        const int b = ch / a.cps, p0 = (ch % a.cps) * 64;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int item = tid + 256 * k, blk = item & 7, c = item >> 3;
            const int p = min(p0 + 8 * blk, a.HW - 1);
            const int ox = (int)((((size_t)b * a.cin + ci0 + c) * a.HW + p) * 4), od = (int)((((size_t)b * a.cout + co0 + c) * a.HW + p) * 4);
            xr[k][0] = __builtin_amdgcn_raw_buffer_load_b128(rsx, ox, 0, 0);
            xr[k][1] = __builtin_amdgcn_raw_buffer_load_b128(rsx, ox + 16, 0, 0);
            dr[k][0] = __builtin_amdgcn_raw_buffer_load_b128(rsd, od, 0, 0);
            dr[k][1] = __builtin_amdgcn_raw_buffer_load_b128(rsd, od + 16, 0, 0);
        }
    };
    if (c_begin < c_end) issue(c_begin);
    for (int ch = c_begin; ch < c_end; ++ch) {
        const int b = ch / a.cps, p0 = (ch % a.cps) * 64;
        __syncthreads();                    // the previous chunk's fragment reads are done
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int item = tid + 256 * k, blk = item & 7, c = item >> 3;
            const int p = p0 + 8 * blk;
            float vx[8], vd[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned wx = xr[k][i >> 2][i & 3], wd = dr[k][i >> 2][i & 3];
                const bool ok = p + i < a.HW;
                const float m = a.mask ? a.mask[(size_t)b * a.W + (ok ? (p + i) % a.W : 0)] : 1.f;
                vx[i] = ok ? __builtin_bit_cast(float, wx) * m : 0.f;
                vd[i] = ok ? __builtin_bit_cast(float, wd) : 0.f;
                bsum[k] += vd[i];
            }
            u32x4 hi, lo;
            split8(vx, hi, lo);
            s_x[0][c * DY_STRIDE + blk] = hi;
            s_x[1][c * DY_STRIDE + blk] = lo;
            split8(vd, hi, lo);
            s_dy[0][c * DY_STRIDE + blk] = hi;
            s_dy[1][c * DY_STRIDE + blk] = lo;
        }
        __syncthreads();
        if (ch + 1 < c_end) issue(ch + 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int blk = 2 * ks + kg;
            const int ai = (wm * 32 + l31) * DY_STRIDE + blk, bi = (wn * 32 + l31) * DY_STRIDE + blk;
            const bf16x8 Ah = __builtin_bit_cast(bf16x8, s_dy[0][ai]), Al = __builtin_bit_cast(bf16x8, s_dy[1][ai]);
            const bf16x8 Bh = __builtin_bit_cast(bf16x8, s_x[0][bi]), Bl = __builtin_bit_cast(bf16x8, s_x[1][bi]);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, acc, 0, 0, 0);
        }
    }
    float *out = a.part + ((size_t)slice * tiles + tile) * (64 * 64);
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
        const int co = wm * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * kg;
        out[co * 64 + wn * 32 + l31] = acc[rg];
    }
    if (a.dbpart && ci0 == 0) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            float v = bsum[k];
            v += __shfl_xor(v, 1, 64);
            v += __shfl_xor(v, 2, 64);
            v += __shfl_xor(v, 4, 64);
            if ((tid & 7) == 0) a.dbpart[(size_t)slice * a.cout + co0 + (tid >> 3) + 32 * k] = v;
        }
    }
}

static void wgrad1_geometry(int B, int cin, int cout, int HW, int W, Wgrad1Args &a) {
    a.B = B; a.cin = cin; a.cout = cout; a.HW = HW; a.W = W;
    a.cps = (HW + 63) / 64;
    a.nchunk = B * a.cps;
    const int tiles = (cin / 64) * (cout / 64);
    int nslice = (512 + tiles - 1) / tiles;       // two workgroups per CU, at least four chunks each
    nslice = std::max(1, std::min(nslice, (a.nchunk + 3) / 4));
    a.nslice = nslice;
}

static void wgrad2_geometry(int B, int cin, int cout, int H, int W, Wgrad2Args &a) {
    a.B = B; a.cin = cin; a.cout = cout; a.H = H; a.W = W;
    a.ncx = (W + 31) / 32; a.ncy = (H + 1) / 2;
    a.nchunk = B * a.ncy * a.ncx;
    const int tiles = (cin / 64) * (cout / 64);
    // one eight-wave workgroup per CU across the chip, at least four chunks (256 pixels) per workgroup
    int nslice = (256 + tiles - 1) / tiles;
    nslice = std::max(1, std::min(nslice, (a.nchunk + 3) / 4));
    a.nslice = nslice;
}

}  // namespace gtts

using namespace gtts;

static int wfail(int code, const char *fmt, ...) {       // text goes to gtts_last_error() (plan.hip)
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return set_error(code, buf);
}
#define WCHK(expr)                                                                                                \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) return wfail(GTTS_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

extern "C" size_t gtts_conv3x3_wgrad_workspace_bytes(int B, int cin, int cout, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0 || cin % 64 || cout % 64) return 0;
    Wgrad2Args a;
    wgrad2_geometry(B, cin, cout, H, W, a);
    return ((size_t)a.nslice * (cin / 64) * (cout / 64) * (9 * 64 * 64) + (size_t)a.nslice * cout) * sizeof(float);
}

// x1 (nullable) / c0: the input is the channel concatenation of x [B,c0,H,W] and x1 [B,cin-c0,H,W] (c0 a multiple of 64)
extern "C" int gtts_conv3x3_wgrad_tiled2(const float *x, const float *x1, int c0, const float *mask, const float *dy, float *dw, float *db,
                                         void *workspace, size_t workspace_bytes, int B, int cin, int cout, int H, int W,
                                         gtts_stream_t stream) {
    if (!x || !mask || !dy || !dw || !workspace) return wfail(GTTS_E_NULL, "gtts_conv3x3_wgrad_tiled: null argument");
    if (x1 && (c0 <= 0 || c0 >= cin || c0 % 64)) return wfail(GTTS_E_SHAPE, "gtts_conv3x3_wgrad_tiled: c0 must be a multiple of 64 inside (0, cin) (got %d of %d)", c0, cin);
    if (B <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0 || cin % 64 || cout % 64)
        return wfail(GTTS_E_SHAPE, "gtts_conv3x3_wgrad_tiled: cin and cout must be multiples of 64 (got %d, %d)", cin, cout);
    if ((size_t)std::max(cin, cout) * H * W >= ((size_t)1 << 30)) return wfail(GTTS_E_SHAPE, "gtts_conv3x3_wgrad_tiled: tensor too large");
    Wgrad2Args a;
    wgrad2_geometry(B, cin, cout, H, W, a);
    const size_t need = gtts_conv3x3_wgrad_workspace_bytes(B, cin, cout, H, W);
    if (workspace_bytes < need) return wfail(GTTS_E_WORKSPACE, "gtts_conv3x3_wgrad_tiled: workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    const int tiles = (cin / 64) * (cout / 64);
    a.x = x; a.x1 = x1; a.c0 = x1 ? c0 : cin; a.mask = mask; a.dy = dy; a.part = (float *)workspace;
    a.dbpart = db ? a.part + (size_t)a.nslice * tiles * (9 * 64 * 64) : nullptr;
    hipStream_t st = (hipStream_t)stream;
    constexpr size_t smem = (size_t)2 * (2 * 64 * DY_STRIDE + 2 * 64 * X_STRIDE) * 16;      // two buffers
    static std::atomic<int> attr_set[64];        // hipFuncSetAttribute is per device
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev].load(std::memory_order_relaxed)) {
        WCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&conv3x3_wgrad2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
        attr_set[dev].store(1, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(conv3x3_wgrad2_kernel, dim3((unsigned)(tiles * a.nslice)), dim3(512), smem, st, a);
    WCHK(hipGetLastError());
    const size_t total = (size_t)tiles * (9 * 64 * 64) + (db ? (size_t)cout : 0);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, st, a.part, a.dbpart, dw, db, cin, cout,
                       a.nslice, 9);
    WCHK(hipGetLastError());
    return GTTS_OK;
}

extern "C" size_t gtts_conv1x1_wgrad_workspace_bytes(int B, int cin, int cout, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0 || cin % 64 || cout % 64) return 0;
    Wgrad1Args a;
    wgrad1_geometry(B, cin, cout, H * W, W, a);
    return ((size_t)a.nslice * (cin / 64) * (cout / 64) * (64 * 64) + (size_t)a.nslice * cout) * sizeof(float);
}

// dw [cout][cin], db [cout] (or null) of y = Conv2d_1x1(x * mask) + bias; mask [B][W] columns or null (no mask)
extern "C" int gtts_conv1x1_wgrad(const float *x, const float *mask, const float *dy, float *dw, float *db, void *workspace,
                                  size_t workspace_bytes, int B, int cin, int cout, int H, int W, gtts_stream_t stream) {
    if (!x || !dy || !dw || !workspace) return wfail(GTTS_E_NULL, "gtts_conv1x1_wgrad: null argument");
    if (B <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0 || cin % 64 || cout % 64)
        return wfail(GTTS_E_SHAPE, "gtts_conv1x1_wgrad: cin and cout must be multiples of 64 (got %d, %d)", cin, cout);
    if ((size_t)B * std::max(cin, cout) * H * W >= ((size_t)1 << 29)) return wfail(GTTS_E_SHAPE, "gtts_conv1x1_wgrad: tensor too large");
    Wgrad1Args a;
    wgrad1_geometry(B, cin, cout, H * W, W, a);
    const size_t need = gtts_conv1x1_wgrad_workspace_bytes(B, cin, cout, H, W);
    if (workspace_bytes < need) return wfail(GTTS_E_WORKSPACE, "gtts_conv1x1_wgrad: workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    const int tiles = (cin / 64) * (cout / 64);
    a.x = x; a.mask = mask; a.dy = dy; a.part = (float *)workspace;
    a.dbpart = db ? a.part + (size_t)a.nslice * tiles * (64 * 64) : nullptr;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(conv1x1_wgrad_kernel, dim3((unsigned)(tiles * a.nslice)), dim3(256), 0, st, a);
    WCHK(hipGetLastError());
    const size_t total = (size_t)tiles * (64 * 64) + (db ? (size_t)cout : 0);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, st, a.part, a.dbpart, dw, db, cin, cout,
                       a.nslice, 1);
    WCHK(hipGetLastError());
    return GTTS_OK;
}

extern "C" int gtts_conv3x3_wgrad_tiled(const float *x, const float *mask, const float *dy, float *dw, float *db, void *workspace,
                                        size_t workspace_bytes, int B, int cin, int cout, int H, int W, gtts_stream_t stream) {
    return gtts_conv3x3_wgrad_tiled2(x, nullptr, 0, mask, dy, dw, db, workspace, workspace_bytes, B, cin, cout, H, W, stream);
}
