// train_wgrad.hip -- weight gradient of Block's 3x3 convolution (training hot path):
//     dW[co][ci][ky][kx] = sum_{b,y,x} dy[b,co,y,x] * (x * mask)[b,ci,y+ky-1,x+kx-1]          (autograd of diffusion.py:56-57)
// as an LDS-tiled MFMA reduction over PIXELS.  GEMM view per workgroup: D[co (64)][ci (64), tap (9)] += A[co][pixel] B[pixel][ci,tap],
// the contraction index is the frame axis (contiguous in NCHW).  A chunk is 2 image rows x 32 columns; per chunk the workgroup
// stages dy[64 co][2 rows][32 px] and (x * mask)[64 ci][4 rows][32 px + one edge column each side] into LDS, split ONCE into
// bf16 hi / lo planes ([channel][row][8-pixel block], a 16-byte slot per block; channel strides are odd numbers of slots:
// conflict-free ds_read_b128).  The lane's 8 k-values are the 8 pixels of one block: the A fragment is one aligned read; the
// B fragment of tap (ky, kx) is block (row + ky) for kx = 1 and, for kx = 0 / 2, the same block shifted by one PIXEL across the
// packed pairs with v_alignbit_b32 (5 per plane give both shifted fragments) using the neighbouring blocks' edge dwords --
// no unaligned LDS access and no second copy of the tile.  Precision: split-bf16, 3 MFMAs per product, fp32 accumulate.
// Parallelism over pixels: `nslice` workgroups per (co tile, ci tile) each own a contiguous range of chunks and write their
// 64 x 64 x 9 partial tile to the workspace; a second kernel adds the slices in a fixed order (deterministic, no atomics)
// and writes the reference layout [cout][cin][3][3].  The bias gradient falls out of the dy staging (every staged value is summed
// by the thread that stages it; the ci-tile-0 workgroups publish per-slice partials, reduced in the same second kernel).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>

#include "../../include/gradtts_abi.h"
#include "common.h"
#include "kernels.h"

namespace gtts {

struct Wgrad2Args {
    const float *x;        // [B][cin][H][W]
    const float *mask;     // [B][W] or nullptr
    const float *dy;       // [B][cout][H][W]
    float *part;           // [nslice][tiles][9][64 co][64 ci]
    float *dbpart;         // [nslice][cout] partial bias gradients (written by the ci-tile-0 workgroups), or nullptr
    int B, cin, cout, H, W;
    int ncx, ncy;          // chunks per row (ceil(W / 32)) and per column (ceil(H / 2))
    int nchunk;            // B * ncy * ncx
    int nslice;
};

constexpr int DY_STRIDE = 9;     // 16-byte slots per co: 2 rows x 4 blocks + 1 pad (odd)
constexpr int X_STRIDE = 17;     // 16-byte slots per ci: 4 rows x 4 blocks + 1 pad (odd)

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

__global__ __launch_bounds__(256, 2) void conv3x3_wgrad2_kernel(const Wgrad2Args a) {
    __shared__ __attribute__((aligned(16))) u32x4 s_dyh[64 * DY_STRIDE], s_dyl[64 * DY_STRIDE];
    __shared__ __attribute__((aligned(16))) u32x4 s_xh[64 * X_STRIDE], s_xl[64 * X_STRIDE];
    __shared__ unsigned s_eh[64 * 4 * 2], s_el[64 * 4 * 2];     // edge columns -1 / 32: [ci][row][side], element in the half used

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kg = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int ncit = a.cin / 64;
    const int tiles = ncit * (a.cout / 64);
    const int tile = blockIdx.x % tiles, slice = blockIdx.x / tiles;
    const int co0 = (tile / ncit) * 64, ci0 = (tile % ncit) * 64;
    const size_t HW = (size_t)a.H * a.W;

    f32x16 acc[3][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ky][kx][r] = 0.f;

    float bsum[2] = {0.f, 0.f};            // bias gradient: this thread's dy items belong to co = (tid >> 3) + 32 k
    const int per = (a.nchunk + a.nslice - 1) / a.nslice;
    const int c_begin = slice * per, c_end = min(a.nchunk, c_begin + per);
    for (int ch = c_begin; ch < c_end; ++ch) {
        const int cx = ch % a.ncx, cy = (ch / a.ncx) % a.ncy, b = ch / (a.ncx * a.ncy);
        const int x0 = cx * 32, y0 = cy * 2;
        // this thread's x items all cover the same 8 columns (block tid & 3): their mask values are loaded once per chunk
        float mk[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) mk[i] = a.mask ? a.mask[(size_t)b * a.W + min(x0 + 8 * (tid & 3) + i, a.W - 1)] : 1.f;
        const float *mrow = a.mask ? a.mask + (size_t)b * a.W : nullptr;
        __syncthreads();                    // the previous chunk's fragment reads are done
        // ---- stage dy: 512 items (co, row, block)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int item = tid + 256 * k, blk = item & 3, r = (item >> 2) & 1, co = item >> 3;
            const int y = y0 + r, px = x0 + 8 * blk;
            const float *p = a.dy + ((size_t)b * a.cout + co0 + co) * HW + (size_t)min(y, a.H - 1) * a.W;
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool ok = y < a.H && px + i < a.W;
                v[i] = ok ? p[min(px + i, a.W - 1)] : 0.f;
                bsum[k] += v[i];
            }
            u32x4 hi, lo;
            split8(v, hi, lo);
            s_dyh[co * DY_STRIDE + r * 4 + blk] = hi;
            s_dyl[co * DY_STRIDE + r * 4 + blk] = lo;
        }
        // ---- stage x * mask: 1024 main items (ci, row 0..3 = image rows y0-1..y0+2, block)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int item = tid + 256 * k, blk = item & 3, rr = (item >> 2) & 3, ci = item >> 4;
            const int y = y0 - 1 + rr, px = x0 + 8 * blk;
            const bool rowok = y >= 0 && y < a.H;
            const float *p = a.x + ((size_t)b * a.cin + ci0 + ci) * HW + (size_t)(rowok ? y : 0) * a.W;
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool ok = rowok && px + i < a.W;
                const int pc = min(px + i, a.W - 1);
                v[i] = (ok ? p[pc] : 0.f) * mk[i];
            }
            u32x4 hi, lo;
            split8(v, hi, lo);
            s_xh[ci * X_STRIDE + rr * 4 + blk] = hi;
            s_xl[ci * X_STRIDE + rr * 4 + blk] = lo;
        }
        // ---- the two edge columns (x0 - 1 -> high half, x0 + 32 -> low half of the stored dword): 512 items
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int item = tid + 256 * k, side = item & 1, rr = (item >> 1) & 3, ci = item >> 3;
            const int y = y0 - 1 + rr, px = side ? x0 + 32 : x0 - 1;
            const bool ok = y >= 0 && y < a.H && px >= 0 && px < a.W;
            const int pc = min(max(px, 0), a.W - 1);
            float t = ok ? a.x[((size_t)b * a.cin + ci0 + ci) * HW + (size_t)(ok ? y : 0) * a.W + pc] : 0.f;
            if (mrow) t *= mrow[pc];
            __bf16 h, l;
            split_bf16(t, h, l);
            const unsigned hb = (unsigned)__builtin_bit_cast(unsigned short, h), lb = (unsigned)__builtin_bit_cast(unsigned short, l);
            s_eh[item] = side ? hb : hb << 16;
            s_el[item] = side ? lb : lb << 16;
        }
        __syncthreads();
        // ---- 4 k-steps of 16 pixels: (row r, column half cb); lane's block = 2 cb + kg
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int r = ks >> 1, blk = (ks & 1) * 2 + kg;
            const int ai = (wm * 32 + l31) * DY_STRIDE + r * 4 + blk;
            const bf16x8 Ah = __builtin_bit_cast(bf16x8, s_dyh[ai]), Al = __builtin_bit_cast(bf16x8, s_dyl[ai]);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int rr = r + ky, ci = wn * 32 + l31;
                const int bi = ci * X_STRIDE + rr * 4 + blk;
                const unsigned *xh32 = reinterpret_cast<const unsigned *>(s_xh), *xl32 = reinterpret_cast<const unsigned *>(s_xl);
                const u32x4 dh = s_xh[bi], dl = s_xl[bi];
                // P: dword holding the pixel left of the block in its HIGH half; N: dword holding the pixel right of it in its LOW half
                const int ei = (ci * 4 + rr) * 2;
                const unsigned Ph = blk > 0 ? xh32[(bi - 1) * 4 + 3] : s_eh[ei], Pl = blk > 0 ? xl32[(bi - 1) * 4 + 3] : s_el[ei];
                const unsigned Nh = blk < 3 ? xh32[(bi + 1) * 4] : s_eh[ei + 1], Nl = blk < 3 ? xl32[(bi + 1) * 4] : s_el[ei + 1];
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
    if (a.dbpart && ci0 == 0) {
        // the 8 threads (row, block) of a co are consecutive lanes: fixed-order butterfly, lane 0 of each octet publishes
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

// dW[co][ci][3][3] = sum over slices (fixed order).  One thread per partial-tile element (tile, tap, co, ci), ci fastest:
// coalesced reads, eight slices in flight per thread; the last blocks of the grid reduce the bias partials the same way.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float *__restrict__ part, const float *__restrict__ dbpart,
                                                           float *__restrict__ dw, float *__restrict__ db, int cin, int cout,
                                                           int nslice) {
    const int ncit = cin / 64, tiles = ncit * (cout / 64);
    const size_t tile_elems = 9 * 64 * 64, total = (size_t)tiles * tile_elems;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < total) {
        const float *p = part + idx;
        float s = 0.f;
        int sl = 0;
        for (; sl + 8 <= nslice; sl += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(sl + u) * total];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; sl < nslice; ++sl) s += p[(size_t)sl * total];
        const int tile = (int)(idx / tile_elems), e = (int)(idx % tile_elems);
        const int tap = e / 4096, co = (tile / ncit) * 64 + (e >> 6) % 64, ci = (tile % ncit) * 64 + (e & 63);
        dw[((size_t)co * cin + ci) * 9 + tap] = s;
    } else if (db && dbpart) {
        const size_t co = idx - total;
        if (co < (size_t)cout) {
            float s = 0.f;
            for (int sl = 0; sl < nslice; ++sl) s += dbpart[(size_t)sl * cout + co];
            db[co] = s;
        }
    }
}

static void wgrad2_geometry(int B, int cin, int cout, int H, int W, Wgrad2Args &a) {
    a.B = B; a.cin = cin; a.cout = cout; a.H = H; a.W = W;
    a.ncx = (W + 31) / 32; a.ncy = (H + 1) / 2;
    a.nchunk = B * a.ncy * a.ncx;
    const int tiles = (cin / 64) * (cout / 64);
    // about three workgroups per CU across the chip, at least four chunks (256 pixels) per workgroup
    int nslice = (768 + tiles - 1) / tiles;
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

extern "C" int gtts_conv3x3_wgrad_tiled(const float *x, const float *mask, const float *dy, float *dw, float *db, void *workspace,
                                        size_t workspace_bytes, int B, int cin, int cout, int H, int W, gtts_stream_t stream) {
    if (!x || !dy || !dw || !workspace) return wfail(GTTS_E_NULL, "gtts_conv3x3_wgrad_tiled: null argument");
    if (B <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0 || cin % 64 || cout % 64)
        return wfail(GTTS_E_SHAPE, "gtts_conv3x3_wgrad_tiled: cin and cout must be multiples of 64 (got %d, %d)", cin, cout);
    if ((size_t)std::max(cin, cout) * H * W >= ((size_t)1 << 30)) return wfail(GTTS_E_SHAPE, "gtts_conv3x3_wgrad_tiled: tensor too large");
    Wgrad2Args a;
    wgrad2_geometry(B, cin, cout, H, W, a);
    const size_t need = gtts_conv3x3_wgrad_workspace_bytes(B, cin, cout, H, W);
    if (workspace_bytes < need) return wfail(GTTS_E_WORKSPACE, "gtts_conv3x3_wgrad_tiled: workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    const int tiles = (cin / 64) * (cout / 64);
    a.x = x; a.mask = mask; a.dy = dy; a.part = (float *)workspace;
    a.dbpart = db ? a.part + (size_t)a.nslice * tiles * (9 * 64 * 64) : nullptr;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(conv3x3_wgrad2_kernel, dim3((unsigned)(tiles * a.nslice)), dim3(256), 0, st, a);
    WCHK(hipGetLastError());
    const size_t total = (size_t)tiles * (9 * 64 * 64) + (db ? (size_t)cout : 0);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, a.part, a.dbpart, dw, db, cin, cout,
                       a.nslice);
    WCHK(hipGetLastError());
    return GTTS_OK;
}
