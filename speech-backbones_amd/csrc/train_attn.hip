// train_attn.hip -- LinearAttention's core and the Rezero residual for the training hot path.
//
//   forward   k~ = softmax_n(k);  ctx[d][e] = sum_n k~[d,n] v[e,n];  out[e,n] = sum_d ctx[d][e] q[d,n]
//             (Grad-TTS/model/diffusion.py:90-100; q, k, v are the three 128-channel thirds of to_qkv's output, 4 heads x 32)
//   backward  dq[d,n] = sum_e ctx[d][e] dout[e,n]        dctx[d][e] = sum_n q[d,n] dout[e,n]
//             dv[e,n] = sum_d k~[d,n] dctx[d][e]         dk[d,n] = k~[d,n] (sum_e dctx[d][e] v[e,n] - r_d),
//             r_d = sum_n k~[d,n] dk~[d,n] = sum_e dctx[d][e] ctx[d][e]   (the softmax row term needs no pass over the pixels)
//   Rezero    y = f * g + x  (diffusion.py:40-46, :103-108) with df = dy * g, dg = sum(dy * f), dx = dy
//
// The two 1x1 convolutions around the core run on the CONV_P1 MFMA kernel (train.hip: gtts_conv1x1_masked / _wgrad).  The core
// itself is 32 x 32 per head and pixel -- 1 % of the network's FLOPs -- and bound by HBM: fp32 FMAs on LDS tiles, every
// tensor read once per kernel.  Reductions over pixels are sliced across workgroups (online softmax: each slice carries its
// row maxima and sums) and combined in a fixed order: deterministic, no atomics.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>

#include "../../include/gradtts_abi.h"
#include "common.h"
#include "kernels.h"

namespace gtts {

constexpr int AT_D = 32;          // channels per head
constexpr int AT_H = 4;           // heads
constexpr int AT_TILE = 64;       // pixels per LDS tile
constexpr int AT_TPS = 8;         // tiles per slice (512 pixels per workgroup)
constexpr int AT_REC = AT_D * AT_D + 2 * AT_D;      // floats of one slice record: ctx partial, row max, row sum

// grid (slices, heads, B).  a = k (softmax != 0) or q; b = v or dout: rec[d][e] = sum_n f(a[d,n]) b[e,n] over the slice's pixels,
// f = exp(a - m_d) with the slice's own row maximum m_d (softmax) or the identity.
__global__ __launch_bounds__(256) void attn_outer_kernel(const float *__restrict__ a, const float *__restrict__ b, float *__restrict__ rec,
                                                         int N, size_t a_bstride, size_t b_bstride, int nslice, int softmax) {
    __shared__ float s_a[AT_D][AT_TILE + 1], s_b[AT_D][AT_TILE + 1];
    const int tid = threadIdx.x, d = tid >> 3, j = tid & 7;
    const int slice = blockIdx.x, h = blockIdx.y, bi = blockIdx.z;
    const float *pa = a + (size_t)bi * a_bstride + (size_t)h * AT_D * N;
    const float *pb = b + (size_t)bi * b_bstride + (size_t)h * AT_D * N;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, s = 0.f;
    const int n_begin = slice * AT_TPS * AT_TILE;
    for (int t = 0; t < AT_TPS; ++t) {
        const int n0 = n_begin + t * AT_TILE;
        if (n0 >= N) break;
        __syncthreads();
        // thread (row d, octet j) stages 8 consecutive pixels of row d of both tiles
        float va[8], vb[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int n = n0 + 8 * j + i;
            va[i] = n < N ? pa[(size_t)d * N + n] : (softmax ? -INFINITY : 0.f);
            vb[i] = n < N ? pb[(size_t)d * N + n] : 0.f;
        }
        if (softmax) {
            float tm = va[0];
#pragma unroll
            for (int i = 1; i < 8; ++i) tm = fmaxf(tm, va[i]);
            tm = fmaxf(tm, __shfl_xor(tm, 1, 64));
            tm = fmaxf(tm, __shfl_xor(tm, 2, 64));
            tm = fmaxf(tm, __shfl_xor(tm, 4, 64));
            const float mn = fmaxf(m, tm);
            const float scale = __expf(m - mn);           // (first tile: exp(-inf) = 0)
            float ts = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                va[i] = __expf(va[i] - mn);               // (padding: exp(-inf) = 0)
                ts += va[i];
            }
            ts += __shfl_xor(ts, 1, 64);
            ts += __shfl_xor(ts, 2, 64);
            ts += __shfl_xor(ts, 4, 64);
            s = s * scale + ts;
            m = mn;
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] *= scale;
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            s_a[d][8 * j + i] = va[i];
            s_b[d][8 * j + i] = vb[i];
        }
        __syncthreads();
        // thread (d, j) owns ctx[d][4j .. 4j+3]
#pragma unroll 8
        for (int n = 0; n < AT_TILE; ++n) {
            const float p = s_a[d][n];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = fmaf(p, s_b[4 * j + i][n], acc[i]);
        }
    }
    float *r = rec + (((size_t)bi * AT_H + h) * nslice + slice) * AT_REC;
#pragma unroll
    for (int i = 0; i < 4; ++i) r[d * AT_D + 4 * j + i] = acc[i];
    if (j == 0) {
        r[AT_D * AT_D + d] = m;
        r[AT_D * AT_D + AT_D + d] = s;
    }
}

// grid (B * heads), 1024 threads (d, e): combine the slice records in slice order.  softmax: ctx = sum_s rec_s e^{m_s - M} / S,
// stat[d] = (M, 1 / S).  Otherwise ctx = sum_s rec_s.  With ctx_fwd != nullptr also r[d] = sum_e ctx[d][e] ctx_fwd[d][e].
__global__ __launch_bounds__(1024) void attn_combine_kernel(const float *__restrict__ rec, float *__restrict__ ctx, float *__restrict__ stat,
                                                            const float *__restrict__ ctx_fwd, float *__restrict__ rdot, int nslice,
                                                            int softmax) {
    __shared__ float s_p[AT_D][AT_D + 1];
    const int tid = threadIdx.x, d = tid >> 5, e = tid & 31;
    const float *r = rec + (size_t)blockIdx.x * nslice * AT_REC;
    float M = -INFINITY, S = 1.f, v = 0.f;
    if (softmax) {
        for (int sl = 0; sl < nslice; ++sl) M = fmaxf(M, r[(size_t)sl * AT_REC + AT_D * AT_D + d]);
        S = 0.f;
        for (int sl = 0; sl < nslice; ++sl) {
            const float w = __expf(r[(size_t)sl * AT_REC + AT_D * AT_D + d] - M);
            S += r[(size_t)sl * AT_REC + AT_D * AT_D + AT_D + d] * w;
            v += r[(size_t)sl * AT_REC + d * AT_D + e] * w;
        }
        v /= S;
        if (e == 0) {
            stat[((size_t)blockIdx.x * AT_D + d) * 2] = M;
            stat[((size_t)blockIdx.x * AT_D + d) * 2 + 1] = 1.f / S;
        }
    } else {
        for (int sl = 0; sl < nslice; ++sl) v += r[(size_t)sl * AT_REC + d * AT_D + e];
    }
    ctx[(size_t)blockIdx.x * AT_D * AT_D + tid] = v;
    if (ctx_fwd) {
        s_p[d][e] = v * ctx_fwd[(size_t)blockIdx.x * AT_D * AT_D + tid];
        __syncthreads();
        if (e == 0) {
            float t = 0.f;
            for (int i = 0; i < AT_D; ++i) t += s_p[d][i];
            rdot[(size_t)blockIdx.x * AT_D + d] = t;
        }
    }
}

// grid (ceil(N / 256), heads, B): out[e,n] = sum_d ctx[d][e] q[d,n]; one pixel per thread
__global__ __launch_bounds__(256) void attn_apply_kernel(const float *__restrict__ q, const float *__restrict__ ctx, float *__restrict__ out,
                                                         int N, size_t q_bstride, size_t o_bstride) {
    __shared__ float s_c[AT_D * AT_D];
    const int tid = threadIdx.x, h = blockIdx.y, bi = blockIdx.z;
    const int n = blockIdx.x * 256 + tid;
    for (int i = tid; i < AT_D * AT_D; i += 256) s_c[i] = ctx[((size_t)bi * AT_H + h) * AT_D * AT_D + i];
    __syncthreads();
    if (n >= N) return;
    const float *pq = q + (size_t)bi * q_bstride + (size_t)h * AT_D * N + n;
    float acc[AT_D];
#pragma unroll
    for (int e = 0; e < AT_D; ++e) acc[e] = 0.f;
#pragma unroll 4
    for (int d = 0; d < AT_D; ++d) {
        const float qv = pq[(size_t)d * N];
#pragma unroll
        for (int e = 0; e < AT_D; ++e) acc[e] = fmaf(s_c[d * AT_D + e], qv, acc[e]);
    }
    float *po = out + (size_t)bi * o_bstride + (size_t)h * AT_D * N + n;
#pragma unroll
    for (int e = 0; e < AT_D; ++e) po[(size_t)e * N] = acc[e];
}

// grid (ceil(N / 256), heads, B): dq, dk, dv of one pixel per thread (formulas in the header); qkv and dqkv are [B][384][N]
__global__ __launch_bounds__(256) void attn_bwd_pixel_kernel(const float *__restrict__ qkv, const float *__restrict__ dout,
                                                             const float *__restrict__ ctx, const float *__restrict__ dctx,
                                                             const float *__restrict__ stat, const float *__restrict__ rdot,
                                                             float *__restrict__ dqkv, int N) {
    __shared__ float s_c[AT_D * AT_D], s_dc[AT_D * AT_D], s_m[AT_D], s_is[AT_D], s_r[AT_D];
    const int tid = threadIdx.x, h = blockIdx.y, bi = blockIdx.z;
    const int n = blockIdx.x * 256 + tid;
    const size_t bh = (size_t)bi * AT_H + h;
    for (int i = tid; i < AT_D * AT_D; i += 256) {
        s_c[i] = ctx[bh * AT_D * AT_D + i];
        s_dc[i] = dctx[bh * AT_D * AT_D + i];               // [d][e]
    }
    if (tid < AT_D) {
        s_m[tid] = stat[(bh * AT_D + tid) * 2];
        s_is[tid] = stat[(bh * AT_D + tid) * 2 + 1];
        s_r[tid] = rdot[bh * AT_D + tid];
    }
    __syncthreads();
    if (n >= N) return;
    const size_t C3 = (size_t)3 * AT_H * AT_D;
    const float *pq = qkv + ((size_t)bi * C3 + (size_t)h * AT_D) * N + n;
    const float *pk = pq + (size_t)AT_H * AT_D * N, *pv = pk + (size_t)AT_H * AT_D * N;
    const float *pdo = dout + ((size_t)bi * AT_H * AT_D + (size_t)h * AT_D) * N + n;
    float *dq = dqkv + ((size_t)bi * C3 + (size_t)h * AT_D) * N + n;
    float *dk = dq + (size_t)AT_H * AT_D * N, *dv = dk + (size_t)AT_H * AT_D * N;
    float acc[AT_D];
    // dq[d] = sum_e ctx[d][e] dout[e]
#pragma unroll
    for (int d = 0; d < AT_D; ++d) acc[d] = 0.f;
#pragma unroll 4
    for (int e = 0; e < AT_D; ++e) {
        const float g = pdo[(size_t)e * N];
#pragma unroll
        for (int d = 0; d < AT_D; ++d) acc[d] = fmaf(s_c[d * AT_D + e], g, acc[d]);
    }
#pragma unroll
    for (int d = 0; d < AT_D; ++d) dq[(size_t)d * N] = acc[d];
    // t[d] = sum_e dctx[d][e] v[e];  dk[d] = k~[d] (t[d] - r_d)
#pragma unroll
    for (int d = 0; d < AT_D; ++d) acc[d] = 0.f;
#pragma unroll 4
    for (int e = 0; e < AT_D; ++e) {
        const float vv = pv[(size_t)e * N];
#pragma unroll
        for (int d = 0; d < AT_D; ++d) acc[d] = fmaf(s_dc[d * AT_D + e], vv, acc[d]);
    }
    float kt[AT_D];
#pragma unroll
    for (int d = 0; d < AT_D; ++d) {
        kt[d] = __expf(pk[(size_t)d * N] - s_m[d]) * s_is[d];
        dk[(size_t)d * N] = kt[d] * (acc[d] - s_r[d]);
    }
    // dv[e] = sum_d k~[d] dctx[d][e]
#pragma unroll
    for (int e = 0; e < AT_D; ++e) acc[e] = 0.f;
#pragma unroll
    for (int d = 0; d < AT_D; ++d) {
#pragma unroll
        for (int e = 0; e < AT_D; ++e) acc[e] = fmaf(kt[d], s_dc[d * AT_D + e], acc[e]);
    }
#pragma unroll
    for (int e = 0; e < AT_D; ++e) dv[(size_t)e * N] = acc[e];
}

// ---------------------------------------------------------------------------------------------------- Rezero residual
// y = f * g + x (g: one device scalar)
__global__ __launch_bounds__(256) void rezero_fwd_kernel(const float4 *__restrict__ f, const float4 *__restrict__ x, const float *__restrict__ g,
                                                         float4 *__restrict__ y, size_t n4) {
    const float gv = g[0];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 a = f[i], b = x[i];
        y[i] = make_float4(fmaf(a.x, gv, b.x), fmaf(a.y, gv, b.y), fmaf(a.z, gv, b.z), fmaf(a.w, gv, b.w));
    }
}
// df = dy * g; part[block] = sum over the block's elements of dy * f (fp64 inside the block, fixed order)
__global__ __launch_bounds__(256) void rezero_bwd_kernel(const float4 *__restrict__ dy, const float4 *__restrict__ f, const float *__restrict__ g,
                                                         float4 *__restrict__ df, double *__restrict__ part, size_t n4) {
    __shared__ double s_a[256];
    const float gv = g[0];
    double acc = 0.0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 a = dy[i], b = f[i];
        df[i] = make_float4(a.x * gv, a.y * gv, a.z * gv, a.w * gv);
        acc += (double)(a.x * b.x + a.y * b.y) + (double)(a.z * b.z + a.w * b.w);
    }
    s_a[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s_a[threadIdx.x] += s_a[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = s_a[0];
}
// one workgroup: fixed-order sum of the block partials (strided per thread, then a tree)
__global__ __launch_bounds__(256) void rezero_bwd_finish_kernel(const double *__restrict__ part, int nblk, float *__restrict__ dg) {
    __shared__ double s_a[256];
    double t = 0.0;
    for (int i = threadIdx.x; i < nblk; i += 256) t += part[i];
    s_a[threadIdx.x] = t;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s_a[threadIdx.x] += s_a[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) dg[0] = (float)s_a[0];
}

}  // namespace gtts

using namespace gtts;

static int afail(int code, const char *fmt, ...) {       // text goes to gtts_last_error() (plan.hip)
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return set_error(code, buf);
}
#define ACHK(expr)                                                                                                \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) return afail(GTTS_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

static int attn_nslice(int N) { return (N + AT_TPS * AT_TILE - 1) / (AT_TPS * AT_TILE); }

// floats of scratch either direction needs: the slice records
extern "C" size_t gtts_attn_train_scratch_floats(int B, int N) {
    if (B <= 0 || N <= 0) return 0;
    return (size_t)B * AT_H * attn_nslice(N) * AT_REC;
}

// qkv [B][384][N] (to_qkv's output: q | k | v, each 4 heads x 32) -> out [B][128][N], ctx [B][4][32][32], stat [B][4][32][2]
extern "C" int gtts_attn_train_forward(const float *qkv, float *out, float *ctx, float *stat, float *scratch, int B, int N,
                                       gtts_stream_t stream) {
    if (!qkv || !out || !ctx || !stat || !scratch) return afail(GTTS_E_NULL, "gtts_attn_train_forward: null argument");
    if (B <= 0 || N <= 0) return afail(GTTS_E_SHAPE, "gtts_attn_train_forward: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const int ns = attn_nslice(N);
    const size_t C = (size_t)AT_H * AT_D, bs = 3 * C * N;
    hipLaunchKernelGGL(attn_outer_kernel, dim3(ns, AT_H, B), dim3(256), 0, st, qkv + C * N, qkv + 2 * C * N, scratch, N, bs, bs, ns, 1);
    ACHK(hipGetLastError());
    hipLaunchKernelGGL(attn_combine_kernel, dim3(B * AT_H), dim3(1024), 0, st, scratch, ctx, stat, (const float *)nullptr, (float *)nullptr, ns, 1);
    ACHK(hipGetLastError());
    hipLaunchKernelGGL(attn_apply_kernel, dim3((N + 255) / 256, AT_H, B), dim3(256), 0, st, qkv, ctx, out, N, bs, C * N);
    ACHK(hipGetLastError());
    return GTTS_OK;
}

// dout [B][128][N] -> dqkv [B][384][N]; dctx [B][4][32][32] and rdot [B][4][32] are scratch outputs
extern "C" int gtts_attn_train_backward(const float *qkv, const float *dout, const float *ctx, const float *stat, float *dqkv,
                                        float *dctx, float *rdot, float *scratch, int B, int N, gtts_stream_t stream) {
    if (!qkv || !dout || !ctx || !stat || !dqkv || !dctx || !rdot || !scratch) return afail(GTTS_E_NULL, "gtts_attn_train_backward: null argument");
    if (B <= 0 || N <= 0) return afail(GTTS_E_SHAPE, "gtts_attn_train_backward: bad shape");
    hipStream_t st = (hipStream_t)stream;
    const int ns = attn_nslice(N);
    const size_t C = (size_t)AT_H * AT_D, bs = 3 * C * N;
    hipLaunchKernelGGL(attn_outer_kernel, dim3(ns, AT_H, B), dim3(256), 0, st, qkv, dout, scratch, N, bs, C * N, ns, 0);
    ACHK(hipGetLastError());
    hipLaunchKernelGGL(attn_combine_kernel, dim3(B * AT_H), dim3(1024), 0, st, scratch, dctx, (float *)nullptr, ctx, rdot, ns, 0);
    ACHK(hipGetLastError());
    hipLaunchKernelGGL(attn_bwd_pixel_kernel, dim3((N + 255) / 256, AT_H, B), dim3(256), 0, st, qkv, dout, ctx, dctx, stat, rdot, dqkv, N);
    ACHK(hipGetLastError());
    return GTTS_OK;
}

static int ew_blocks(size_t n4) { return (int)std::min<size_t>((n4 + 255) / 256, 2048); }

// y = f * g + x over n floats (n % 4 == 0); g is a device scalar
extern "C" int gtts_rezero_forward(const float *f, const float *x, const float *g, float *y, size_t n, gtts_stream_t stream) {
    if (!f || !x || !g || !y) return afail(GTTS_E_NULL, "gtts_rezero_forward: null argument");
    if (n == 0 || n % 4) return afail(GTTS_E_SHAPE, "gtts_rezero_forward: element count must be a positive multiple of 4");
    hipLaunchKernelGGL(rezero_fwd_kernel, dim3(ew_blocks(n / 4)), dim3(256), 0, (hipStream_t)stream, (const float4 *)f, (const float4 *)x, g,
                       (float4 *)y, n / 4);
    ACHK(hipGetLastError());
    return GTTS_OK;
}

extern "C" size_t gtts_rezero_scratch_bytes(size_t n) { return (size_t)ew_blocks(n / 4) * sizeof(double); }

// df = dy * g, dg = sum(dy * f)
extern "C" int gtts_rezero_backward(const float *dy, const float *f, const float *g, float *df, float *dg, void *scratch, size_t n,
                                    gtts_stream_t stream) {
    if (!dy || !f || !g || !df || !dg || !scratch) return afail(GTTS_E_NULL, "gtts_rezero_backward: null argument");
    if (n == 0 || n % 4) return afail(GTTS_E_SHAPE, "gtts_rezero_backward: element count must be a positive multiple of 4");
    const int nb = ew_blocks(n / 4);
    hipLaunchKernelGGL(rezero_bwd_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, (const float4 *)dy, (const float4 *)f, g, (float4 *)df,
                       (double *)scratch, n / 4);
    ACHK(hipGetLastError());
    hipLaunchKernelGGL(rezero_bwd_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double *)scratch, nb, dg);
    ACHK(hipGetLastError());
    return GTTS_OK;
}
