// train_norm.hip -- Block's GroupNorm + Mish + mask for the training hot path (second stage).
//
//   forward   out = Mish(GroupNorm_8(y)) * mask                     Grad-TTS/model/diffusion.py:53-58 (Block), :13-15 (Mish)
//   backward  dy, dgamma, dbeta from d out                          (what autograd derives for those lines)
//
// GroupNorm statistics span the whole H x W plane of the group's channels, masked frames included, exactly like
// torch.nn.GroupNorm on the padded batch (eps 1e-5, biased variance).  Everything here is bound by HBM:
// the forward reads y twice (statistics, apply) and writes out once; the backward reads (d out, y) twice and writes dy once --
// z = x_hat * gamma + beta and Mish'(z) are recomputed instead of stored (two transcendentals per element against 4 B of
// traffic each way).  Reductions are fixed-order (deterministic): per-thread partial sums in fp32 over strided elements,
// combined across the workgroup in fp64.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/gradtts_abi.h"
#include "common.h"
#include "kernels.h"

namespace gtts {

// fixed-order workgroup sum of two doubles (256 threads); result valid in every thread
__device__ __forceinline__ void block_sum2(double &a, double &b, double *s_a, double *s_b) {
    const int tid = threadIdx.x;
    s_a[tid] = a;
    s_b[tid] = b;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
            s_a[tid] += s_a[tid + o];
            s_b[tid] += s_b[tid + o];
        }
        __syncthreads();
    }
    a = s_a[0];
    b = s_b[0];
    __syncthreads();
}

// grid (B * groups, GN_SPLIT): partial (sum, sum of squares) of one slice of a (sample, group) region -- a contiguous run of
// (C / groups) * HW floats -- as doubles; 128 regions alone would leave half of the chip idle
constexpr int GN_SPLIT = 16;
__global__ __launch_bounds__(256) void gn_stats_kernel(const float *__restrict__ y, double *__restrict__ partial, size_t n) {
    __shared__ double s_a[256], s_b[256];
    const size_t per = (n + GN_SPLIT - 1) / GN_SPLIT, i0 = blockIdx.y * per, i1 = i0 + per < n ? i0 + per : n;
    const float *p = y + (size_t)blockIdx.x * n;
    float s1 = 0.f, s2 = 0.f;
    double d1 = 0.0, d2 = 0.0;
    int k = 0;
    for (size_t i = i0 + threadIdx.x; i < i1; i += 256) {
        const float v = p[i];
        s1 += v;
        s2 += v * v;
        if (++k == 64) {        // bound the fp32 run length: flush into fp64 every 64 elements
            d1 += (double)s1; d2 += (double)s2;
            s1 = 0.f; s2 = 0.f; k = 0;
        }
    }
    d1 += (double)s1;
    d2 += (double)s2;
    block_sum2(d1, d2, s_a, s_b);
    if (threadIdx.x == 0) {
        partial[2 * ((size_t)blockIdx.x * GN_SPLIT + blockIdx.y)] = d1;
        partial[2 * ((size_t)blockIdx.x * GN_SPLIT + blockIdx.y) + 1] = d2;
    }
}
// one thread per (sample, group): mean and 1 / sqrt(var + eps) from the GN_SPLIT partials (fixed order)
__global__ void gn_stats_finish_kernel(const double *__restrict__ partial, float *__restrict__ stats, int nbg, double inv_n, float eps) {
    const int bg = blockIdx.x * 256 + threadIdx.x;
    if (bg >= nbg) return;
    double d1 = 0.0, d2 = 0.0;
    for (int s = 0; s < GN_SPLIT; ++s) {
        d1 += partial[2 * ((size_t)bg * GN_SPLIT + s)];
        d2 += partial[2 * ((size_t)bg * GN_SPLIT + s) + 1];
    }
    const double mean = d1 * inv_n;
    double var = d2 * inv_n - mean * mean;
    if (var < 0.0) var = 0.0;
    stats[2 * bg] = (float)mean;
    stats[2 * bg + 1] = (float)(1.0 / sqrt(var + (double)eps));
}

// Mish and its derivative from one exponential: e = exp(z), n = e (e + 2), tanh(softplus(z)) = n / (n + 2),
// sigmoid(z) = e / (1 + e);  Mish'(z) = tanh(softplus z) + z sigmoid(z) (1 - tanh(softplus z)^2)
__device__ __forceinline__ void mish_and_grad(float z, float &m, float &dm) {
    const float e = __expf(fminf(z, 40.0f));
    const float n = e * (e + 2.0f);
    const float t = n * __builtin_amdgcn_rcpf(n + 2.0f);
    const float sg = e * __builtin_amdgcn_rcpf(1.0f + e);
    m = z * t;
    dm = t + z * sg * (1.0f - t * t);
}

// grid (B * C, ceil(HW / 1024)): out = Mish(x_hat * gamma + beta) * mask[b, w] (+ tb[b, c])
__global__ __launch_bounds__(256) void gn_mish_fwd_kernel(const float *__restrict__ y, const float *__restrict__ gamma,
                                                           const float *__restrict__ beta, const float *__restrict__ mask,
                                                           const float *__restrict__ stats, const float *__restrict__ tb,
                                                           float *__restrict__ out, int C, int HW, int W, int cpg) {
    const int bc = blockIdx.x, b = bc / C, c = bc - b * C, g = c / cpg;
    const float mean = stats[2 * (b * (C / cpg) + g)], rstd = stats[2 * (b * (C / cpg) + g) + 1];
    const float sc = rstd * gamma[c], sh = beta[c] - mean * sc;
    const float tbv = tb ? tb[bc] : 0.f;          // ResnetBlock's time bias, added AFTER the mask (diffusion.py:75-76)
    const float *p = y + (size_t)bc * HW;
    float *q = out + (size_t)bc * HW;
    const float *mrow = mask + (size_t)b * W;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = blockIdx.y * 1024 + k * 256 + threadIdx.x;
        if (i < HW) q[i] = mish_f(p[i] * sc + sh) * mrow[i % W] + tbv;
    }
}

// grid (B * C): ws[bc] = (sum dz, sum dz x_hat) over the plane, dz = d out * mask * Mish'(z)
__global__ __launch_bounds__(256) void gn_mish_bwd_reduce_kernel(const float *__restrict__ dout, const float *__restrict__ y,
                                                                  const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                  const float *__restrict__ mask, const float *__restrict__ stats,
                                                                  float *__restrict__ ws, float *__restrict__ dtb, int C, int HW, int W,
                                                                  int cpg) {
    __shared__ double s_a[256], s_b[256];
    const int bc = blockIdx.x, b = bc / C, c = bc - b * C, g = c / cpg;
    const float mean = stats[2 * (b * (C / cpg) + g)], rstd = stats[2 * (b * (C / cpg) + g) + 1];
    const float ga = gamma[c], be = beta[c];
    const float *p = y + (size_t)bc * HW, *d = dout + (size_t)bc * HW;
    const float *mrow = mask + (size_t)b * W;
    float s1 = 0.f, s2 = 0.f, s3 = 0.f;
    double d1 = 0.0, d2 = 0.0, d3 = 0.0;
    int k = 0;
    for (int i = threadIdx.x; i < HW; i += 256) {
        const float xh = (p[i] - mean) * rstd;
        float m, dm;
        mish_and_grad(xh * ga + be, m, dm);
        const float dz = d[i] * mrow[i % W] * dm;
        s1 += dz;
        s2 += dz * xh;
        s3 += d[i];
        if (++k == 64) {
            d1 += (double)s1; d2 += (double)s2; d3 += (double)s3;
            s1 = 0.f; s2 = 0.f; s3 = 0.f; k = 0;
        }
    }
    d1 += (double)s1;
    d2 += (double)s2;
    d3 += (double)s3;
    block_sum2(d1, d2, s_a, s_b);
    if (threadIdx.x == 0) {
        ws[2 * bc] = (float)d1;
        ws[2 * bc + 1] = (float)d2;
    }
    if (dtb) {          // gradient of the time bias: the plain sum of d out over the plane
        double z = 0.0;
        block_sum2(d3, z, s_a, s_b);
        if (threadIdx.x == 0) dtb[bc] = (float)d3;
    }
}

// one workgroup: dgamma[c] = sum_b ws[b,c].1, dbeta[c] = sum_b ws[b,c].0 (fixed order over b);
// coef[b,g] = (sum_{c in g} gamma_c ws[b,c].0, sum_{c in g} gamma_c ws[b,c].1) / N
__global__ __launch_bounds__(256) void gn_mish_bwd_finish_kernel(const float *__restrict__ ws, const float *__restrict__ gamma,
                                                                  float *__restrict__ dgamma, float *__restrict__ dbeta,
                                                                  float *__restrict__ coef, int B, int C, int cpg, double inv_n) {
    for (int c = threadIdx.x; c < C; c += 256) {
        double a = 0.0, g = 0.0;
        for (int b = 0; b < B; ++b) {
            a += (double)ws[2 * (b * C + c)];
            g += (double)ws[2 * (b * C + c) + 1];
        }
        dbeta[c] = (float)a;
        dgamma[c] = (float)g;
    }
    const int G = C / cpg;
    for (int bg = threadIdx.x; bg < B * G; bg += 256) {
        const int b = bg / G, g = bg - b * G;
        double s1 = 0.0, s2 = 0.0;
        for (int c = g * cpg; c < (g + 1) * cpg; ++c) {
            s1 += (double)gamma[c] * (double)ws[2 * (b * C + c)];
            s2 += (double)gamma[c] * (double)ws[2 * (b * C + c) + 1];
        }
        coef[2 * bg] = (float)(s1 * inv_n);
        coef[2 * bg + 1] = (float)(s2 * inv_n);
    }
}

// grid (B * C, ceil(HW / 1024)): dy = rstd (dz gamma - coef1 - x_hat coef2)
__global__ __launch_bounds__(256) void gn_mish_bwd_apply_kernel(const float *__restrict__ dout, const float *__restrict__ y,
                                                                 const float *__restrict__ gamma, const float *__restrict__ beta,
                                                                 const float *__restrict__ mask, const float *__restrict__ stats,
                                                                 const float *__restrict__ coef, float *__restrict__ dy, int C, int HW,
                                                                 int W, int cpg) {
    const int bc = blockIdx.x, b = bc / C, c = bc - b * C, g = c / cpg, G = C / cpg;
    const float mean = stats[2 * (b * G + g)], rstd = stats[2 * (b * G + g) + 1];
    const float c1 = coef[2 * (b * G + g)], c2 = coef[2 * (b * G + g) + 1];
    const float ga = gamma[c], be = beta[c];
    const float *p = y + (size_t)bc * HW, *d = dout + (size_t)bc * HW;
    float *q = dy + (size_t)bc * HW;
    const float *mrow = mask + (size_t)b * W;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = blockIdx.y * 1024 + k * 256 + threadIdx.x;
        if (i < HW) {
            const float xh = (p[i] - mean) * rstd;
            float m, dm;
            mish_and_grad(xh * ga + be, m, dm);
            const float dz = d[i] * mrow[i % W] * dm;
            q[i] = rstd * (dz * ga - c1 - xh * c2);
        }
    }
}

}  // namespace gtts

using namespace gtts;

static int nfail(int code, const char *fmt, ...) {       // text goes to gtts_last_error() (plan.hip)
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return set_error(code, buf);
}
#define NCHK(expr)                                                                                                \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) return nfail(GTTS_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

static int norm_shape_ok(int B, int C, int H, int W, int groups) {
    return B > 0 && C > 0 && H > 0 && W > 0 && groups > 0 && C % groups == 0 && (long)H * W < (1l << 30);
}

// tb (nullable): [B][C] bias added after the mask (ResnetBlock's time embedding term, diffusion.py:75-76)
extern "C" int gtts_gn_mish_forward_tb(const float *y, const float *gamma, const float *beta, const float *mask, const float *tb,
                                       float *out, float *stats, int B, int C, int H, int W, int groups, float eps,
                                       gtts_stream_t stream) {
    if (!y || !gamma || !beta || !mask || !out || !stats) return nfail(GTTS_E_NULL, "gtts_gn_mish_forward: null argument");
    if (!norm_shape_ok(B, C, H, W, groups)) return nfail(GTTS_E_SHAPE, "gtts_gn_mish_forward: bad shape B=%d C=%d H=%d W=%d groups=%d", B, C, H, W, groups);
    const int HW = H * W, cpg = C / groups;
    hipStream_t st = (hipStream_t)stream;
    double *partial = reinterpret_cast<double *>(stats + (size_t)B * groups * 2);      // behind the (mean, rstd) pairs (8-byte aligned)
    hipLaunchKernelGGL(gn_stats_kernel, dim3(B * groups, GN_SPLIT), dim3(256), 0, st, y, partial, (size_t)cpg * HW);
    NCHK(hipGetLastError());
    hipLaunchKernelGGL(gn_stats_finish_kernel, dim3((B * groups + 255) / 256), dim3(256), 0, st, partial, stats, B * groups,
                       1.0 / ((double)cpg * HW), eps);
    NCHK(hipGetLastError());
    hipLaunchKernelGGL(gn_mish_fwd_kernel, dim3(B * C, (HW + 1023) / 1024), dim3(256), 0, st, y, gamma, beta, mask, stats, tb, out, C, HW, W, cpg);
    NCHK(hipGetLastError());
    return GTTS_OK;
}

extern "C" int gtts_gn_mish_forward(const float *y, const float *gamma, const float *beta, const float *mask, float *out,
                                    float *stats, int B, int C, int H, int W, int groups, float eps, gtts_stream_t stream) {
    return gtts_gn_mish_forward_tb(y, gamma, beta, mask, nullptr, out, stats, B, C, H, W, groups, eps, stream);
}

extern "C" size_t gtts_gn_mish_stats_floats(int B, int groups) {
    return B > 0 && groups > 0 ? (size_t)B * groups * 2 + (size_t)B * groups * GN_SPLIT * 4 : 0;
}

extern "C" size_t gtts_gn_mish_scratch_bytes(int B, int C) { return B > 0 && C > 0 ? (size_t)B * C * 2 * 4 * 2 : 0; }

// dtb (nullable): [B][C] gradient of the time bias
extern "C" int gtts_gn_mish_backward_tb(const float *dout, const float *y, const float *gamma, const float *beta, const float *mask,
                                        const float *stats, float *dy, float *dgamma, float *dbeta, float *dtb, void *scratch, int B,
                                        int C, int H, int W, int groups, gtts_stream_t stream) {
    if (!dout || !y || !gamma || !beta || !mask || !stats || !dy || !dgamma || !dbeta || !scratch)
        return nfail(GTTS_E_NULL, "gtts_gn_mish_backward: null argument");
    if (!norm_shape_ok(B, C, H, W, groups)) return nfail(GTTS_E_SHAPE, "gtts_gn_mish_backward: bad shape B=%d C=%d H=%d W=%d groups=%d", B, C, H, W, groups);
    const int HW = H * W, cpg = C / groups;
    hipStream_t st = (hipStream_t)stream;
    float *ws = (float *)scratch;                 // [B][C][2]
    float *coef = ws + (size_t)B * C * 2;         // [B][groups][2]  (fits: groups <= C)
    hipLaunchKernelGGL(gn_mish_bwd_reduce_kernel, dim3(B * C), dim3(256), 0, st, dout, y, gamma, beta, mask, stats, ws, dtb, C, HW, W, cpg);
    NCHK(hipGetLastError());
    hipLaunchKernelGGL(gn_mish_bwd_finish_kernel, dim3(1), dim3(256), 0, st, ws, gamma, dgamma, dbeta, coef, B, C, cpg,
                       1.0 / ((double)cpg * HW));
    NCHK(hipGetLastError());
    hipLaunchKernelGGL(gn_mish_bwd_apply_kernel, dim3(B * C, (HW + 1023) / 1024), dim3(256), 0, st, dout, y, gamma, beta, mask, stats, coef,
                       dy, C, HW, W, cpg);
    NCHK(hipGetLastError());
    return GTTS_OK;
}

extern "C" int gtts_gn_mish_backward(const float *dout, const float *y, const float *gamma, const float *beta, const float *mask,
                                     const float *stats, float *dy, float *dgamma, float *dbeta, void *scratch, int B, int C, int H,
                                     int W, int groups, gtts_stream_t stream) {
    return gtts_gn_mish_backward_tb(dout, y, gamma, beta, mask, stats, dy, dgamma, dbeta, nullptr, scratch, B, C, H, W, groups, stream);
}
