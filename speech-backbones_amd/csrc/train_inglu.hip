// train_inglu.hip -- RefBlock's InstanceNorm2d + GLU for the DiffVC decoder's training step (round 6).
//
//   forward   out[b, c] = IN(y[b, c]) * sigmoid(IN(y[b, C + c]))      DiffVC/model/modules.py:128-157 (the `_conv_in_glu` blocks:
//             IN(v) = (v - mean) rstd gamma + beta per (sample, channel) plane                InstanceNorm2d(affine=True) -> GLU(dim=1))
//   backward  dy [B, 2C, H, W], dgamma [2C], dbeta [2C] from d out    (what autograd derives for those two modules)
//
// InstanceNorm statistics span the whole H x W plane, padded frames included, biased variance, eps 1e-5 -- torch's modules on the
// padded batch.  One workgroup owns one (sample, channel pair): it reads the two planes (value half a = channel c, gate half
// g = channel C + c) twice -- statistics, then apply; the second read is an L2 hit at RefBlock's plane sizes (80 x T_ref) -- and
// nothing but y, d out and the four statistics travels between forward and backward: the normalised values, the sigmoid and the
// products are recomputed (three transcendentals per output element against 12 B of traffic).  Reductions: per-thread fp32
// partial sums over strided elements, combined across the workgroup in fp64 in a fixed order (deterministic); the parameter
// gradients are reduced over the batch by a second, tiny kernel in sample order.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/gradtts_abi.h"
#include "common.h"
#include "kernels.h"

namespace gtts {

__device__ __forceinline__ void block_sum4(double (&v)[4], double *s) {      // s: 4 x 256 doubles; result valid in every thread
    const int tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) s[k * 256 + tid] = v[k];
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (tid < o) {
#pragma unroll
            for (int k = 0; k < 4; ++k) s[k * 256 + tid] += s[k * 256 + tid + o];
        }
        __syncthreads();
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = s[k * 256];
    __syncthreads();
}

// (full-precision exp and division: the kernels are bound by HBM, and RefBlock's output conditions the whole score network)
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

// grid (C, B), 256 threads.  stats [B][C][4] = mean_a, rstd_a, mean_g, rstd_g.
__global__ __launch_bounds__(256) void in_glu_fwd_kernel(const float *__restrict__ y, const float *__restrict__ gamma,
                                                          const float *__restrict__ beta, float *__restrict__ out,
                                                          float *__restrict__ stats, int C, int HW, float eps) {
    __shared__ double s_red[4 * 256];
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float *ya = y + ((size_t)b * 2 * C + c) * HW, *yg = y + ((size_t)b * 2 * C + C + c) * HW;
    float s1a = 0.f, s2a = 0.f, s1g = 0.f, s2g = 0.f;
    for (int i = tid; i < HW; i += 256) {
        const float va = ya[i], vg = yg[i];
        s1a += va; s2a = fmaf(va, va, s2a);
        s1g += vg; s2g = fmaf(vg, vg, s2g);
    }
    double r[4] = {(double)s1a, (double)s2a, (double)s1g, (double)s2g};
    block_sum4(r, s_red);
    const double inv = 1.0 / (double)HW;
    const double ma = r[0] * inv, mg = r[2] * inv;
    double vva = fma(-ma, ma, r[1] * inv), vvg = fma(-mg, mg, r[3] * inv);
    if (vva < 0.0) vva = 0.0;
    if (vvg < 0.0) vvg = 0.0;
    const float mean_a = (float)ma, rstd_a = (float)(1.0 / sqrt(vva + (double)eps));
    const float mean_g = (float)mg, rstd_g = (float)(1.0 / sqrt(vvg + (double)eps));
    if (tid == 0) {
        float *st = stats + ((size_t)b * C + c) * 4;
        st[0] = mean_a; st[1] = rstd_a; st[2] = mean_g; st[3] = rstd_g;
    }
    const float ga = gamma[c] * rstd_a, ba = fmaf(-mean_a, ga, beta[c]);
    const float gg = gamma[C + c] * rstd_g, bg = fmaf(-mean_g, gg, beta[C + c]);
    float *o = out + ((size_t)b * C + c) * HW;
    for (int i = tid; i < HW; i += 256) o[i] = fmaf(ya[i], ga, ba) * sigmoid_f(fmaf(yg[i], gg, bg));
}

// grid (C, B), 256 threads.  pgrad [B][2C][2]: this sample's share of (dgamma, dbeta) of the two channels.
__global__ __launch_bounds__(256) void in_glu_bwd_kernel(const float *__restrict__ dout, const float *__restrict__ y,
                                                          const float *__restrict__ gamma, const float *__restrict__ beta,
                                                          const float *__restrict__ stats, float *__restrict__ dy,
                                                          float *__restrict__ pgrad, int C, int HW) {
    __shared__ double s_red[4 * 256];
    const int c = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const float *ya = y + ((size_t)b * 2 * C + c) * HW, *yg = y + ((size_t)b * 2 * C + C + c) * HW;
    const float *dob = dout + ((size_t)b * C + c) * HW;
    const float *st = stats + ((size_t)b * C + c) * 4;
    const float mean_a = st[0], rstd_a = st[1], mean_g = st[2], rstd_g = st[3];
    const float gma = gamma[c], bta = beta[c], gmg = gamma[C + c], btg = beta[C + c];
    // d a = d out sigma(g), d g = d out a sigma(g) (1 - sigma(g)); per plane: S1 = sum d, S2 = sum d x_hat
    float s1a = 0.f, s2a = 0.f, s1g = 0.f, s2g = 0.f;
    for (int i = tid; i < HW; i += 256) {
        const float xa = (ya[i] - mean_a) * rstd_a, xg = (yg[i] - mean_g) * rstd_g;
        const float a = fmaf(xa, gma, bta), sg = sigmoid_f(fmaf(xg, gmg, btg));
        const float d = dob[i];
        const float da = d * sg, dg = d * a * sg * (1.0f - sg);
        s1a += da; s2a = fmaf(da, xa, s2a);
        s1g += dg; s2g = fmaf(dg, xg, s2g);
    }
    double r[4] = {(double)s1a, (double)s2a, (double)s1g, (double)s2g};
    block_sum4(r, s_red);
    if (tid == 0) {
        float *pa = pgrad + ((size_t)b * 2 * C + c) * 2, *pg = pgrad + ((size_t)b * 2 * C + C + c) * 2;
        pa[0] = (float)r[1]; pa[1] = (float)r[0];          // dgamma share = sum d x_hat, dbeta share = sum d
        pg[0] = (float)r[3]; pg[1] = (float)r[2];
    }
    const double inv = 1.0 / (double)HW;
    const float m1a = (float)(r[0] * inv), m2a = (float)(r[1] * inv), m1g = (float)(r[2] * inv), m2g = (float)(r[3] * inv);
    const float ka = gma * rstd_a, kg = gmg * rstd_g;
    float *dya = dy + ((size_t)b * 2 * C + c) * HW, *dyg = dy + ((size_t)b * 2 * C + C + c) * HW;
    for (int i = tid; i < HW; i += 256) {
        const float xa = (ya[i] - mean_a) * rstd_a, xg = (yg[i] - mean_g) * rstd_g;
        const float a = fmaf(xa, gma, bta), sg = sigmoid_f(fmaf(xg, gmg, btg));
        const float d = dob[i];
        const float da = d * sg, dg = d * a * sg * (1.0f - sg);
        dya[i] = ka * (da - m1a - xa * m2a);
        dyg[i] = kg * (dg - m1g - xg * m2g);
    }
}

// dgamma[ch] = sum_b pgrad[b][ch][0], dbeta[ch] = sum_b pgrad[b][ch][1], in sample order
__global__ void in_glu_param_kernel(const float *__restrict__ pgrad, float *__restrict__ dgamma, float *__restrict__ dbeta, int B,
                                    int C2) {
    const int ch = blockIdx.x * blockDim.x + threadIdx.x;
    if (ch >= C2) return;
    double g = 0.0, bsum = 0.0;
    for (int b = 0; b < B; ++b) {
        g += (double)pgrad[((size_t)b * C2 + ch) * 2];
        bsum += (double)pgrad[((size_t)b * C2 + ch) * 2 + 1];
    }
    dgamma[ch] = (float)g;
    dbeta[ch] = (float)bsum;
}

static int ifail(int code, const char *fmt, ...) {       // text goes to gtts_last_error() (plan.hip)
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    return set_error(code, buf);
}

}  // namespace gtts

using namespace gtts;

extern "C" size_t gtts_in_glu_stats_floats(int B, int C) { return B > 0 && C > 0 ? (size_t)B * C * 4 : 0; }
extern "C" size_t gtts_in_glu_scratch_floats(int B, int C) { return B > 0 && C > 0 ? (size_t)B * 2 * C * 2 : 0; }

extern "C" int gtts_in_glu_forward(const float *y, const float *gamma, const float *beta, float *out, float *stats, int B, int C, int H,
                                   int W, float eps, gtts_stream_t stream) {
    if (!y || !gamma || !beta || !out || !stats) return ifail(GTTS_E_NULL, "gtts_in_glu_forward: null argument");
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || B > 65535 || (size_t)H * W >= ((size_t)1 << 31))
        return ifail(GTTS_E_SHAPE, "gtts_in_glu_forward: bad shape B=%d C=%d H=%d W=%d", B, C, H, W);
    hipLaunchKernelGGL(in_glu_fwd_kernel, dim3(C, B), dim3(256), 0, (hipStream_t)stream, y, gamma, beta, out, stats, C, H * W, eps);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? GTTS_OK : ifail(GTTS_E_HIP, "gtts_in_glu_forward: %s", hipGetErrorString(e));
}

extern "C" int gtts_in_glu_backward(const float *dout, const float *y, const float *gamma, const float *beta, const float *stats, float *dy,
                                    float *dgamma, float *dbeta, float *scratch, int B, int C, int H, int W, gtts_stream_t stream) {
    if (!dout || !y || !gamma || !beta || !stats || !dy || !dgamma || !dbeta || !scratch)
        return ifail(GTTS_E_NULL, "gtts_in_glu_backward: null argument");
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || B > 65535 || (size_t)H * W >= ((size_t)1 << 31))
        return ifail(GTTS_E_SHAPE, "gtts_in_glu_backward: bad shape B=%d C=%d H=%d W=%d", B, C, H, W);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(in_glu_bwd_kernel, dim3(C, B), dim3(256), 0, st, dout, y, gamma, beta, stats, dy, scratch, C, H * W);
    hipLaunchKernelGGL(in_glu_param_kernel, dim3((2 * C + 127) / 128), dim3(128), 0, st, scratch, dgamma, dbeta, B, 2 * C);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? GTTS_OK : ifail(GTTS_E_HIP, "gtts_in_glu_backward: %s", hipGetErrorString(e));
}
