// vc.hip -- DiffVC-only kernels around the shared U-Net (DiffVC/model/diffusion.py:61-76,151-155,
// DiffVC/model/modules.py:128-166).  All tiny or bandwidth-bound next to the 2 TFLOP-per-utterance trunk.
//   xt_ref           compute_diffused_mean(ref, ref_mask, mean_ref, t)                diffusion.py:151-155,173
//   instnorm_stats   InstanceNorm2d(affine) statistics -> per-(sample, channel) scale/shift   modules.py:141,...
//   ref_pool         sum over (mel-bin, frame) of GLU(IN(block32 raw)) * mask^2        modules.py:165-166 (folded)
//   vc_cond          final_conv of RefBlock applied to the pooled vector + cond_block MLP     diffusion.py:69-73
//   prep_vc          cat([stack([mean, x]), condition broadcast over (80, T)], 1)      diffusion.py:65,74-76
#include "common.h"
#include "kernels.h"

namespace gtts {

__global__ void xt_ref_kernel(const float *__restrict__ ref, const float *__restrict__ mean_ref,
                              const float *__restrict__ ref_mask, float *__restrict__ out, float w0, float w1, int F, int Tr,
                              size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int col = (int)(i % Tr);
    const size_t b = i / ((size_t)F * Tr);
    // xt_mean = x0 * x0_weight + mean * mean_weight; return xt_mean * mask    (fp32, no contraction)
    const float v = __fadd_rn(__fmul_rn(ref[i], w0), __fmul_rn(mean_ref[i], w1));
    out[i] = __fmul_rn(v, ref_mask[b * Tr + col]);
}

hipError_t launch_xt_ref(const float *ref, const float *mean_ref, const float *ref_mask, float *out, float w0, float w1,
                         int B, int F, int Tr, hipStream_t st) {
    const size_t total = (size_t)B * F * Tr;
    hipLaunchKernelGGL(xt_ref_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, ref, mean_ref, ref_mask, out,
                       w0, w1, F, Tr, total);
    return hipGetLastError();
}

__device__ __forceinline__ double block_sum_d(double v, double *sm) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) sm[wave] = v;
    __syncthreads();
    double t = 0.0;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sm[w];
    return t;
}

// grid (C, B): statistics over all H*W positions of one (sample, channel) plane, masked frames included (like
// torch's InstanceNorm2d), biased variance, eps 1e-5.  Fixed reduction order -> deterministic.
__global__ void instnorm_stats_kernel(const float *__restrict__ x, const float *__restrict__ gamma,
                                      const float *__restrict__ beta, float *__restrict__ sc, float *__restrict__ sh, int C,
                                      int HW) {
    __shared__ double sm[8];
    const int c = blockIdx.x, b = blockIdx.y;
    const float *p = x + ((size_t)b * C + c) * HW;
    double s1 = 0.0, s2 = 0.0;
    for (int i = threadIdx.x; i < HW; i += 256) {
        const double v = (double)p[i];
        s1 += v;
        s2 += v * v;
    }
    s1 = block_sum_d(s1, sm);
    s2 = block_sum_d(s2, sm);
    if (threadIdx.x == 0) {
        const double mean = s1 / HW;
        double var = s2 / HW - mean * mean;
        if (var < 0.0) var = 0.0;
        const double a = (double)gamma[c] / sqrt(var + 1e-5);
        sc[(size_t)b * C + c] = (float)a;
        sh[(size_t)b * C + c] = (float)((double)beta[c] - mean * a);
    }
}

hipError_t launch_instnorm_stats(const float *x, const float *gamma, const float *beta, float *sc, float *sh, int B, int C,
                                 int HW, hipStream_t st) {
    hipLaunchKernelGGL(instnorm_stats_kernel, dim3(C, B), dim3(256), 0, st, x, gamma, beta, sc, sh, C, HW);
    return hipGetLastError();
}

// grid (Ch, B): S[b][c] = sum_{h,t} GLU(IN(raw))[c][h,t] * mask[t]^2.   (y*mask -> 1x1 conv -> *mask -> sum is linear
// in this vector: sum_t mask*(W (g*mask) + b) = W S + b * F * sum_t mask.)
__global__ void ref_pool_kernel(const float *__restrict__ raw, const float *__restrict__ sc, const float *__restrict__ sh,
                                const float *__restrict__ ref_mask, float *__restrict__ S, int Ch, int F, int Tr) {
    __shared__ double sm[8];
    const int c = blockIdx.x, b = blockIdx.y;
    const int HW = F * Tr;
    const float *pa = raw + ((size_t)b * 2 * Ch + c) * HW;
    const float *pb = raw + ((size_t)b * 2 * Ch + Ch + c) * HW;
    const float sa = sc[(size_t)b * 2 * Ch + c], ha = sh[(size_t)b * 2 * Ch + c];
    const float sb = sc[(size_t)b * 2 * Ch + Ch + c], hb = sh[(size_t)b * 2 * Ch + Ch + c];
    double acc = 0.0;
    for (int i = threadIdx.x; i < HW; i += 256) {
        const float m = ref_mask[(size_t)b * Tr + i % Tr];
        const float g = (pa[i] * sa + ha) * __builtin_amdgcn_rcpf(1.0f + __expf(-(pb[i] * sb + hb)));
        acc += (double)(g * m * m);
    }
    acc = block_sum_d(acc, sm);
    if (threadIdx.x == 0) S[(size_t)b * Ch + c] = (float)acc;
}

hipError_t launch_ref_pool(const float *raw, const float *sc, const float *sh, const float *ref_mask, float *S, int B,
                           int Ch, int F, int Tr, hipStream_t st) {
    hipLaunchKernelGGL(ref_pool_kernel, dim3(Ch, B), dim3(256), 0, st, raw, sc, sh, ref_mask, S, Ch, F, Tr);
    return hipGetLastError();
}

__device__ __forceinline__ float dot_f(const float *__restrict__ w, const float *v, int n) {
    float acc = 0.f;
    for (int i = 0; i < n; ++i) acc = fmaf(w[i], v[i], acc);
    return acc;
}

// grid B.  condition = [sinusoidal emb (dim) | RefBlock feature (dim_cond, optional) | c (cdim)]
//          -> Linear -> Mish -> Linear -> cond [dim_cond]                           (diffusion.py:62,69-73)
__global__ void vc_cond_kernel(const float *__restrict__ tb, int tb_stride, int semb_off, int dim,
                               const float *__restrict__ S, const float *__restrict__ ref_mask, const float *__restrict__ fw,
                               const float *__restrict__ fb, const float *__restrict__ c, const float *__restrict__ w0,
                               const float *__restrict__ b0, const float *__restrict__ w2, const float *__restrict__ b2,
                               float *__restrict__ cond, int Ch, int dim_cond, int cdim, int F, int Tr, int use_ref) {
    extern __shared__ float sm[];
    const int b = blockIdx.x;
    const int nin = dim + (use_ref ? dim_cond : 0) + cdim;
    float *in = sm, *h = sm + nin, *sv = h + 4 * dim_cond;
    __shared__ float s_len;
    for (int i = threadIdx.x; i < dim; i += blockDim.x) in[i] = tb[(size_t)b * tb_stride + semb_off + i];
    if (use_ref) {
        for (int i = threadIdx.x; i < Ch; i += blockDim.x) sv[i] = S[(size_t)b * Ch + i];
        if (threadIdx.x == 0) {
            float L = 0.f;
            for (int t = 0; t < Tr; ++t) L += ref_mask[(size_t)b * Tr + t];
            s_len = L;
        }
        __syncthreads();
        // RefBlock tail: (final_conv(y*mask)*mask).sum((2,3)) / (mask.sum((2,3)) * F)      modules.py:165-166
        for (int j = threadIdx.x; j < dim_cond; j += blockDim.x) {
            const float num = dot_f(fw + (size_t)j * Ch, sv, Ch) + fb[j] * (float)F * s_len;
            in[dim + j] = num / (s_len * (float)F);
        }
    }
    const int coff = dim + (use_ref ? dim_cond : 0);
    for (int i = threadIdx.x; i < cdim; i += blockDim.x) in[coff + i] = c[(size_t)b * cdim + i];
    __syncthreads();
    for (int j = threadIdx.x; j < 4 * dim_cond; j += blockDim.x) h[j] = mish_f(b0[j] + dot_f(w0 + (size_t)j * nin, in, nin));
    __syncthreads();
    for (int j = threadIdx.x; j < dim_cond; j += blockDim.x)
        cond[(size_t)b * dim_cond + j] = b2[j] + dot_f(w2 + (size_t)j * 4 * dim_cond, h, 4 * dim_cond);
}

hipError_t launch_vc_cond(const float *tb, int tb_stride, int semb_off, int dim, const float *S, const float *ref_mask,
                          const float *fw, const float *fb, const float *c, const float *w0, const float *b0,
                          const float *w2, const float *b2, float *cond, int B, int Ch, int dim_cond, int cdim, int F, int Tr,
                          int use_ref, hipStream_t st) {
    const size_t smem = (size_t)(dim + dim_cond + cdim + 4 * dim_cond + Ch) * sizeof(float);
    hipLaunchKernelGGL(vc_cond_kernel, dim3(B), dim3(256), smem, st, tb, tb_stride, semb_off, dim, S, ref_mask, fw, fb, c, w0,
                       b0, w2, b2, cond, Ch, dim_cond, cdim, F, Tr, use_ref);
    return hipGetLastError();
}

__global__ void prep_vc_kernel(const float *__restrict__ mean, const float *__restrict__ x, const float *__restrict__ cond,
                               float *__restrict__ x0, int F, int T, int ncond) {
    // grid: (ceil(F*T/256), 2 + ncond, B)
    const int b = blockIdx.z, ch = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= F * T) return;
    float v;
    if (ch == 0) v = mean[(size_t)b * F * T + i];
    else if (ch == 1) v = x[(size_t)b * F * T + i];
    else v = cond[(size_t)b * ncond + ch - 2];
    x0[((size_t)b * (2 + ncond) + ch) * F * T + i] = v;
}

hipError_t launch_prep_vc(const float *mean, const float *x, const float *cond, float *x0, int B, int F, int T, int ncond,
                          hipStream_t st) {
    dim3 grid((F * T + 255) / 256, 2 + ncond, B);
    hipLaunchKernelGGL(prep_vc_kernel, grid, dim3(256), 0, st, mean, x, cond, x0, F, T, ncond);
    return hipGetLastError();
}

}  // namespace gtts
