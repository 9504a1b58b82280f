// glue.hip -- pre-decoder glue of GradTTS.forward on the device:
//   generate_path(duration, mask)        Grad-TTS/model/utils.py:26-39
//   mu_y = attn^T . mu_x                 Grad-TTS/model/tts.py:89-91
//   z = mu_y + randn / temperature       Grad-TTS/model/tts.py:94
// The reference builds a dense 0/1 path [B, t_x, T] from cumulative durations and multiplies it with mu_x.  A path
// column has at most one 1, so mu_y[:, :, j] is a gather mu_x[:, :, i(j)] with i(j) = the token whose duration
// interval [cum[i-1], cum[i]) contains frame j.  One workgroup per utterance:
//   1. cum = cumsum(duration) sequentially in fp32 (the order of the reference's CPU path: durations scaled by a
//      non-integer length_scale are not exactly summable, so the order matters for bit-exactness);
//   2. per frame j: i(j) by binary search on cum (the reference's test is the float compare j < cum[i],
//      utils.py:6-10,31), validity = x_mask[i] * y_mask[j] (tts.py:85), attn column written as 0/1;
//   3. mu_y column gathered (exactly the value the reference's matmul produces: one term times 1.0 plus zeros),
//      z = mu_y + noise / temperature with the reference's two fp32 roundings.
#include "common.h"
#include "kernels.h"

namespace gtts {

__global__ __launch_bounds__(256) void expand_alignment_kernel(const float *__restrict__ dur, const float *__restrict__ x_mask,
                                                               const int *__restrict__ y_len, const float *__restrict__ mu_x,
                                                               const float *__restrict__ noise, float temperature,
                                                               float *__restrict__ attn, float *__restrict__ mu_y,
                                                               float *__restrict__ z, int F, int tx, int T) {
    extern __shared__ float s_cum[];       // [tx] cumulative durations, then [T] token index per frame (as int)
    int *s_idx = reinterpret_cast<int *>(s_cum + tx);
    const int b = blockIdx.x, tid = threadIdx.x;
    const float *d = dur + (size_t)b * tx;
    const float *xm = x_mask + (size_t)b * tx;
    if (tid == 0) {
        float c = 0.f;
        for (int i = 0; i < tx; ++i) {
            c = __fadd_rn(c, d[i]);
            s_cum[i] = c;
        }
    }
    __syncthreads();
    const int ylen = y_len[b];
    for (int j = tid; j < T; j += 256) {
        const float fj = (float)j;
        int lo = 0, hi = tx;                   // first i with fj < cum[i]   (cum is non-decreasing)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (fj < s_cum[mid]) hi = mid; else lo = mid + 1;
        }
        const bool ok = lo < tx && j < ylen && xm[lo] != 0.f;
        // a non-binary x_mask scales the path entry exactly like path * mask does (utils.py:38)
        s_idx[j] = ok ? lo : -1;
    }
    __syncthreads();
    // attn [t_x][T]: path * mask
    float *ab = attn + (size_t)b * tx * T;
    for (size_t e = tid; e < (size_t)tx * T; e += 256) {
        const int i = (int)(e / T), j = (int)(e - (size_t)i * T);
        ab[e] = s_idx[j] == i ? xm[i] : 0.f;
    }
    const float *mx = mu_x + (size_t)b * F * tx;
    const size_t ob = (size_t)b * F * T;
    for (size_t e = tid; e < (size_t)F * T; e += 256) {
        const int f = (int)(e / T), j = (int)(e - (size_t)f * T);
        const int i = s_idx[j];
        const float m = i >= 0 ? __fmul_rn(xm[i], mx[(size_t)f * tx + i]) : 0.f;
        mu_y[ob + e] = m;
        if (z) z[ob + e] = noise ? __fadd_rn(m, __fdiv_rn(noise[ob + e], temperature)) : m;
    }
}

hipError_t launch_expand_alignment(const float *dur, const float *x_mask, const int *y_len, const float *mu_x,
                                   const float *noise, float temperature, float *attn, float *mu_y, float *z, int B, int F,
                                   int tx, int T, hipStream_t st) {
    const size_t smem = ((size_t)tx + T) * 4;
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    if (smem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&expand_alignment_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(expand_alignment_kernel, dim3(B), dim3(256), smem, st, dur, x_mask, y_len, mu_x, noise, temperature,
                       attn, mu_y, z, F, tx, T);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ log_prior
// Gaussian log-likelihood of every (token, frame) pair, the score matrix MAS walks (Grad-TTS/model/tts.py:130-139):
//   log_prior[b, i, j] = sum_f -0.5 y[f,j]^2 + sum_f mu_x[f,i] y[f,j] + sum_f -0.5 mu_x[f,i]^2 - 0.5 log(2 pi) F
// The reference builds it from three dense [t_x,F]x[F,T] matmuls and two broadcast adds; here one workgroup owns a
// 16-token x 64-frame tile, stages both operand tiles in LDS once and evaluates the same quantity as
// -0.5 sum_f (y - mu)^2 + const (no cancellation between the three large terms).  Output feeds gtts_mas_maximum_path
// directly on the device.
__global__ __launch_bounds__(256) void log_prior_kernel(const float *__restrict__ mu_x, const float *__restrict__ y,
                                                        float *__restrict__ out, int F, int tx, int T, float cst) {
    extern __shared__ float sm[];          // [F][64] frames, then [F][16] tokens
    float *s_y = sm, *s_mu = sm + (size_t)F * 64;
    const int b = blockIdx.z, i0 = blockIdx.y * 16, j0 = blockIdx.x * 64, tid = threadIdx.x;
    const float *yb = y + (size_t)b * F * T, *mb = mu_x + (size_t)b * F * tx;
    for (int e = tid; e < F * 64; e += 256) {
        const int f = e >> 6, j = j0 + (e & 63);
        s_y[e] = j < T ? yb[(size_t)f * T + j] : 0.f;
    }
    for (int e = tid; e < F * 16; e += 256) {
        const int f = e >> 4, i = i0 + (e & 15);
        s_mu[e] = i < tx ? mb[(size_t)f * tx + i] : 0.f;
    }
    __syncthreads();
    const int jl = tid & 63, ig = tid >> 6;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int f = 0; f < F; ++f) {
        const float yv = s_y[f * 64 + jl];
        const float4 m4 = *reinterpret_cast<const float4 *>(s_mu + f * 16 + ig * 4);
        const float d0 = yv - m4.x, d1 = yv - m4.y, d2 = yv - m4.z, d3 = yv - m4.w;
        acc[0] = fmaf(d0, d0, acc[0]); acc[1] = fmaf(d1, d1, acc[1]);
        acc[2] = fmaf(d2, d2, acc[2]); acc[3] = fmaf(d3, d3, acc[3]);
    }
    const int j = j0 + jl;
    if (j < T) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = i0 + ig * 4 + k;
            if (i < tx) out[((size_t)b * tx + i) * T + j] = fmaf(-0.5f, acc[k], cst);
        }
    }
}

hipError_t launch_log_prior(const float *mu_x, const float *y, float *out, int B, int F, int tx, int T, hipStream_t st) {
    const size_t smem = (size_t)F * 80 * sizeof(float);
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    if (smem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&log_prior_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
    }
    const float cst = (float)(-0.5 * 1.8378770664093453 * (double)F);     // -0.5 * log(2 pi) * n_feats
    hipLaunchKernelGGL(log_prior_kernel, dim3((T + 63) / 64, (tx + 15) / 16, B), dim3(256), smem, st, mu_x, y, out, F, tx, T, cst);
    return hipGetLastError();
}

}  // namespace gtts
