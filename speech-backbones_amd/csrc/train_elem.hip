// train_elem.hip -- the elementwise glue of the training path: ResnetBlock's residual add
// (Grad-TTS/model/diffusion.py:77-78), the column mask of resampling inputs and gradients (diffusion.py:158,171), and the
// final 64 -> 1 convolution with its two masks (diffusion.py:175-176).  All bound by HBM: float4 / plane-strided streams.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>

#include "../../include/gradtts_abi.h"
#include "common.h"
#include "kernels.h"

namespace gtts {

// grid (B * C, ceil(HW / 1024)): out = a + b * mask[batch, w]   (a nullable: out = b * mask; mask nullable: out = a + b);
// b may be a channel slice of a wider tensor (batch stride b_bstride floats)
__global__ __launch_bounds__(256) void add_masked_kernel(const float *__restrict__ a, const float *__restrict__ b, const float *__restrict__ mask,
                                                         float *__restrict__ out, int C, int HW, int W, size_t b_bstride) {
    const int bc = blockIdx.x, bi = bc / C;
    const size_t base = (size_t)bc * HW;
    const float *bp = b + (size_t)bi * b_bstride + (size_t)(bc - bi * C) * HW;      // this (sample, channel) plane of the slice
    const float *mrow = mask ? mask + (size_t)bi * W : nullptr;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = blockIdx.y * 1024 + k * 256 + threadIdx.x;
        if (i < HW) {
            const float bv = mrow ? bp[i] * mrow[i % W] : bp[i];
            out[base + i] = a ? a[base + i] + bv : bv;
        }
    }
}

// grid (B * C, ceil(4hw / 1024)): out [2h][2w] = in [h][w] at the even (row, column) positions, zero elsewhere.  The weight
// gradient of a stride-2 convolution is the stride-1 weight gradient against the zero-inserted output gradient.
__global__ __launch_bounds__(256) void zero_insert2_kernel(const float *__restrict__ in, float *__restrict__ out, int h, int w) {
    const size_t bc = blockIdx.x;
    const int W2 = 2 * w, n = 4 * h * w;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int i = blockIdx.y * 1024 + k * 256 + threadIdx.x;
        if (i < n) {
            const int y = i / W2, x = i - y * W2;
            out[bc * n + i] = ((y | x) & 1) ? 0.f : in[bc * (size_t)(h * w) + (size_t)(y >> 1) * w + (x >> 1)];
        }
    }
}

// grid (B * C, ceil(hw / 256)): the four stride-2 phases of in [B,C,2h,2w] as channel blocks of out [B,4C,h,w]:
// out[(pr * 2 + pc) * C + c][y][x] = in[c][2y + 1 - pr][2x + 1 - pc]   (pr / pc = 1: even rows / columns).
// Upsample's gradients are stride-1 3x3 convolutions / weight gradients over these planes (model/_train_ops.py: ResampleConv).
__global__ __launch_bounds__(256) void space_to_depth2_kernel(const float *__restrict__ in, float *__restrict__ out, int C, int h, int w) {
    const int bc = blockIdx.x, bi = bc / C, c = bc - bi * C;
    const int i = blockIdx.y * 256 + threadIdx.x;
    if (i >= h * w) return;
    const int y = i / w, x = i - y * w;
    const float *p = in + (size_t)bc * 4 * h * w + (size_t)(2 * y) * (2 * w) + 2 * x;      // rows 2y, 2y + 1 of the 2h x 2w plane
    const float2 r0 = *reinterpret_cast<const float2 *>(p), r1 = *reinterpret_cast<const float2 *>(p + 2 * w);
    const size_t plane = (size_t)h * w, base = ((size_t)bi * 4 * C + c) * plane + i;
    out[base + (size_t)(3 * C) * plane] = r0.x;      // pr 1, pc 1: even row, even column
    out[base + (size_t)(2 * C) * plane] = r0.y;      // pr 1, pc 0
    out[base + (size_t)(1 * C) * plane] = r1.x;      // pr 0, pc 1
    out[base] = r1.y;                                // pr 0, pc 0
}

// First-layer weight gradient (the stacked (mu, x[, spk]) input: cin = 2 or 3; diffusion.py:140-147): a GEMM with K <= 27 is
// no MFMA job -- it is one pass over dy.  grid (cout / 4, B, WS_SPLIT): a thread sums, over its pixels of the slice, the
// products of four channels' dy[co,p] with the (x m)[ci, p + tap] neighbourhood it loads once (x is 2-3 planes, cache
// resident) into 4 x (cin * taps + 1) registers; the workgroup reduces them in a fixed order into part[b][slice][co][cin*taps + 1]
// (last: the bias gradient).
constexpr int WS_SPLIT = 4, WS_CO = 4;
template <int CIN, int K>
__global__ __launch_bounds__(256) void wgrad_small_kernel(const float *__restrict__ x, const float *__restrict__ mask, const float *__restrict__ dy,
                                                          float *__restrict__ part, int cout, int H, int W) {
    constexpr int NT = CIN * K * K, R = K / 2;
    const int co0 = blockIdx.x * WS_CO, bi = blockIdx.y, sl = blockIdx.z, HW = H * W;
    const float *pd = dy + ((size_t)bi * cout + co0) * HW;
    const float *px = x + (size_t)bi * CIN * HW;
    const float *mrow = mask + (size_t)bi * W;
    float acc[WS_CO][NT + 1];
#pragma unroll
    for (int c = 0; c < WS_CO; ++c)
#pragma unroll
        for (int i = 0; i <= NT; ++i) acc[c][i] = 0.f;
    const int per = (HW + WS_SPLIT - 1) / WS_SPLIT, p_end = min(HW, (sl + 1) * per);
    for (int p = sl * per + threadIdx.x; p < p_end; p += 256) {
        const int y = p / W, xx = p - y * W;
        float g[WS_CO];
#pragma unroll
        for (int c = 0; c < WS_CO; ++c) {
            g[c] = pd[(size_t)c * HW + p];
            acc[c][NT] += g[c];
        }
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
            for (int ky = 0; ky < K; ++ky)
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const int yy = y + ky - R, xc = xx + kx - R;
                    const bool in = yy >= 0 && yy < H && xc >= 0 && xc < W;
                    const float v = in ? px[(size_t)ci * HW + (size_t)yy * W + xc] * mrow[xc] : 0.f;
#pragma unroll
                    for (int c = 0; c < WS_CO; ++c) acc[c][(ci * K + ky) * K + kx] = fmaf(g[c], v, acc[c][(ci * K + ky) * K + kx]);
                }
    }
    // fixed-order reduction: butterfly inside each wave, then the four waves through LDS
    __shared__ float s_w[4][WS_CO * (NT + 1)];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < WS_CO; ++c)
#pragma unroll
        for (int i = 0; i <= NT; ++i) {
            float v = acc[c][i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
            if (lane == 0) s_w[wave][c * (NT + 1) + i] = v;
        }
    __syncthreads();
    for (int e = threadIdx.x; e < WS_CO * (NT + 1); e += 256) {
        const int c = e / (NT + 1), i = e - c * (NT + 1);
        part[(((size_t)bi * WS_SPLIT + sl) * cout + co0 + c) * (NT + 1) + i] = (s_w[0][e] + s_w[1][e]) + (s_w[2][e] + s_w[3][e]);
    }
}
// dw[co][ci][ky][kx] (reference layout) and db[co]: sum of the per-sample records in sample order; one thread per element
__global__ void wgrad_small_finish_kernel(const float *__restrict__ part, float *__restrict__ dw, float *__restrict__ db, int B, int cout, int nt) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= cout * (nt + 1)) return;
    const int co = i / (nt + 1), e = i - co * (nt + 1);
    double t = 0.0;
    for (int b = 0; b < B; ++b) t += (double)part[((size_t)b * cout + co) * (nt + 1) + e];
    if (e < nt) dw[(size_t)co * nt + e] = (float)t;
    else if (db) db[co] = (float)t;
}

// final_conv: out[b,p] = (sum_c w[c] x[b,c,p] m + bias) m,  m = mask[b, p % W]   (diffusion.py:175-176); grid (ceil(HW/256), B)
__global__ __launch_bounds__(256) void final_conv_fwd_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                                             const float *__restrict__ mask, float *__restrict__ out, int C, int HW, int W) {
    const int p = blockIdx.x * 256 + threadIdx.x, bi = blockIdx.y;
    if (p >= HW) return;
    const float m = mask[(size_t)bi * W + p % W];
    const float *px = x + (size_t)bi * C * HW + p;
    float acc = 0.f;
#pragma unroll 8
    for (int c = 0; c < C; ++c) acc = fmaf(w[c], px[(size_t)c * HW], acc);
    out[(size_t)bi * HW + p] = (acc * m + bias[0]) * m;
}

// dx[b,c,p] = w[c] dout[b,p] m^2; part[(b * nblk + blk)][c] = sum over the block's pixels of dout m^2 x[b,c,p] (c < C),
// part[..][C] = sum of dout m (bias); grid (nblk = ceil(HW/256), B).  One pixel per thread, channel loop, LDS tree per channel
// group would cost a barrier per channel: instead every wave reduces with shuffles and 4 waves combine through LDS.
__global__ __launch_bounds__(256) void final_conv_bwd_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ mask,
                                                             const float *__restrict__ dout, float *__restrict__ dx, float *__restrict__ part,
                                                             int C, int HW, int W) {
    extern __shared__ float s_w[];                 // [4 waves][C + 1]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int p = blockIdx.x * 256 + tid, bi = blockIdx.y;
    const bool ok = p < HW;
    const float m = ok ? mask[(size_t)bi * W + p % W] : 0.f;
    const float g = ok ? dout[(size_t)bi * HW + p] : 0.f;
    const float gm2 = g * m * m;
    const float *px = x + (size_t)bi * C * HW + p;
    float *pd = dx + (size_t)bi * C * HW + p;
    for (int c = 0; c <= C; ++c) {
        float v;
        if (c < C) {
            const float xv = ok ? px[(size_t)c * HW] : 0.f;
            if (ok) pd[(size_t)c * HW] = w[c] * gm2;
            v = gm2 * xv;
        } else {
            v = g * m;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if (lane == 0) s_w[wave * (C + 1) + c] = v;
    }
    __syncthreads();
    for (int c = tid; c <= C; c += 256)
        part[((size_t)bi * gridDim.x + blockIdx.x) * (C + 1) + c] = (s_w[c] + s_w[(C + 1) + c]) + (s_w[2 * (C + 1) + c] + s_w[3 * (C + 1) + c]);
}

// dw[c] (c < C), db = dw[C]: fixed-order sum of the partial records (fp64); one workgroup per channel
__global__ __launch_bounds__(256) void final_conv_bwd_finish_kernel(const float *__restrict__ part, int nrec, int C, float *__restrict__ dw,
                                                                    float *__restrict__ db) {
    __shared__ double s_a[256];
    const int c = blockIdx.x;
    double t = 0.0;
    for (int r = threadIdx.x; r < nrec; r += 256) t += (double)part[(size_t)r * (C + 1) + c];
    s_a[threadIdx.x] = t;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) s_a[threadIdx.x] += s_a[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (c < C) dw[c] = (float)s_a[0];
        else db[0] = (float)s_a[0];
    }
}

}  // namespace gtts

using namespace gtts;

static int efail(int code, const char *fmt, ...) {       // text goes to gtts_last_error() (plan.hip)
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return set_error(code, buf);
}
#define ECHK(expr)                                                                                                \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) return efail(GTTS_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// out = a + b * mask (a nullable: out = b * mask; mask nullable: out = a + b); a, b, out [B,C,H,W], mask [B,W]
// b_cstride: channels of the tensor b is a slice of (0: b is contiguous [B,C,H,W])
extern "C" int gtts_add_masked(const float *a, const float *b, const float *mask, float *out, int B, int C, int H, int W, int b_cstride,
                               gtts_stream_t stream) {
    if (!b || !out || (!a && !mask)) return efail(GTTS_E_NULL, "gtts_add_masked: null argument");
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0 || (long)H * W >= (1l << 30)) return efail(GTTS_E_SHAPE, "gtts_add_masked: bad shape");
    const int HW = H * W;
    hipLaunchKernelGGL(add_masked_kernel, dim3(B * C, (HW + 1023) / 1024), dim3(256), 0, (hipStream_t)stream, a, b, mask, out, C, HW, W,
                       (size_t)(b_cstride > 0 ? b_cstride : C) * HW);
    ECHK(hipGetLastError());
    return GTTS_OK;
}

extern "C" size_t gtts_conv_wgrad_small_scratch_floats(int B, int cin, int cout, int ksize) {
    if (B <= 0 || cin <= 0 || cout <= 0) return 0;
    return (size_t)B * WS_SPLIT * cout * (cin * ksize * ksize + 1);
}

// dw [cout][cin][k][k], db [cout] (nullable) of y = Conv2d_kxk(x * mask) + bias for the first layer: cin 2 or 3, k 3 or 1
extern "C" int gtts_conv_wgrad_small(const float *x, const float *mask, const float *dy, float *dw, float *db, float *scratch, int B, int cin,
                                     int cout, int H, int W, int ksize, gtts_stream_t stream) {
    if (!x || !mask || !dy || !dw || !scratch) return efail(GTTS_E_NULL, "gtts_conv_wgrad_small: null argument");
    if (B <= 0 || cout <= 0 || cout % WS_CO || H <= 0 || W <= 0 || (cin != 2 && cin != 3) || (ksize != 1 && ksize != 3))
        return efail(GTTS_E_SHAPE, "gtts_conv_wgrad_small: cin must be 2 or 3 and the kernel 1x1 or 3x3 (got cin %d, k %d)", cin, ksize);
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(cout / WS_CO, B, WS_SPLIT);
    if (cin == 2 && ksize == 3) hipLaunchKernelGGL((wgrad_small_kernel<2, 3>), grid, dim3(256), 0, st, x, mask, dy, scratch, cout, H, W);
    else if (cin == 3 && ksize == 3) hipLaunchKernelGGL((wgrad_small_kernel<3, 3>), grid, dim3(256), 0, st, x, mask, dy, scratch, cout, H, W);
    else if (cin == 2) hipLaunchKernelGGL((wgrad_small_kernel<2, 1>), grid, dim3(256), 0, st, x, mask, dy, scratch, cout, H, W);
    else hipLaunchKernelGGL((wgrad_small_kernel<3, 1>), grid, dim3(256), 0, st, x, mask, dy, scratch, cout, H, W);
    ECHK(hipGetLastError());
    const int nt = cin * ksize * ksize;
    hipLaunchKernelGGL(wgrad_small_finish_kernel, dim3((cout * (nt + 1) + 255) / 256), dim3(256), 0, st, scratch, dw, db, B * WS_SPLIT, cout, nt);
    ECHK(hipGetLastError());
    return GTTS_OK;
}

// out [B,4C,h,w] = the four stride-2 phases of in [B,C,2h,2w] (channel block (pr * 2 + pc): rows 2y + 1 - pr, columns 2x + 1 - pc)
extern "C" int gtts_space_to_depth2(const float *in, float *out, int B, int C, int h, int w, gtts_stream_t stream) {
    if (!in || !out) return efail(GTTS_E_NULL, "gtts_space_to_depth2: null argument");
    if (B <= 0 || C <= 0 || h <= 0 || w <= 0 || (long)h * w >= (1l << 28)) return efail(GTTS_E_SHAPE, "gtts_space_to_depth2: bad shape");
    hipLaunchKernelGGL(space_to_depth2_kernel, dim3(B * C, (h * w + 255) / 256), dim3(256), 0, (hipStream_t)stream, in, out, C, h, w);
    ECHK(hipGetLastError());
    return GTTS_OK;
}

// out [B,C,2h,2w] = in [B,C,h,w] at the even positions, zero elsewhere
extern "C" int gtts_zero_insert2(const float *in, float *out, int B, int C, int h, int w, gtts_stream_t stream) {
    if (!in || !out) return efail(GTTS_E_NULL, "gtts_zero_insert2: null argument");
    if (B <= 0 || C <= 0 || h <= 0 || w <= 0 || (long)h * w >= (1l << 28)) return efail(GTTS_E_SHAPE, "gtts_zero_insert2: bad shape");
    hipLaunchKernelGGL(zero_insert2_kernel, dim3(B * C, (4 * h * w + 1023) / 1024), dim3(256), 0, (hipStream_t)stream, in, out, h, w);
    ECHK(hipGetLastError());
    return GTTS_OK;
}

// out [B,1,H,W] = (Conv2d_1x1(x * mask; w [C], bias [1]) ) * mask
extern "C" int gtts_final_conv_forward(const float *x, const float *w, const float *bias, const float *mask, float *out, int B, int C,
                                       int H, int W, gtts_stream_t stream) {
    if (!x || !w || !bias || !mask || !out) return efail(GTTS_E_NULL, "gtts_final_conv_forward: null argument");
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return efail(GTTS_E_SHAPE, "gtts_final_conv_forward: bad shape");
    const int HW = H * W;
    hipLaunchKernelGGL(final_conv_fwd_kernel, dim3((HW + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, x, w, bias, mask, out, C, HW, W);
    ECHK(hipGetLastError());
    return GTTS_OK;
}

extern "C" size_t gtts_final_conv_scratch_floats(int B, int C, int H, int W) {
    if (B <= 0 || C <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)B * ((H * W + 255) / 256) * (C + 1);
}

// dx [B,C,H,W], dw [C], db [1] of gtts_final_conv_forward given dout [B,1,H,W]
extern "C" int gtts_final_conv_backward(const float *x, const float *w, const float *mask, const float *dout, float *dx, float *dw,
                                        float *db, float *scratch, int B, int C, int H, int W, gtts_stream_t stream) {
    if (!x || !w || !mask || !dout || !dx || !dw || !db || !scratch) return efail(GTTS_E_NULL, "gtts_final_conv_backward: null argument");
    if (B <= 0 || C <= 0 || C > 1024 || H <= 0 || W <= 0) return efail(GTTS_E_SHAPE, "gtts_final_conv_backward: bad shape");
    const int HW = H * W, nblk = (HW + 255) / 256;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(final_conv_bwd_kernel, dim3(nblk, B), dim3(256), (size_t)4 * (C + 1) * sizeof(float), st, x, w, mask, dout, dx, scratch, C,
                       HW, W);
    ECHK(hipGetLastError());
    hipLaunchKernelGGL(final_conv_bwd_finish_kernel, dim3(C + 1), dim3(256), 0, st, scratch, B * nblk, C, dw, db);
    ECHK(hipGetLastError());
    return GTTS_OK;
}
