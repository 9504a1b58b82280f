// train.hip -- first kernels of the training hot path: one score-network forward + backward
// per optimiser step at [16, C, 80, 172] (Grad-TTS/train.py:105-119; Grad-TTS/model/diffusion.py:244-252,281-294).
//
//   gtts_diffusion_noising   forward_diffusion (diffusion.py:244-252): xt = (x0 e^{-c/2} + mu (1 - e^{-c/2}) + z sqrt(1 - e^{-c})) mask
//   gtts_score_loss          loss_t (diffusion.py:281-288): sum((eps sqrt(1 - e^{-c}) + z)^2) / (sum(mask) F) and d loss / d eps
//   gtts_conv3x3_masked      y = Conv2d_3x3(x * mask) + bias      (Block.forward, diffusion.py:56-57) on the inference MFMA kernel
//   gtts_conv3x3_dgrad       d loss / d (x * mask) = Conv2d_3x3(dy, W^T flipped): the same kernel with repacked weights
//   gtts_conv3x3_wgrad       dW[co][ci][ky][kx] = sum_{b,y,x} dy[b,co,y,x] (x mask)[b,ci,y+ky-1,x+kx-1], db = sum dy:
//                            an MFMA reduction over PIXELS (the K dimension is the frame axis, contiguous in NCHW, so both
//                            operands are read straight from HBM as 8-pixel runs per lane -- no LDS, no barriers);
//                            split-bf16 (3 MFMAs per product, fp32 accumulate), partial sums combined with fp32 atomics.
// The 3x3 convolutions are 84 % of the U-Net's FLOPs forward and (twice that) backward; GroupNorm / Mish / attention
// backward stay PyTorch autograd for now (model/_train_ops.py wraps these entry points in torch.autograd.Function).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "../../include/gradtts_abi.h"
#include "common.h"
#include "kernels.h"

namespace gtts {

// ------------------------------------------------------------------------------------------------ elementwise
// cum = beta_min t + 0.5 (beta_max - beta_min) t^2  (get_noise cumulative, diffusion.py:219-224), per sample
__global__ void noising_kernel(const float *__restrict__ x0, const float *__restrict__ mu, const float *__restrict__ z,
                               const float *__restrict__ mask, const float *__restrict__ t, float bmin, float bmax,
                               float *__restrict__ xt, float *__restrict__ zm, int F, int T, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const size_t b = i / ((size_t)F * T);
    const int col = (int)(i % T);
    const float tv = t[b];
    const float cum = bmin * tv + 0.5f * (bmax - bmin) * (tv * tv);
    // correctly rounded fp32 exp (through double): 1 - e^{-cum} cancels catastrophically for t -> 0 (train.py clamps t at
    // 1e-5), where a 1-ulp difference in expf changes the noise scale by percents
    const float decay = (float)exp(-0.5 * (double)cum);
    const float m = mask[b * T + col];
    const float mean = x0[i] * decay + mu[i] * (1.0f - decay);
    const float zz = z[i];
    xt[i] = (mean + zz * sqrtf(1.0f - (float)exp(-(double)cum))) * m;
    zm[i] = zz * m;
}

// r = eps * sqrt(1 - e^{-cum}) + z;  partial[blk] = sum r^2 (fixed order inside a workgroup);  geps = 2 r s / denom
__global__ void score_loss_kernel(const float *__restrict__ eps, const float *__restrict__ z, const float *__restrict__ t,
                                  float bmin, float bmax, float inv_denom, float *__restrict__ partial,
                                  float *__restrict__ geps, int F, int T, size_t total) {
    __shared__ float red[4];
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    float sq = 0.f;
    if (i < total) {
        const size_t b = i / ((size_t)F * T);
        const float tv = t[b];
        const float cum = bmin * tv + 0.5f * (bmax - bmin) * (tv * tv);
        const float s = sqrtf(1.0f - (float)exp(-(double)cum));
        const float r = eps[i] * s + z[i];
        sq = r * r;
        if (geps) geps[i] = 2.0f * r * s * inv_denom;
    }
    sq = wave_sum(sq);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// ------------------------------------------------------------------------------------------------ weight gradient
// One wave owns a 32 (cout) x 32 (cin) x 9 (taps) tile of dW and a slice of the pixel blocks (b, y, 16 consecutive x).
// MFMA 32x32x16: A[m = co][k = pixel] from dy, B[n = ci][k = pixel] from x * mask shifted by the tap; lane (l31, kg) holds
// the 8 consecutive pixels x0 + 8 kg .. + 7 of its channel, read as two 16-byte loads.
struct WgradArgs {
    const float *x;        // [B][cin][H][W]
    const float *mask;     // [B][W] or nullptr
    const float *dy;       // [B][cout][H][W]
    float *dw;             // [cout][cin][3][3]  (zeroed by the caller; fp32 atomics)
    float *db;             // [cout] or nullptr
    int B, cin, cout, H, W;
    int nblk;              // B * H * ceil(W / 16)
    int nslice;            // pixel slices (waves) per (co tile, ci tile)
};

__device__ __forceinline__ unsigned pack_bf16(float a, float b) {        // (lo half = a, hi half = b), RNE
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
    bf16x2 v;
    v[0] = (__bf16)a;
    v[1] = (__bf16)b;
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf16_hi_part(float x) { return (float)(__bf16)x; }

__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_kernel(const WgradArgs a) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, kg = lane >> 5;
    const int ncit = a.cin / 32, ncot = a.cout / 32;
    const int gw = blockIdx.x * 4 + wave;                  // global wave id
    const int tile = gw % (ncit * ncot), slice = gw / (ncit * ncot);
    if (slice >= a.nslice) return;
    const int co0 = (tile / ncit) * 32, ci0 = (tile % ncit) * 32;
    const int xb16 = (a.W + 15) / 16;
    const size_t HW = (size_t)a.H * a.W;

    f32x16 acc[3][3];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ky][kx][r] = 0.f;
    float bsum = 0.f;

    const int per = (a.nblk + a.nslice - 1) / a.nslice;
    const int blk0 = slice * per, blk1 = min(a.nblk, blk0 + per);
    for (int blk = blk0; blk < blk1; ++blk) {
        const int xb = blk % xb16;
        const int y = (blk / xb16) % a.H;
        const int b = blk / (xb16 * a.H);
        const int px0 = xb * 16 + 8 * kg;                   // first pixel of this lane's run
        // ---- A: dy[b][co0 + l31][y][px0 .. px0 + 7]
        float av[8];
        {
            const float *p = a.dy + ((size_t)b * a.cout + co0 + l31) * HW + (size_t)y * a.W + px0;
#pragma unroll
            for (int i = 0; i < 8; ++i) av[i] = (px0 + i < a.W) ? p[i] : 0.f;
        }
        u32x4 ah, al;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float h0 = bf16_hi_part(av[2 * i]), h1 = bf16_hi_part(av[2 * i + 1]);
            ah[i] = pack_bf16(av[2 * i], av[2 * i + 1]);
            al[i] = pack_bf16(av[2 * i] - h0, av[2 * i + 1] - h1);
            bsum += av[2 * i] + av[2 * i + 1];
        }
        const bf16x8 Ah = __builtin_bit_cast(bf16x8, ah), Al = __builtin_bit_cast(bf16x8, al);
        // ---- B: rows y-1, y, y+1 of (x * mask)[b][ci0 + l31], pixels px0 - 1 .. px0 + 8
        const float *xrow = a.x + ((size_t)b * a.cin + ci0 + l31) * HW;
        const float *mrow = a.mask ? a.mask + (size_t)b * a.W : nullptr;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int yy = y + ky - 1;
            float v[10];
            const bool rowok = yy >= 0 && yy < a.H;
#pragma unroll
            for (int i = 0; i < 10; ++i) {
                const int px = px0 - 1 + i;
                const bool ok = rowok && px >= 0 && px < a.W;
                float t = ok ? xrow[(size_t)(rowok ? yy : 0) * a.W + (ok ? px : 0)] : 0.f;
                if (mrow) t *= ok ? mrow[px] : 0.f;
                v[i] = t;
            }
            float lo[10];
#pragma unroll
            for (int i = 0; i < 10; ++i) lo[i] = v[i] - bf16_hi_part(v[i]);
            // kx = 0: elements 0..7, kx = 1: 1..8, kx = 2: 2..9 of v[]
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                u32x4 bh, bl;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    bh[i] = pack_bf16(v[kx + 2 * i], v[kx + 2 * i + 1]);
                    bl[i] = pack_bf16(lo[kx + 2 * i], lo[kx + 2 * i + 1]);
                }
                const bf16x8 Bh = __builtin_bit_cast(bf16x8, bh), Bl = __builtin_bit_cast(bf16x8, bl);
                acc[ky][kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al, Bh, acc[ky][kx], 0, 0, 0);
                acc[ky][kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bl, acc[ky][kx], 0, 0, 0);
                acc[ky][kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, acc[ky][kx], 0, 0, 0);
            }
        }
    }
    // ---- combine: D[m = co][n = ci]; lane (l31 = ci, kg) holds rows (rg&3) + 8 (rg>>2) + 4 kg
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) {
                const int co = co0 + (rg & 3) + 8 * (rg >> 2) + 4 * kg;
                atomicAdd(a.dw + (((size_t)co * a.cin + ci0 + l31) * 3 + ky) * 3 + kx, acc[ky][kx][rg]);
            }
    if (a.db && ci0 == 0) {
        bsum += __shfl_xor(bsum, 32, 64);                   // the two 8-pixel halves of the run
        if (kg == 0) atomicAdd(a.db + co0 + l31, bsum);
    }
}

}  // namespace gtts

using namespace gtts;

static int tfail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return set_error(code, buf);
}
#define TCHK(expr)                                                                                                \
    do {                                                                                                          \
        hipError_t e_ = (expr);                                                                                   \
        if (e_ != hipSuccess) return tfail(GTTS_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

extern "C" int gtts_diffusion_noising(const float *x0, const float *mu, const float *z, const float *mask, const float *t,
                                      float beta_min, float beta_max, float *xt, float *z_masked, int B, int F, int T,
                                      gtts_stream_t stream) {
    if (!x0 || !mu || !z || !mask || !t || !xt || !z_masked) return tfail(GTTS_E_NULL, "gtts_diffusion_noising: null argument");
    if (B <= 0 || F <= 0 || T <= 0) return tfail(GTTS_E_SHAPE, "gtts_diffusion_noising: bad shape");
    const size_t total = (size_t)B * F * T;
    hipLaunchKernelGGL(noising_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x0, mu, z, mask, t,
                       beta_min, beta_max, xt, z_masked, F, T, total);
    TCHK(hipGetLastError());
    return GTTS_OK;
}

extern "C" size_t gtts_score_loss_partials(int B, int F, int T) { return ((size_t)B * F * T + 255) / 256; }

extern "C" int gtts_score_loss(const float *eps, const float *z_masked, const float *t, float beta_min, float beta_max,
                               float inv_denom, float *partials, float *grad_eps, int B, int F, int T, gtts_stream_t stream) {
    if (!eps || !z_masked || !t || !partials) return tfail(GTTS_E_NULL, "gtts_score_loss: null argument");
    if (B <= 0 || F <= 0 || T <= 0) return tfail(GTTS_E_SHAPE, "gtts_score_loss: bad shape");
    const size_t total = (size_t)B * F * T;
    hipLaunchKernelGGL(score_loss_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, eps, z_masked, t,
                       beta_min, beta_max, inv_denom, partials, grad_eps, F, T, total);
    TCHK(hipGetLastError());
    return GTTS_OK;
}

extern "C" size_t gtts_conv3x3_packed_bytes(int cin, int cout) {
    if (cin <= 0 || cout <= 0) return 0;
    return (conv_packed_bytes(CONV_C3, cin, cout) + 255) / 256 * 256;
}

// transposed != 0: w is the FORWARD weight [cin_of_this_conv... i.e. forward cout][forward cin = cout of this conv][3][3] and
// is packed transposed with flipped taps (the data-gradient convolution)
extern "C" int gtts_conv3x3_pack(const float *w, void *packed, int cin, int cout, int transposed, gtts_stream_t stream) {
    if (!w || !packed) return tfail(GTTS_E_NULL, "gtts_conv3x3_pack: null argument");
    if (cin <= 0 || cout <= 0) return tfail(GTTS_E_SHAPE, "gtts_conv3x3_pack: bad shape");
    // cin / cout are those of the convolution being packed (for the data gradient: cin = forward cout, cout = forward cin)
    TCHK(launch_pack_conv(transposed ? CONV_C3 + 16 : CONV_C3, w, (unsigned char *)packed, cin, cout, (hipStream_t)stream));
    return GTTS_OK;
}

// y = Conv2d_3x3(x * mask, packed W) + bias; x [B,cin,H,W], mask [B,W] (columns), y [B,cout,H,W].  cout % 64 == 0 (128 above 64).
// x1 (nullable) / c0: the input is the channel concatenation of x [B,c0,H,W] and x1 [B,cin-c0,H,W] (c0 a multiple of 16), read
// in place (torch.cat of the up path, diffusion.py:166)
// omask (nullable): [B][W] column mask multiplied into the output -- the data gradient of a masked convolution is
// conv(dy; transposed weights) * mask, and the mask rides in the epilogue instead of a second pass over dx
extern "C" int gtts_conv3x3_masked3(const float *x, const float *x1, int c0, const float *mask, const float *omask, const void *packed,
                                    const float *bias, float *y, int B, int cin, int cout, int H, int W, gtts_stream_t stream) {
    if (!x || !mask || !packed || !bias || !y) return tfail(GTTS_E_NULL, "gtts_conv3x3_masked: null argument");
    if (x1 && (c0 <= 0 || c0 >= cin || c0 % 16)) return tfail(GTTS_E_SHAPE, "gtts_conv3x3_masked: c0 must be a multiple of 16 inside (0, cin) (got %d of %d)", c0, cin);
    if (B <= 0 || cin <= 0 || cout <= 0 || H <= 0 || W <= 0) return tfail(GTTS_E_SHAPE, "gtts_conv3x3_masked: bad shape");
    if (cout % (cout > 64 ? 128 : 64) != 0) return tfail(GTTS_E_SHAPE, "gtts_conv3x3_masked: cout must be 64 or a multiple of 128 (got %d)", cout);
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.src0 = x; a.src1 = x1 ? x1 : x; a.c0 = x1 ? c0 : cin; a.c1 = x1 ? cin - c0 : 0; a.cin = cin;
    a.B = B; a.Hin = a.Hout = H; a.Win = a.Wout = W;
    a.mask = mask; a.T = W; a.lvl_in = a.lvl_out = 0;
    a.pro = PRO_MASK; a.epi = EPI_PLAIN;
    a.w = (const unsigned char *)packed; a.w_bstride = 0;
    a.bias = bias; a.bias_bstride = 0;
    a.cout = cout; a.out = y; a.groups = 8; a.nsplit = 2;
    a.omask = omask;
    const hipError_t e = launch_conv(CONV_C3, a, (hipStream_t)stream);
    if (e != hipSuccess) return tfail(GTTS_E_HIP, "conv3x3 (cin %d, cout %d): %s", cin, cout, hipGetErrorString(e));
    return GTTS_OK;
}

extern "C" int gtts_conv3x3_masked2(const float *x, const float *x1, int c0, const float *mask, const void *packed, const float *bias,
                                    float *y, int B, int cin, int cout, int H, int W, gtts_stream_t stream) {
    return gtts_conv3x3_masked3(x, x1, c0, mask, nullptr, packed, bias, y, B, cin, cout, H, W, stream);
}

extern "C" int gtts_conv3x3_masked(const float *x, const float *mask, const void *packed, const float *bias, float *y, int B,
                                   int cin, int cout, int H, int W, gtts_stream_t stream) {
    return gtts_conv3x3_masked3(x, nullptr, 0, mask, nullptr, packed, bias, y, B, cin, cout, H, W, stream);
}

// ---- 1x1 convolutions of the training path (res_conv, to_qkv, to_out: diffusion.py:70,87-88): forward and data gradient on
// the inference CONV_P1 kernel (mask prologue, plain epilogue); the weight gradient is gtts_conv1x1_wgrad (train_wgrad.hip)
extern "C" size_t gtts_conv1x1_packed_bytes(int cin, int cout) {
    if (cin <= 0 || cout <= 0) return 0;
    return (conv_packed_bytes(CONV_P1, cin, cout) + 255) / 256 * 256;
}

// transposed != 0: w is the FORWARD weight [forward cout = cin of this conv][forward cin = cout of this conv]
extern "C" int gtts_conv1x1_pack(const float *w, void *packed, int cin, int cout, int transposed, gtts_stream_t stream) {
    if (!w || !packed) return tfail(GTTS_E_NULL, "gtts_conv1x1_pack: null argument");
    if (cin <= 0 || cout <= 0) return tfail(GTTS_E_SHAPE, "gtts_conv1x1_pack: bad shape");
    TCHK(launch_pack_conv(transposed ? CONV_P1 + 16 : CONV_P1, w, (unsigned char *)packed, cin, cout, (hipStream_t)stream));
    return GTTS_OK;
}

// y = Conv2d_1x1(x * mask, packed W) + bias; x [B,cin,H,W], mask [B,W] (columns), bias [cout], y [B,cout,H,W]
extern "C" int gtts_conv1x1_masked(const float *x, const float *mask, const void *packed, const float *bias, float *y, int B,
                                   int cin, int cout, int H, int W, gtts_stream_t stream) {
    if (!x || !mask || !packed || !bias || !y) return tfail(GTTS_E_NULL, "gtts_conv1x1_masked: null argument");
    if (B <= 0 || cin <= 0 || cout <= 0 || H <= 0 || W <= 0) return tfail(GTTS_E_SHAPE, "gtts_conv1x1_masked: bad shape");
    if (cout % (cout > 64 ? 128 : 64) != 0) return tfail(GTTS_E_SHAPE, "gtts_conv1x1_masked: cout must be 64 or a multiple of 128 (got %d)", cout);
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.src0 = x; a.src1 = x; a.c0 = cin; a.c1 = 0; a.cin = cin;
    a.B = B; a.Hin = a.Hout = H; a.Win = a.Wout = W;
    a.mask = mask; a.T = W; a.lvl_in = a.lvl_out = 0;
    a.pro = PRO_MASK; a.epi = EPI_PLAIN;
    a.w = (const unsigned char *)packed; a.w_bstride = 0;
    a.bias = bias; a.bias_bstride = 0;
    a.cout = cout; a.out = y; a.groups = 8; a.nsplit = 2;
    const hipError_t e = launch_conv(CONV_P1, a, (hipStream_t)stream);
    if (e != hipSuccess) return tfail(GTTS_E_HIP, "conv1x1 (cin %d, cout %d): %s", cin, cout, hipGetErrorString(e));
    return GTTS_OK;
}

// ---- Downsample (Conv2d 3x3, stride 2, pad 1: diffusion.py:28-34) and Upsample (ConvTranspose2d 4x4, stride 2, pad 1:
// diffusion.py:19-25) of the training path on the inference kernels.  up = 0: w [cout][cin][3][3], y [B,cout,H/2,W/2];
// up = 1: w [cin][cout][4][4] (ConvTranspose2d layout), y [B,cout,2H,2W].  The data gradient of Downsample IS an Upsample call:
// a transposed 3x3 stride-2 convolution is the 4x4 one whose fourth kernel row and column are zero (same index map
// y = 2 oy - 1 + ky), so the host packs the zero-padded forward weight with up = 1.
extern "C" size_t gtts_conv_resample_packed_bytes(int cin, int cout, int up) {
    if (cin <= 0 || cout <= 0) return 0;
    return (conv_packed_bytes(up ? CONV_UP : CONV_DN, cin, cout) + 255) / 256 * 256;
}

extern "C" int gtts_conv_resample_pack(const float *w, void *packed, int cin, int cout, int up, gtts_stream_t stream) {
    if (!w || !packed) return tfail(GTTS_E_NULL, "gtts_conv_resample_pack: null argument");
    if (cin <= 0 || cout <= 0) return tfail(GTTS_E_SHAPE, "gtts_conv_resample_pack: bad shape");
    TCHK(launch_pack_conv(up ? CONV_UP : CONV_DN, w, (unsigned char *)packed, cin, cout, (hipStream_t)stream));
    return GTTS_OK;
}

// ---- all weight packs of a training step in one launch (ABI 4) ----------------------------------------------------------
// gtts_pack_batch_describe fills `desc_host` (gtts_pack_batch_desc_bytes(n) bytes of HOST memory) with one launch descriptor per
// item; the caller copies the table to the device once (the weight and blob addresses of a module do not change between steps)
// and calls gtts_pack_batch(desc_dev, n, grid_x) at the top of every step: one launch, graph-capturable, no host work.
extern "C" size_t gtts_pack_batch_desc_bytes(int n) { return n > 0 ? (size_t)n * sizeof(PackDesc) : 0; }

extern "C" int gtts_pack_batch_describe(const gtts_pack_item *items, int n, void *desc_host, int *grid_x) {
    if (!items || !desc_host || !grid_x) return tfail(GTTS_E_NULL, "gtts_pack_batch_describe: null argument");
    if (n <= 0) return tfail(GTTS_E_SHAPE, "gtts_pack_batch_describe: n must be positive");
    PackDesc *d = reinterpret_cast<PackDesc *>(desc_host);
    size_t mx = 0;
    for (int k = 0; k < n; ++k) {
        const gtts_pack_item &it = items[k];
        if (!it.w || !it.packed || it.cin <= 0 || it.cout <= 0) return tfail(GTTS_E_SHAPE, "gtts_pack_batch_describe: bad item %d", k);
        int mode;
        switch (it.kind) {
            case 0: mode = it.transposed ? CONV_C3 + 16 : CONV_C3; break;
            case 1: mode = it.transposed ? CONV_P1 + 16 : CONV_P1; break;
            case 2: mode = CONV_DN; break;
            case 3: mode = CONV_UP; break;
            case 4: mode = CONV_UP + 16; break;       // Downsample's data gradient: the 3x3 forward weight as a zero-padded 4x4 transposed conv
            default: return tfail(GTTS_E_SHAPE, "gtts_pack_batch_describe: unknown kind %d", it.kind);
        }
        memset(&d[k], 0, sizeof(PackDesc));
        pack_describe(mode, it.w, it.packed, it.cin, it.cout, &d[k]);
        mx = std::max(mx, d[k].total);
    }
    *grid_x = (int)std::min<size_t>(std::max<size_t>((mx + 1023) / 1024, 1), 256);       // ~4 element pairs per thread on the largest weight
    return GTTS_OK;
}

extern "C" int gtts_pack_batch(const void *desc_dev, int n, int grid_x, gtts_stream_t stream) {
    if (!desc_dev) return tfail(GTTS_E_NULL, "gtts_pack_batch: null argument");
    if (n <= 0 || grid_x <= 0) return tfail(GTTS_E_SHAPE, "gtts_pack_batch: bad sizes");
    TCHK(launch_pack_batch(reinterpret_cast<const PackDesc *>(desc_dev), n, grid_x, (hipStream_t)stream));
    return GTTS_OK;
}

// y = conv(x * mask) + bias; x [B,cin,H,W], mask [B,W] (columns of the INPUT).  H and W even for up = 0.
extern "C" int gtts_conv_resample(const float *x, const float *mask, const void *packed, const float *bias, float *y, int B, int cin,
                                  int cout, int H, int W, int up, gtts_stream_t stream) {
    if (!x || !mask || !packed || !bias || !y) return tfail(GTTS_E_NULL, "gtts_conv_resample: null argument");
    if (B <= 0 || cin <= 0 || cout <= 0 || H <= 0 || W <= 0) return tfail(GTTS_E_SHAPE, "gtts_conv_resample: bad shape");
    if (!up && ((H | W) & 1)) return tfail(GTTS_E_SHAPE, "gtts_conv_resample: Downsample needs even H and W (got %d x %d)", H, W);
    if (cin % 16 != 0 || cout % (cout > 64 ? 128 : 64) != 0)
        return tfail(GTTS_E_SHAPE, "gtts_conv_resample: cin must be a multiple of 16 and cout 64 or a multiple of 128 (got %d, %d)", cin, cout);
    ConvArgs a;
    memset(&a, 0, sizeof(a));
    a.src0 = x; a.src1 = x; a.c0 = cin; a.c1 = 0; a.cin = cin;
    a.B = B; a.Hin = H; a.Win = W;
    a.Hout = up ? 2 * H : H / 2; a.Wout = up ? 2 * W : W / 2;
    a.mask = mask; a.T = W; a.lvl_in = 0; a.lvl_out = 0;
    a.pro = PRO_MASK; a.epi = EPI_PLAIN;
    a.w = (const unsigned char *)packed; a.w_bstride = 0;
    a.bias = bias; a.bias_bstride = 0;
    a.cout = cout; a.out = y; a.groups = 8; a.nsplit = 2;
    const hipError_t e = launch_conv(up ? CONV_UP : CONV_DN, a, (hipStream_t)stream);
    if (e != hipSuccess) return tfail(GTTS_E_HIP, "conv resample (cin %d, cout %d, up %d): %s", cin, cout, up, hipGetErrorString(e));
    return GTTS_OK;
}

extern "C" int gtts_conv3x3_wgrad(const float *x, const float *mask, const float *dy, float *dw, float *db, int B, int cin, int cout,
                                  int H, int W, gtts_stream_t stream) {
    if (!x || !dy || !dw) return tfail(GTTS_E_NULL, "gtts_conv3x3_wgrad: null argument");
    if (B <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0 || cin % 32 || cout % 32)
        return tfail(GTTS_E_SHAPE, "gtts_conv3x3_wgrad: cin and cout must be multiples of 32 (got %d, %d)", cin, cout);
    hipStream_t st = (hipStream_t)stream;
    TCHK(hipMemsetAsync(dw, 0, (size_t)cout * cin * 9 * 4, st));
    if (db) TCHK(hipMemsetAsync(db, 0, (size_t)cout * 4, st));
    WgradArgs a;
    a.x = x; a.mask = mask; a.dy = dy; a.dw = dw; a.db = db;
    a.B = B; a.cin = cin; a.cout = cout; a.H = H; a.W = W;
    a.nblk = B * H * ((W + 15) / 16);
    const int tiles = (cin / 32) * (cout / 32);
    // ~8 waves per SIMD-pair worth of work across the chip, at least 8 pixel blocks per wave
    int nslice = (256 * 8 * 4 + tiles - 1) / tiles;
    nslice = std::max(1, std::min(nslice, (a.nblk + 7) / 8));
    a.nslice = nslice;
    const long long waves = (long long)tiles * nslice;
    hipLaunchKernelGGL(conv3x3_wgrad_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, a);
    TCHK(hipGetLastError());
    return GTTS_OK;
}
