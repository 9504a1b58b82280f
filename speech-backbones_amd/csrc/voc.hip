// voc.hip -- HiFi-GAN generator (mel -> waveform), the step right after the sampling path in Grad-TTS/inference.py:81
// (SURVEY.md section 8f rank 3).  Reference: Grad-TTS/hifi-gan/models.py:77-128 (Generator), :13-75 (ResBlock1/2),
// configuration Grad-TTS/checkpts/hifigan-config.json (V1: 512 channels, rates 8,8,2,2, kernels 3/7/11, dilations 1/3/5).
//
// Everything dense is ONE 1-D implicit-GEMM kernel on the bf16 MFMA pipe (split-bf16 hi/lo, 3 MFMAs per product, fp32
// accumulate: fp32-grade like the decoder's convolutions):
//     D[row m][position q] = sum_{tap, cin} W[m][cin, tap] * lrelu(X[cin][q + off(tap)])
//   Conv1d(k, dilation d):        rows = cout, off(tap) = (tap - (k-1)/2) * d                       models.py:18-35,83
//   ConvTranspose1d(k, stride s): rows = (cout, phase r) with r = output position mod s; every phase is a 2-tap
//       convolution over the INPUT positions, the union over phases is a 3-tap convolution with per-row zero weights.
//       The C/D fragment holds 4 consecutive rows per lane = consecutive output positions of one channel.  models.py:88-91
// fused around it: LeakyReLU of the input on load (models.py:39,41,105,117), bias, the ResBlock residual `xt + x`
// (:43), the sum over the three ResBlocks and the division by num_kernels (:110-114) in the reference's fp32 order.
// Layout: [B][C][L] fp32, L fastest -> coalesced along time.  conv_post (32 -> 1 channel, tanh) is a VALU kernel.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/gradtts_abi.h"
#include "common.h"
#include "conv1d.h"
#include "kernels.h"

namespace gtts {

// ---- conv_post: Conv1d(C -> 1, k = 7) on leaky_relu(x, 0.01), then tanh  (models.py:116-118).  Four consecutive outputs per
// thread from three aligned 16-byte loads per channel (positions t0-4 .. t0+7); L % 4 == 0 (L = T * hop).
__global__ void conv_post_kernel(const float *__restrict__ x, const float *__restrict__ w, const float *__restrict__ bias,
                                 float *__restrict__ out, int C, int L, int K, float slope) {
    extern __shared__ float sw[];          // [C][K]
    for (int i = threadIdx.x; i < C * K; i += 256) sw[i] = w[i];
    __syncthreads();
    const int b = blockIdx.y;
    const int t0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (t0 >= L) return;
    const float *xb = x + (size_t)b * C * L;
    const float b0 = bias[0];
    float acc[4] = {b0, b0, b0, b0};
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = 0; c < C; ++c) {
        const float *row = xb + (size_t)c * L + t0;
        const float4 lo = t0 >= 4 ? *reinterpret_cast<const float4 *>(row - 4) : zero;
        const float4 mid = *reinterpret_cast<const float4 *>(row);
        const float4 hi = t0 + 8 <= L ? *reinterpret_cast<const float4 *>(row + 4) : zero;
        float v[12] = {lo.x, lo.y, lo.z, lo.w, mid.x, mid.y, mid.z, mid.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int i = 0; i < 12; ++i) v[i] = v[i] > 0.f ? v[i] : v[i] * slope;
        const float *wr = sw + c * K;
#pragma unroll
        for (int k = 0; k < 7; ++k) {
            const float wk = wr[k];
#pragma unroll
            for (int o = 0; o < 4; ++o) acc[o] = fmaf(wk, v[1 + o + k], acc[o]);     // position t0 + o + k - 3 = v[4 + o + k - 3]
        }
    }
    *reinterpret_cast<float4 *>(out + (size_t)b * L + t0) = make_float4(tanhf(acc[0]), tanhf(acc[1]), tanhf(acc[2]), tanhf(acc[3]));
}

}  // namespace gtts

using namespace gtts;

// ------------------------------------------------------------------------------------------------ host plan + C ABI
static int vfail(int code, const char *fmt, ...) {       // text goes to gtts_last_error() (plan.hip)
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return set_error(code, buf);
}
#define VCHK(expr)                                                                                               \
    do {                                                                                                         \
        hipError_t e_ = (expr);                                                                                  \
        if (e_ != hipSuccess) return vfail(GTTS_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

struct VocLayer {
    std::string name;
    int mode;              // 0 Conv1d, 1 ConvTranspose1d
    int cin, cout, K, dil, S, pad;
    int ntap, nst, tps, MT;
    int toff[C1_MAXTAP];
    int halo_lo, halo_hi;
    size_t w_off, b_off;   // blob offsets (packed weights, fp32 bias)
};
struct gtts_voc {
    gtts_voc_cfg cfg;
    std::vector<VocLayer> layers;       // conv_pre, then per stage: up, resblocks..., conv_post last (VALU)
    size_t blob_bytes = 0;
    size_t post_w_off = 0, post_b_off = 0;
    int post_cin = 0, post_K = 7;
    int pre = -1;
    std::vector<int> ups;               // layer index of each upsampler
    std::vector<std::vector<std::vector<int>>> rb;   // [stage][kernel][conv index in reference order] layer ids
};

static size_t valign(size_t x) { return (x + 255) / 256 * 256; }

static int voc_add_layer(gtts_voc *v, const std::string &name, int mode, int cin, int cout, int K, int dil, int S, int pad) {
    VocLayer L;
    L.name = name; L.mode = mode; L.cin = cin; L.cout = cout; L.K = K; L.dil = dil; L.S = S; L.pad = pad;
    const int real = mode == 0 ? K : 3;
    const C1Geom g = c1_geom(cout * S, real);
    L.tps = g.tps; L.MT = g.MT;
    L.nst = (real + g.tps - 1) / g.tps;
    L.ntap = L.nst * g.tps;
    int lo = 0, hi = 0;
    for (int t = 0; t < C1_MAXTAP; ++t) L.toff[t] = 0;
    for (int t = 0; t < real; ++t) {
        L.toff[t] = mode == 0 ? (t - (K - 1) / 2) * dil : t - 1;
        lo = std::min(lo, L.toff[t]);
        hi = std::max(hi, L.toff[t]);
    }
    L.halo_lo = -lo; L.halo_hi = hi;
    const size_t nchunk = (cin + 15) / 16, ncot = ((size_t)cout * S + g.MT - 1) / g.MT;
    L.w_off = v->blob_bytes;
    v->blob_bytes = valign(v->blob_bytes + nchunk * L.nst * ncot * (size_t)2 * g.tps * 2 * g.MT * 16);
    L.b_off = v->blob_bytes;
    v->blob_bytes = valign(v->blob_bytes + (size_t)cout * 4);
    v->layers.push_back(L);
    return (int)v->layers.size() - 1;
}

extern "C" int gtts_voc_create(const gtts_voc_cfg *cfg, gtts_voc **out) {
    if (!cfg || !out) return vfail(GTTS_E_NULL, "gtts_voc_create: null argument");
    if (cfg->n_ups < 1 || cfg->n_ups > 8 || cfg->n_kernels < 1 || cfg->n_kernels > 8)
        return vfail(GTTS_E_CONFIG, "unsupported number of upsamplers / resblock kernels");
    if (cfg->resblock_type != 1 && cfg->resblock_type != 2) return vfail(GTTS_E_CONFIG, "resblock must be 1 or 2");
    gtts_voc *v = new gtts_voc();
    v->cfg = *cfg;
    int ch = cfg->upsample_initial_channel;
    v->pre = voc_add_layer(v, "conv_pre", 0, cfg->n_mels, ch, 7, 1, 1, 3);
    for (int i = 0; i < cfg->n_ups; ++i) {
        const int u = cfg->upsample_rates[i], k = cfg->upsample_kernel_sizes[i];
        if (ch % 2 || k % u || (k - u) % 2 || k / u != 2 || (u & (u - 1))) {
            delete v;
            return vfail(GTTS_E_CONFIG, "upsampler %d: need kernel = 2 * rate, a power-of-two rate and even channels (k=%d, u=%d)", i, k, u);
        }
        char nm[64];
        snprintf(nm, sizeof nm, "ups.%d", i);
        v->ups.push_back(voc_add_layer(v, nm, 1, ch, ch / 2, k, 1, u, (k - u) / 2));
        ch /= 2;
        v->rb.emplace_back();
        for (int j = 0; j < cfg->n_kernels; ++j) {
            const int kk = cfg->resblock_kernel_sizes[j];
            if (kk % 2 == 0 || kk > C1_MAXTAP - 1) { delete v; return vfail(GTTS_E_CONFIG, "resblock kernel %d unsupported", kk); }
            std::vector<int> ids;
            const int rbi = i * cfg->n_kernels + j;
            const int nd = cfg->resblock_type == 1 ? 3 : 2;
            if (cfg->resblock_type == 1) {
                for (int d = 0; d < nd; ++d) {
                    snprintf(nm, sizeof nm, "resblocks.%d.convs1.%d", rbi, d);
                    ids.push_back(voc_add_layer(v, nm, 0, ch, ch, kk, cfg->resblock_dilations[j][d], 1, 0));
                }
                for (int d = 0; d < nd; ++d) {
                    snprintf(nm, sizeof nm, "resblocks.%d.convs2.%d", rbi, d);
                    ids.push_back(voc_add_layer(v, nm, 0, ch, ch, kk, 1, 1, 0));
                }
            } else {
                for (int d = 0; d < nd; ++d) {
                    snprintf(nm, sizeof nm, "resblocks.%d.convs.%d", rbi, d);
                    ids.push_back(voc_add_layer(v, nm, 0, ch, ch, kk, cfg->resblock_dilations[j][d], 1, 0));
                }
            }
            for (int id : ids)
                if (v->layers[id].halo_lo + v->layers[id].halo_hi > 128) { delete v; return vfail(GTTS_E_CONFIG, "receptive field too wide"); }
            v->rb.back().push_back(ids);
        }
    }
    v->post_cin = ch;
    v->post_w_off = v->blob_bytes;
    v->blob_bytes = valign(v->blob_bytes + (size_t)ch * 7 * 4);
    v->post_b_off = v->blob_bytes;
    v->blob_bytes = valign(v->blob_bytes + 4);
    *out = v;
    return GTTS_OK;
}

extern "C" void gtts_voc_destroy(gtts_voc *v) { delete v; }

// parameters in this order: for each MFMA layer (conv_pre, then per stage the upsampler and its ResBlock convs)
// weight, bias; then conv_post.weight, conv_post.bias.  Names are the reference's module paths (models.py).
extern "C" int gtts_voc_num_params(const gtts_voc *v) { return v ? (int)v->layers.size() * 2 + 2 : 0; }
extern "C" int gtts_voc_param_info(const gtts_voc *v, int i, const char **name, int *rank, int dims[4]) {
    if (!v) return vfail(GTTS_E_NULL, "null vocoder");
    static thread_local std::string s;
    const int n = (int)v->layers.size();
    if (i < 0 || i >= 2 * n + 2) return vfail(GTTS_E_SHAPE, "parameter index out of range");
    int d[4] = {1, 1, 1, 1}, rk = 1;
    if (i < 2 * n) {
        const VocLayer &L = v->layers[i / 2];
        if (i % 2 == 0) {
            s = L.name + ".weight"; rk = 3;
            if (L.mode == 0) { d[0] = L.cout; d[1] = L.cin; } else { d[0] = L.cin; d[1] = L.cout; }
            d[2] = L.K;
        } else { s = L.name + ".bias"; d[0] = L.cout; }
    } else if (i == 2 * n) { s = "conv_post.weight"; rk = 3; d[0] = 1; d[1] = v->post_cin; d[2] = v->post_K; }
    else { s = "conv_post.bias"; d[0] = 1; }
    if (name) *name = s.c_str();
    if (rank) *rank = rk;
    if (dims) for (int k = 0; k < 4; ++k) dims[k] = d[k];
    return GTTS_OK;
}
extern "C" size_t gtts_voc_packed_bytes(const gtts_voc *v) { return v ? v->blob_bytes : 0; }

extern "C" int gtts_voc_pack(const gtts_voc *v, const void *const *ptrs, int n_params, void *packed, gtts_stream_t stream) {
    if (!v || !ptrs || !packed) return vfail(GTTS_E_NULL, "gtts_voc_pack: null argument");
    if (n_params != gtts_voc_num_params(v)) return vfail(GTTS_E_PARAMS, "expected %d parameters, got %d", gtts_voc_num_params(v), n_params);
    hipStream_t st = (hipStream_t)stream;
    unsigned char *blob = (unsigned char *)packed;
    VCHK(hipMemsetAsync(blob, 0, v->blob_bytes, st));
    const int n = (int)v->layers.size();
    for (int li = 0; li < n; ++li) {
        const VocLayer &L = v->layers[li];
        const float *w = (const float *)ptrs[2 * li], *bb = (const float *)ptrs[2 * li + 1];
        if (!w || !bb) return vfail(GTTS_E_NULL, "parameter of %s is null", L.name.c_str());
        const int nchunk = (L.cin + 15) / 16, ncot = (L.cout * L.S + L.MT - 1) / L.MT;
        const size_t total = (size_t)nchunk * L.nst * ncot * L.tps * 2 * L.MT * 8;
        hipLaunchKernelGGL(pack_conv1d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w,
                           reinterpret_cast<__bf16 *>(blob + L.w_off), L.mode, L.cin, L.cout, L.K, L.S, L.pad, L.MT, L.nst, L.tps,
                           nchunk, ncot, total);
        VCHK(hipGetLastError());
        VCHK(hipMemcpyAsync(blob + L.b_off, bb, (size_t)L.cout * 4, hipMemcpyDeviceToDevice, st));
    }
    if (!ptrs[2 * n] || !ptrs[2 * n + 1]) return vfail(GTTS_E_NULL, "conv_post parameter is null");
    VCHK(hipMemcpyAsync(blob + v->post_w_off, ptrs[2 * n], (size_t)v->post_cin * v->post_K * 4, hipMemcpyDeviceToDevice, st));
    VCHK(hipMemcpyAsync(blob + v->post_b_off, ptrs[2 * n + 1], 4, hipMemcpyDeviceToDevice, st));
    return GTTS_OK;
}

// workspace: conv_pre output + per stage four buffers X (upsampled), T1, R (running ResBlock state), XS (sum)
static size_t voc_stage_elems(const gtts_voc *v, int B, int T, int stage, int *C, size_t *L) {
    int ch = v->cfg.upsample_initial_channel;
    size_t len = (size_t)T;
    for (int i = 0; i <= stage; ++i) { ch /= 2; len *= v->cfg.upsample_rates[i]; }
    if (C) *C = ch;
    if (L) *L = len;
    return (size_t)B * ch * len;
}
extern "C" size_t gtts_voc_workspace_bytes(const gtts_voc *v, int B, int T) {
    if (!v || B <= 0 || T <= 0) return 0;
    size_t big = (size_t)B * v->cfg.upsample_initial_channel * T;
    for (int i = 0; i < v->cfg.n_ups; ++i) big = std::max(big, voc_stage_elems(v, B, T, i, nullptr, nullptr));
    return 5 * valign(big * 4);
}
extern "C" int gtts_voc_hop(const gtts_voc *v) {
    if (!v) return 0;
    int hop = 1;
    for (int i = 0; i < v->cfg.n_ups; ++i) hop *= v->cfg.upsample_rates[i];
    return hop;
}

static int voc_run_layer(const gtts_voc *v, const unsigned char *blob, int li, const float *x, float *out, const float *res,
                         const float *accsrc, int accmode, float slope, int B, int Lin, hipStream_t st) {
    const VocLayer &L = v->layers[li];
    C1Args a;
    a.x = x; a.out = out; a.res = res; a.accsrc = accsrc; a.w = blob + L.w_off; a.bias = (const float *)(blob + L.b_off);
    a.B = B; a.cin = L.cin; a.cout = L.cout; a.Lin = Lin; a.S = L.S;
    a.nchunk = (L.cin + 15) / 16; a.nst = L.nst;
    for (int t = 0; t < C1_MAXTAP; ++t) a.toff[t] = L.toff[t];
    a.halo_lo = L.halo_lo;
    const C1Geom g = c1_geom(L.cout * L.S, L.mode == 0 ? L.K : 3);
    a.npx = g.NT + L.halo_lo + L.halo_hi;
    a.slope = slope; a.accmode = accmode; a.div = (float)v->cfg.n_kernels;
    a.in_mask = nullptr; a.out_mask = nullptr;
    a.ls = 0;
    while ((1 << a.ls) < L.S) ++a.ls;
    if ((size_t)L.cout * Lin * L.S >= ((size_t)1 << 31)) return vfail(GTTS_E_SHAPE, "%s: tensor too large", L.name.c_str());
    const hipError_t e = L.tps == 3 ? launch_c1_t<3>(a, st) : launch_c1_t<4>(a, st);
    if (e != hipSuccess) return vfail(GTTS_E_HIP, "conv1d %s: %s", L.name.c_str(), hipGetErrorString(e));
    return GTTS_OK;
}

// Generator.forward (models.py:103-120): mel [B, n_mels, T] -> wav [B, 1, T * hop]
extern "C" int gtts_voc_forward(const gtts_voc *v, const void *packed, const float *mel, float *wav, void *workspace,
                                size_t workspace_bytes, int B, int T, gtts_stream_t stream) {
    if (!v || !packed || !mel || !wav || !workspace) return vfail(GTTS_E_NULL, "gtts_voc_forward: null argument");
    if (B <= 0 || T <= 0) return vfail(GTTS_E_SHAPE, "gtts_voc_forward: bad shape B=%d T=%d", B, T);
    const size_t need = gtts_voc_workspace_bytes(v, B, T);
    if (workspace_bytes < need) return vfail(GTTS_E_WORKSPACE, "workspace too small: need %zu bytes, got %zu", need, workspace_bytes);
    hipStream_t st = (hipStream_t)stream;
    const unsigned char *blob = (const unsigned char *)packed;
    const size_t slot = need / 5;
    float *buf[5];
    for (int i = 0; i < 5; ++i) buf[i] = (float *)((unsigned char *)workspace + i * slot);
    const float LR = 0.1f;                                        // LRELU_SLOPE (models.py:10)
    int rc = voc_run_layer(v, blob, v->pre, mel, buf[0], nullptr, nullptr, 0, 1.0f, B, T, st);     // conv_pre (:104)
    if (rc) return rc;
    float *cur = buf[0];                                          // stage input
    size_t len = (size_t)T;
    const int nk = v->cfg.n_kernels;
    for (int i = 0; i < v->cfg.n_ups; ++i) {
        // four work buffers that are not `cur`
        float *w4[4];
        for (int k = 0, j = 0; k < 5; ++k) if (buf[k] != cur) w4[j++] = buf[k];
        float *X = w4[0], *T1 = w4[1], *R = w4[2], *XS = w4[3];
        // x = ups[i](leaky_relu(x, 0.1))  (:105-106).  From stage 1 on `cur` holds the ResBlock mean already.
        rc = voc_run_layer(v, blob, v->ups[i], cur, X, nullptr, nullptr, 0, LR, B, (int)len, st);
        if (rc) return rc;
        len *= v->cfg.upsample_rates[i];
        for (int j = 0; j < nk; ++j) {
            const std::vector<int> &ids = v->rb[i][j];
            // xs = r0; xs += r1; x = (xs + r2) / num_kernels   (a single ResBlock: x = r0 / 1 = r0)
            const int accmode = j == 0 ? 0 : (j + 1 < nk ? 1 : 2);
            if (v->cfg.resblock_type == 1) {
                // ResBlock1 (:37-44): three times  xt = c2(lrelu(c1(lrelu(x))));  x = xt + x
                for (int d = 0; d < 3; ++d) {
                    const float *xin = d == 0 ? X : R;
                    rc = voc_run_layer(v, blob, ids[d], xin, T1, nullptr, nullptr, 0, LR, B, (int)len, st);
                    if (rc) return rc;
                    const bool last = d == 2;
                    rc = voc_run_layer(v, blob, ids[3 + d], T1, last ? XS : R, xin, last && accmode ? XS : nullptr,
                                       last ? accmode : 0, LR, B, (int)len, st);
                    if (rc) return rc;
                }
            } else {
                // ResBlock2 (:66-71): twice  x = c(lrelu(x)) + x
                for (int d = 0; d < 2; ++d) {
                    const float *xin = d == 0 ? X : R;
                    const bool last = d == 1;
                    rc = voc_run_layer(v, blob, ids[d], xin, last ? XS : R, xin, last && accmode ? XS : nullptr,
                                       last ? accmode : 0, LR, B, (int)len, st);
                    if (rc) return rc;
                }
            }
        }
        cur = XS;
    }
    // x = tanh(conv_post(leaky_relu(x)))  (:116-118; default slope 0.01)
    const int C = v->post_cin;
    if (v->post_K != 7 || len % 4 != 0) return vfail(GTTS_E_CONFIG, "conv_post needs k = 7 and a multiple-of-4 length");
    hipLaunchKernelGGL(conv_post_kernel, dim3((unsigned)((len / 4 + 255) / 256), B), dim3(256), (size_t)C * v->post_K * 4, st, cur,
                       (const float *)(blob + v->post_w_off), (const float *)(blob + v->post_b_off), wav, C, (int)len, v->post_K, 0.01f);
    VCHK(hipGetLastError());
    return GTTS_OK;
}
