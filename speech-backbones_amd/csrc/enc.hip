// enc.hip -- the once-per-utterance encoders either side of the sampling path:
//   Grad-TTS TextEncoder        Grad-TTS/model/text_encoder.py:281-326  (embedding, ConvReluNorm prenet :30-61, 6-layer
//                                relative-position transformer Encoder :240-278 / MultiHeadAttention :100-205 / FFN :208-237,
//                                proj_m, DurationPredictor :64-97)
//   DiffVC MelEncoder           DiffVC/model/encoder.py:257-284 (init_proj, the same prenet and Encoder, term_proj)
// Every Conv1d (k = 1, 3, 5) is the shared 1-D MFMA kernel of conv1d.h (split-bf16, fp32 accumulate) with the `x * x_mask`
// and ReLU prologues, residual and mask epilogues; LayerNorm over channels and the windowed relative-position attention
// are fp32 VALU kernels (the whole encoder is ~4 GFLOP per utterance, 3 % of ONE decoder call: latency, not throughput).
// Inference only (dropout is the identity in eval mode); layout [B][C][t] fp32, t fastest.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/gradtts_abi.h"
#include "common.h"
#include "conv1d.h"
#include "kernels.h"

namespace gtts {

// x[b][c][t] = emb[ids[b][t]][c] * sqrt(C)      (text_encoder.py:311-312)
__global__ void enc_embed_kernel(const long long *__restrict__ ids, const float *__restrict__ emb, float *__restrict__ x, int C,
                                 int L, int n_vocab, float scale) {
    const int b = blockIdx.z, c = blockIdx.y, t = blockIdx.x * 256 + threadIdx.x;
    if (t >= L) return;
    long long id = ids[(size_t)b * L + t];
    id = id < 0 ? 0 : (id >= n_vocab ? n_vocab - 1 : id);
    x[((size_t)b * C + c) * L + t] = emb[(size_t)id * C + c] * scale;
}

// LayerNorm over the channel axis per position (text_encoder.py:12-27: eps 1e-4, biased variance, two-pass like the
// reference: mean, then mean of squared deviations):  out = LN(relu?(a * a_mask?) + b?) ; relu? ; * out_mask?
// Workgroup = 16 positions x 16 channel lanes (lane = 16 * channel lane + position: a wave's loads are four 64-byte runs); every
// thread walks C / 16 channels per pass and the three passes (sum, squared deviations, normalise) are reduced across the channel
// lanes with two shuffles + one LDS exchange.  (Round 3's form -- one thread per position walking all C channels three times, 3 x C
// dependent strided loads -- took 155 us per call at C = 192, L = 180 and was 54 % of the text encoder's time at B = 1.)
__global__ __launch_bounds__(256) void enc_layernorm_kernel(const float *__restrict__ a, const float *__restrict__ bres,
                                                            const float *__restrict__ gamma, const float *__restrict__ beta,
                                                            const float *__restrict__ a_mask, const float *__restrict__ out_mask,
                                                            float *__restrict__ out, int C, int L, float eps, int relu_in, int relu_out) {
    __shared__ float s_red[2][4][16];
    const int b = blockIdx.y, tl = threadIdx.x & 15, cl = threadIdx.x >> 4, wave = threadIdx.x >> 6;
    const int t = blockIdx.x * 16 + tl;
    const bool ok = t < L;
    const int tc = ok ? t : L - 1;                          // (clamped: out-of-range lanes compute on valid memory, store nothing)
    const size_t base = (size_t)b * C * L + tc;
    const float am = a_mask ? a_mask[(size_t)b * L + tc] : 1.f;
    auto val = [&](int c) {
        float v = a[base + (size_t)c * L];
        if (relu_in) v = fmaxf(v, 0.f);
        v *= am;
        if (bres) v += bres[base + (size_t)c * L];
        return v;
    };
    auto reduce16 = [&](float x, int slot) {                 // sum over the 16 channel lanes of this thread's position
        x += __shfl_xor(x, 16, 64);
        x += __shfl_xor(x, 32, 64);
        if ((threadIdx.x & 48) == 0) s_red[slot][wave][tl] = x;
        __syncthreads();
        return (s_red[slot][0][tl] + s_red[slot][1][tl]) + (s_red[slot][2][tl] + s_red[slot][3][tl]);
    };
    float s = 0.f;
    for (int c = cl; c < C; c += 16) s += val(c);
    const float mean = reduce16(s, 0) / (float)C;
    float q = 0.f;
    for (int c = cl; c < C; c += 16) {
        const float d = val(c) - mean;
        q = fmaf(d, d, q);
    }
    const float rstd = rsqrtf(reduce16(q, 1) / (float)C + eps);
    const float om = out_mask ? out_mask[(size_t)b * L + tc] : 1.f;
    if (!ok) return;
    for (int c = cl; c < C; c += 16) {
        float y = (val(c) - mean) * rstd * gamma[c] + beta[c];
        if (relu_out) y = fmaxf(y, 0.f);
        out[base + (size_t)c * L] = y * om;
    }
}

// MultiHeadAttention.attention (text_encoder.py:145-175) for self-attention with a relative-position window w:
//   score[i][j] = (q_i . k_j + [|j-i| <= w] q_i . Ek[j-i+w]) / sqrt(dk);  masked_fill(mask_i * mask_j == 0, -1e4);
//   p = softmax_j;  out_i = sum_j p_ij v_j + sum_{|j-i|<=w} p_ij Ev[j-i+w]
// (the reference pads the 2w+1 embeddings with zeros to 2t-1 relative positions and re-indexes them with pad/reshape tricks;
// only |j-i| <= w survives).  One workgroup per (query tile of 8, head, sample); a wave handles 2 queries.
constexpr int ATT_QT = 8;
__global__ __launch_bounds__(256) void enc_attention_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                            const float *__restrict__ v, const float *__restrict__ mask,
                                                            const float *__restrict__ ek, const float *__restrict__ ev,
                                                            float *__restrict__ out, int C, int L, int heads, int win) {
    extern __shared__ float sm[];          // [ATT_QT][dk] queries, [ATT_QT][L] probabilities
    const int dk = C / heads;
    float *s_q = sm, *s_p = sm + ATT_QT * dk;
    const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * ATT_QT;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const size_t hb = ((size_t)b * C + (size_t)h * dk) * L;
    const float *qb = q + hb, *kb = k + hb, *vb = v + hb, *mb = mask + (size_t)b * L;
    const float inv = 1.0f / sqrtf((float)dk);
    for (int e = tid; e < ATT_QT * dk; e += 256) {
        const int qi = e / dk, d = e - qi * dk;
        s_q[e] = i0 + qi < L ? qb[(size_t)d * L + i0 + qi] : 0.f;
    }
    __syncthreads();
    for (int qq = 0; qq < 2; ++qq) {
        const int qi = wave * 2 + qq, i = i0 + qi;
        if (i >= L) continue;                                     // wave-uniform
        const float *qv = s_q + qi * dk;
        float *pv = s_p + (size_t)qi * L;
        const float mi = mb[i];
        float mx = -INFINITY;
        for (int j = lane; j < L; j += 64) {
            float acc = 0.f;
            for (int d = 0; d < dk; ++d) acc = fmaf(qv[d], kb[(size_t)d * L + j], acc);
            float sc = acc * inv;
            const int r = j - i + win;
            if (r >= 0 && r <= 2 * win) {
                float ar = 0.f;
                for (int d = 0; d < dk; ++d) ar = fmaf(qv[d], ek[r * dk + d], ar);
                sc = sc + ar * inv;
            }
            if (mi * mb[j] == 0.f) sc = -1e4f;
            pv[j] = sc;
            mx = fmaxf(mx, sc);
        }
        mx = wave_max(mx);
        float sum = 0.f;
        for (int j = lane; j < L; j += 64) {
            const float e = expf(pv[j] - mx);
            pv[j] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        const float rs = 1.0f / sum;
        for (int j = lane; j < L; j += 64) pv[j] *= rs;
    }
    __syncthreads();
    // out[d][i] = sum_j p[i][j] v[d][j] (+ relative values): thread = (query, channel) pairs
    for (int e = tid; e < ATT_QT * dk; e += 256) {
        const int qi = e / dk, d = e - qi * dk, i = i0 + qi;
        if (i >= L) continue;
        const float *pv = s_p + (size_t)qi * L;
        const float *vr = vb + (size_t)d * L;
        float acc = 0.f;
        for (int j = 0; j < L; ++j) acc = fmaf(pv[j], vr[j], acc);
        for (int r = 0; r <= 2 * win; ++r) {
            const int j = i + r - win;
            if (j >= 0 && j < L) acc = fmaf(pv[j], ev[r * dk + d], acc);
        }
        out[hb + (size_t)d * L + i] = acc;
    }
}

// ---- round 4: the same attention with LANES ALONG THE KEYS (coalesced K / V reads) and 16 queries per workgroup.
// The kernel above reads a K element once per query (8 x per workgroup) and walks V with one row per lane (each lane its own
// cache line): 2.1 ms per call for the DiffVC MelEncoder at L = 1024, B = 16 -- half of the average-voice encoder.  Here a K or V
// element is loaded once per workgroup and used for all 16 queries:
//   scores   lane = key j, 16 accumulators (one per query), the 16 query values of channel d as four broadcast 16-byte LDS reads;
//   softmax  one wave per four query rows (unchanged arithmetic: max, exp, sum, scale);
//   output   lane = key j again: four channels x 16 queries of accumulators per pass, wave sums at the end, relative-value terms
//            added by the 16 lanes that store.
// Same formulas and the same fp32 operation order per element as above up to the association of the two long sums.
constexpr int ATT16_QT = 16;
static inline size_t att16_smem_bytes(int dk, int L) {
    const int LP = (L + 63) / 64 * 64;
    return ((size_t)dk * ATT16_QT + ATT16_QT * 16 + (size_t)ATT16_QT * LP + 4 * 64) * sizeof(float);
}
__global__ __launch_bounds__(256) void enc_attention16_kernel(const float *__restrict__ q, const float *__restrict__ k,
                                                              const float *__restrict__ v, const float *__restrict__ mask,
                                                              const float *__restrict__ ek, const float *__restrict__ ev,
                                                              float *__restrict__ out, int C, int L, int heads, int win) {
    constexpr int QT = ATT16_QT;
    extern __shared__ float sm[];
    const int dk = C / heads, LP = (L + 63) / 64 * 64;
    float *s_q = sm;                         // [dk][QT]
    float *s_rel = s_q + dk * QT;            // [QT][16]: q_i . Ek[r]
    float *s_p = s_rel + QT * 16;            // [QT][LP] scores, then probabilities
    float *s_o = s_p + (size_t)QT * LP;      // [4 waves][4 dd x 16 q]: totals of one output pass
    const int b = blockIdx.z, h = blockIdx.y, i0 = blockIdx.x * QT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const size_t hb = ((size_t)b * C + (size_t)h * dk) * L;
    const float *qb = q + hb, *kb = k + hb, *vb = v + hb, *mb = mask + (size_t)b * L;
    const float inv = 1.0f / sqrtf((float)dk);
    const int nr = win >= 0 ? 2 * win + 1 : 0;
    for (int e = tid; e < QT * dk; e += 256) {
        const int d = e / QT, qi = e - d * QT;
        s_q[e] = i0 + qi < L ? qb[(size_t)d * L + i0 + qi] : 0.f;
    }
    __syncthreads();
    for (int e = tid; e < QT * nr; e += 256) {
        const int qi = e / nr, r = e - qi * nr;
        float ar = 0.f;
        for (int d = 0; d < dk; ++d) ar = fmaf(s_q[d * QT + qi], ek[r * dk + d], ar);
        s_rel[qi * 16 + r] = ar;
    }
    __syncthreads();
    const int nblk = LP / 64;
    // ---- scores
    for (int jb = wave; jb < nblk; jb += 4) {
        const int j = jb * 64 + lane;
        const bool valid = j < L;
        const int jc = valid ? j : L - 1;
        float acc[QT];
#pragma unroll
        for (int qi = 0; qi < QT; ++qi) acc[qi] = 0.f;
        for (int d = 0; d < dk; ++d) {
            const float kv = kb[(size_t)d * L + jc];
            const float4 *q4 = reinterpret_cast<const float4 *>(s_q + d * QT);
#pragma unroll
            for (int g = 0; g < QT / 4; ++g) {
                const float4 qq = q4[g];
                acc[4 * g + 0] = fmaf(qq.x, kv, acc[4 * g + 0]);
                acc[4 * g + 1] = fmaf(qq.y, kv, acc[4 * g + 1]);
                acc[4 * g + 2] = fmaf(qq.z, kv, acc[4 * g + 2]);
                acc[4 * g + 3] = fmaf(qq.w, kv, acc[4 * g + 3]);
            }
        }
        const float mj = mb[jc];
#pragma unroll
        for (int qi = 0; qi < QT; ++qi) {
            const int i = i0 + qi;
            float sc = acc[qi] * inv;
            const int r = j - i + win;
            if (win >= 0 && r >= 0 && r <= 2 * win) sc = sc + s_rel[qi * 16 + r] * inv;
            const float mi = mb[min(i, L - 1)];
            if (mi * mj == 0.f) sc = -1e4f;
            if (valid) s_p[(size_t)qi * LP + j] = sc;
        }
    }
    __syncthreads();
    // ---- softmax over the keys, one wave per four query rows
    for (int qq = 0; qq < QT / 4; ++qq) {
        const int qi = wave * (QT / 4) + qq;
        if (i0 + qi >= L) continue;                               // wave-uniform
        float *pv = s_p + (size_t)qi * LP;
        float mx = -INFINITY;
        for (int j = lane; j < L; j += 64) mx = fmaxf(mx, pv[j]);
        mx = wave_max(mx);
        float sum = 0.f;
        for (int j = lane; j < L; j += 64) {
            const float e = expf(pv[j] - mx);
            pv[j] = e;
            sum += e;
        }
        sum = wave_sum(sum);
        const float rs = 1.0f / sum;
        for (int j = lane; j < L; j += 64) pv[j] *= rs;
    }
    __syncthreads();
    // ---- out[d][i] = sum_j p[i][j] v[d][j] (+ relative values): four channels per pass and wave
    float *so = s_o + wave * 64;
    for (int dg = wave; dg * 4 < dk; dg += 4) {
        float acc[4][QT];
#pragma unroll
        for (int dd = 0; dd < 4; ++dd)
#pragma unroll
            for (int qi = 0; qi < QT; ++qi) acc[dd][qi] = 0.f;
        for (int jb = 0; jb < nblk; ++jb) {
            const int j = jb * 64 + lane;
            const bool valid = j < L;
            const int jc = valid ? j : L - 1;
            float vv[4], pp[QT];
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) {
                const int d = min(dg * 4 + dd, dk - 1);
                vv[dd] = valid ? vb[(size_t)d * L + jc] : 0.f;
            }
#pragma unroll
            for (int qi = 0; qi < QT; ++qi) pp[qi] = valid ? s_p[(size_t)qi * LP + j] : 0.f;
#pragma unroll
            for (int dd = 0; dd < 4; ++dd)
#pragma unroll
                for (int qi = 0; qi < QT; ++qi) acc[dd][qi] = fmaf(pp[qi], vv[dd], acc[dd][qi]);
        }
#pragma unroll
        for (int dd = 0; dd < 4; ++dd)
#pragma unroll
            for (int qi = 0; qi < QT; ++qi) {
                const float tot = wave_sum(acc[dd][qi]);
                if (lane == dd * QT + qi) so[lane] = tot;
            }
        // (one wave: the LDS writes above are ordered before the reads below by the wave's own program order)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront", "local");
        const int dd = lane / QT, qi = lane % QT;
        const int d = dg * 4 + dd, i = i0 + qi;
        if (d < dk && i < L) {
            float o = so[lane];
            const float *pv = s_p + (size_t)qi * LP;
            for (int r = 0; r < nr; ++r) {
                const int j = i + r - win;
                if (j >= 0 && j < L) o = fmaf(pv[j], ev[r * dk + d], o);
            }
            out[hb + (size_t)d * L + i] = o;
        }
    }
}

}  // namespace gtts

using namespace gtts;

static int efail(int code, const char *fmt, ...) {
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

struct EncParam { std::string name; int rank; int dims[4]; int kind; int cin, cout, K; size_t off; };   // kind 0 fp32 copy, 1 conv1d
struct gtts_enc {
    gtts_enc_cfg cfg;
    std::vector<EncParam> params;
    size_t blob_bytes = 0;
    int find(const std::string &n) const {
        for (size_t i = 0; i < params.size(); ++i) if (params[i].name == n) return (int)i;
        return -1;
    }
};
static size_t ealign(size_t x) { return (x + 255) / 256 * 256; }
static void enc_add(gtts_enc *e, const std::string &name, std::vector<int> dims, int kind, int cin = 0, int cout = 0, int K = 0) {
    EncParam p;
    p.name = name; p.rank = (int)dims.size(); p.kind = kind; p.cin = cin; p.cout = cout; p.K = K;
    for (int i = 0; i < 4; ++i) p.dims[i] = i < (int)dims.size() ? dims[i] : 1;
    size_t n = 1;
    for (int d : dims) n *= (size_t)d;
    p.off = e->blob_bytes;
    e->blob_bytes = ealign(e->blob_bytes + (kind == 1 ? conv1d_packed_bytes(0, cin, cout, K, 1) : n * 4));
    e->params.push_back(p);
}
static void enc_add_conv(gtts_enc *e, const std::string &name, int cin, int cout, int K) {
    enc_add(e, name + ".weight", {cout, cin, K}, 1, cin, cout, K);
    enc_add(e, name + ".bias", {cout}, 0);
}
static void enc_add_ln(gtts_enc *e, const std::string &name, int C) {
    enc_add(e, name + ".gamma", {C}, 0);
    enc_add(e, name + ".beta", {C}, 0);
}

// parameters in the reference's registration order (text_encoder.py:296-309 / DiffVC encoder.py:270-278)
extern "C" int gtts_enc_create(const gtts_enc_cfg *cfg, gtts_enc **out) {
    if (!cfg || !out) return efail(GTTS_E_NULL, "gtts_enc_create: null argument");
    if (cfg->mode != 0 && cfg->mode != 1) return efail(GTTS_E_CONFIG, "mode must be 0 (TextEncoder) or 1 (MelEncoder)");
    if (cfg->channels <= 0 || cfg->n_heads <= 0 || cfg->channels % cfg->n_heads || cfg->n_layers < 0 || cfg->kernel_size % 2 == 0 ||
        cfg->kernel_size > 11 || cfg->window_size < 0 || cfg->n_feats <= 0)
        return efail(GTTS_E_CONFIG, "unsupported encoder configuration");
    gtts_enc *e = new gtts_enc();
    e->cfg = *cfg;
    const int C = cfg->channels, dk = C / cfg->n_heads;
    if (cfg->mode == 0) enc_add(e, "emb.weight", {cfg->n_vocab, C}, 0);
    else enc_add_conv(e, "init_proj", cfg->n_feats, C, 1);
    for (int i = 0; i < 3; ++i) {
        char nm[64];
        snprintf(nm, sizeof nm, "prenet.conv_layers.%d", i);
        enc_add_conv(e, nm, C, C, 5);
    }
    for (int i = 0; i < 3; ++i) {
        char nm[64];
        snprintf(nm, sizeof nm, "prenet.norm_layers.%d", i);
        enc_add_ln(e, nm, C);
    }
    enc_add_conv(e, "prenet.proj", C, C, 1);
    // Encoder registers attn_layers, norm_layers_1, ffn_layers, norm_layers_2 as four ModuleLists
    for (int i = 0; i < cfg->n_layers; ++i) {
        char nm[64];
        snprintf(nm, sizeof nm, "encoder.attn_layers.%d.", i);
        const std::string p = nm;
        if (cfg->window_size > 0) {
            enc_add(e, p + "emb_rel_k", {1, 2 * cfg->window_size + 1, dk}, 0);
            enc_add(e, p + "emb_rel_v", {1, 2 * cfg->window_size + 1, dk}, 0);
        }
        enc_add_conv(e, p + "conv_q", C, C, 1);
        enc_add_conv(e, p + "conv_k", C, C, 1);
        enc_add_conv(e, p + "conv_v", C, C, 1);
        enc_add_conv(e, p + "conv_o", C, C, 1);
    }
    for (int i = 0; i < cfg->n_layers; ++i) { char nm[64]; snprintf(nm, sizeof nm, "encoder.norm_layers_1.%d", i); enc_add_ln(e, nm, C); }
    for (int i = 0; i < cfg->n_layers; ++i) {
        char nm[64];
        snprintf(nm, sizeof nm, "encoder.ffn_layers.%d.", i);
        enc_add_conv(e, std::string(nm) + "conv_1", C, cfg->filter_channels, cfg->kernel_size);
        enc_add_conv(e, std::string(nm) + "conv_2", cfg->filter_channels, C, cfg->kernel_size);
    }
    for (int i = 0; i < cfg->n_layers; ++i) { char nm[64]; snprintf(nm, sizeof nm, "encoder.norm_layers_2.%d", i); enc_add_ln(e, nm, C); }
    if (cfg->mode == 0) {
        enc_add_conv(e, "proj_m", C, cfg->n_feats, 1);
        enc_add_conv(e, "proj_w.conv_1", C, cfg->filter_channels_dp, cfg->kernel_size);
        enc_add_ln(e, "proj_w.norm_1", cfg->filter_channels_dp);
        enc_add_conv(e, "proj_w.conv_2", cfg->filter_channels_dp, cfg->filter_channels_dp, cfg->kernel_size);
        enc_add_ln(e, "proj_w.norm_2", cfg->filter_channels_dp);
        enc_add_conv(e, "proj_w.proj", cfg->filter_channels_dp, 1, 1);
    } else {
        enc_add_conv(e, "term_proj", C, cfg->n_feats, 1);
    }
    *out = e;
    return GTTS_OK;
}
extern "C" void gtts_enc_destroy(gtts_enc *e) { delete e; }
extern "C" int gtts_enc_num_params(const gtts_enc *e) { return e ? (int)e->params.size() : 0; }
extern "C" int gtts_enc_param_info(const gtts_enc *e, int i, const char **name, int *rank, int dims[4]) {
    if (!e) return efail(GTTS_E_NULL, "null encoder");
    if (i < 0 || i >= (int)e->params.size()) return efail(GTTS_E_SHAPE, "parameter index out of range");
    const EncParam &p = e->params[i];
    if (name) *name = p.name.c_str();
    if (rank) *rank = p.rank;
    if (dims) for (int k = 0; k < 4; ++k) dims[k] = p.dims[k];
    return GTTS_OK;
}
extern "C" size_t gtts_enc_packed_bytes(const gtts_enc *e) { return e ? e->blob_bytes : 0; }
extern "C" int gtts_enc_pack(const gtts_enc *e, const void *const *ptrs, int n_params, void *packed, gtts_stream_t stream) {
    if (!e || !ptrs || !packed) return efail(GTTS_E_NULL, "gtts_enc_pack: null argument");
    if (n_params != (int)e->params.size()) return efail(GTTS_E_PARAMS, "expected %d parameters, got %d", (int)e->params.size(), n_params);
    hipStream_t st = (hipStream_t)stream;
    unsigned char *blob = (unsigned char *)packed;
    ECHK(hipMemsetAsync(blob, 0, e->blob_bytes, st));
    for (int i = 0; i < n_params; ++i) {
        const EncParam &p = e->params[i];
        if (!ptrs[i]) return efail(GTTS_E_NULL, "parameter %s is null", p.name.c_str());
        if (p.kind == 1) {
            ECHK(launch_pack_conv1d((const float *)ptrs[i], blob + p.off, 0, p.cin, p.cout, p.K, 1, 0, st));
        } else {
            size_t n = 1;
            for (int k = 0; k < p.rank; ++k) n *= (size_t)p.dims[k];
            ECHK(hipMemcpyAsync(blob + p.off, ptrs[i], n * 4, hipMemcpyDeviceToDevice, st));
        }
    }
    return GTTS_OK;
}

// workspace: X, Y, Z (C channels), Q, K, V, A (C channels), H (max(filter, filter_dp) channels)
static size_t enc_slot(const gtts_enc *e, int B, int L, int ch) { return ealign((size_t)B * ch * L * 4); }
extern "C" size_t gtts_enc_workspace_bytes(const gtts_enc *e, int B, int L) {
    if (!e || B <= 0 || L <= 0) return 0;
    const int C = e->cfg.channels;
    const int Hc = std::max(std::max(e->cfg.filter_channels, e->cfg.filter_channels_dp), C);
    return 7 * enc_slot(e, B, L, C) + 2 * enc_slot(e, B, L, Hc);
}

struct EncRun {
    const gtts_enc *e;
    const unsigned char *blob;
    const float *mask;
    int B, L;
    hipStream_t st;
};
static const float *bp(const EncRun &r, const std::string &name) {
    const int i = r.e->find(name);
    return i < 0 ? nullptr : (const float *)(r.blob + r.e->params[i].off);
}
// out = conv(name)(relu?(x) * in_mask?) [+ res] [* out_mask]
static int enc_conv(const EncRun &r, const std::string &name, const float *x, float *out, bool in_mask, bool relu_in,
                    const float *res, bool out_mask) {
    const int wi = r.e->find(name + ".weight"), bi = r.e->find(name + ".bias");
    if (wi < 0 || bi < 0) return efail(GTTS_E_CONFIG, "encoder has no layer %s", name.c_str());
    const EncParam &p = r.e->params[wi];
    C1Args a;
    a.x = x; a.out = out; a.res = res; a.accsrc = nullptr; a.accmode = 0; a.div = 1.f;
    a.w = r.blob + p.off; a.bias = (const float *)(r.blob + r.e->params[bi].off);
    a.B = r.B; a.cin = p.cin; a.cout = p.cout; a.Lin = r.L; a.S = 1;
    a.slope = relu_in ? 0.f : 1.f;
    a.in_mask = in_mask ? r.mask : nullptr;
    a.out_mask = out_mask ? r.mask : nullptr;
    const hipError_t e = launch_conv1d(a, 0, p.K, 1, r.st);
    if (e != hipSuccess) return efail(GTTS_E_HIP, "conv1d %s: %s", name.c_str(), hipGetErrorString(e));
    return GTTS_OK;
}
static int enc_ln(const EncRun &r, const std::string &name, const float *a, const float *bres, float *out, int C, bool a_mask,
                  bool relu_in, bool relu_out) {
    const float *g = bp(r, name + ".gamma"), *b = bp(r, name + ".beta");
    if (!g || !b) return efail(GTTS_E_CONFIG, "encoder has no layer %s", name.c_str());
    hipLaunchKernelGGL(enc_layernorm_kernel, dim3((r.L + 15) / 16, r.B), dim3(256), 0, r.st, a, bres, g, b, a_mask ? r.mask : nullptr,
                       (const float *)nullptr, out, C, r.L, 1e-4f, relu_in ? 1 : 0, relu_out ? 1 : 0);
    ECHK(hipGetLastError());
    return GTTS_OK;
}

// TextEncoder.forward (text_encoder.py:310-326) / MelEncoder.forward (DiffVC encoder.py:279-284).
//   mode 0: ids [B,L] int64 -> mu [B,n_feats,L], logw [B,1,L];   mode 1: mel [B,n_feats,L] -> out [B,n_feats,L] (in `mu`)
extern "C" int gtts_enc_forward(const gtts_enc *e, const void *packed, const long long *ids, const float *mel, const float *x_mask,
                                float *mu, float *logw, void *workspace, size_t workspace_bytes, int B, int L, gtts_stream_t stream) {
    if (!e || !packed || !x_mask || !mu || !workspace) return efail(GTTS_E_NULL, "gtts_enc_forward: null argument");
    if (B <= 0 || L <= 0) return efail(GTTS_E_SHAPE, "gtts_enc_forward: bad shape B=%d L=%d", B, L);
    const gtts_enc_cfg &cf = e->cfg;
    if (cf.mode == 0 && (!ids || !logw)) return efail(GTTS_E_NULL, "TextEncoder needs ids and logw");
    if (cf.mode == 1 && !mel) return efail(GTTS_E_NULL, "MelEncoder needs mel");
    if (workspace_bytes < gtts_enc_workspace_bytes(e, B, L)) return efail(GTTS_E_WORKSPACE, "workspace too small");
    const int C = cf.channels, dk = C / cf.n_heads;
    if ((size_t)(ATT_QT * dk + ATT_QT * L) * 4 > 160 * 1024) return efail(GTTS_E_SHAPE, "sequence too long for the attention kernel (%d)", L);
    hipStream_t st = (hipStream_t)stream;
    EncRun r{e, (const unsigned char *)packed, x_mask, B, L, st};
    unsigned char *ws = (unsigned char *)workspace;
    const size_t sc = enc_slot(e, B, L, C);
    float *X = (float *)ws, *Y = (float *)(ws + sc), *Z = (float *)(ws + 2 * sc), *Q = (float *)(ws + 3 * sc),
          *K = (float *)(ws + 4 * sc), *V = (float *)(ws + 5 * sc), *A = (float *)(ws + 6 * sc);
    const int Hc = std::max(std::max(cf.filter_channels, cf.filter_channels_dp), C);
    float *H = (float *)(ws + 7 * sc), *H2 = (float *)(ws + 7 * sc + enc_slot(e, B, L, Hc));
    int rc;
    if (cf.mode == 0) {
        hipLaunchKernelGGL(enc_embed_kernel, dim3((L + 255) / 256, C, B), dim3(256), 0, st, ids, bp(r, "emb.weight"), X, C, L, cf.n_vocab,
                           sqrtf((float)C));
        ECHK(hipGetLastError());
    } else {
        if ((rc = enc_conv(r, "init_proj", mel, X, true, false, nullptr, false))) return rc;          // init_proj(x * x_mask)
    }
    // ---- ConvReluNorm prenet (text_encoder.py:54-61): 3 x (conv5(x * mask) -> LayerNorm -> ReLU), x_org + proj(x), * mask
    const float *cur = X;
    for (int i = 0; i < 3; ++i) {
        char nm[64], nn[64];
        snprintf(nm, sizeof nm, "prenet.conv_layers.%d", i);
        snprintf(nn, sizeof nn, "prenet.norm_layers.%d", i);
        if ((rc = enc_conv(r, nm, cur, Y, true, false, nullptr, false))) return rc;
        if ((rc = enc_ln(r, nn, Y, nullptr, Z, C, false, false, true))) return rc;
        cur = Z;
    }
    if ((rc = enc_conv(r, "prenet.proj", Z, Y, false, false, X, true))) return rc;                     // (x_org + proj(x)) * mask
    float *x = Y, *t1 = X;                                                                             // x: current activations
    // ---- Encoder (text_encoder.py:264-278)
    for (int i = 0; i < cf.n_layers; ++i) {
        char nm[64];
        snprintf(nm, sizeof nm, "encoder.attn_layers.%d.", i);
        const std::string p = nm;
        if ((rc = enc_conv(r, p + "conv_q", x, Q, true, false, nullptr, false))) return rc;            // x = x * x_mask feeds q, k, v
        if ((rc = enc_conv(r, p + "conv_k", x, K, true, false, nullptr, false))) return rc;
        if ((rc = enc_conv(r, p + "conv_v", x, V, true, false, nullptr, false))) return rc;
        const float *ek = bp(r, p + "emb_rel_k"), *ev = bp(r, p + "emb_rel_v");
        const int win = cf.window_size > 0 ? cf.window_size : -1;
        const size_t smem16 = att16_smem_bytes(dk, L);
        if (smem16 <= (size_t)160 * 1024 && dk % 4 == 0 && win <= 7) {
            // 16 queries per workgroup, lanes along the keys (two workgroups per CU up to L ~ 1000)
            ECHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&enc_attention16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem16));
            hipLaunchKernelGGL(enc_attention16_kernel, dim3((L + ATT16_QT - 1) / ATT16_QT, cf.n_heads, B), dim3(256), smem16, st, Q, K, V,
                               x_mask, ek, ev, A, C, L, cf.n_heads, win);
        } else {
            const size_t smem = (size_t)(ATT_QT * dk + ATT_QT * L) * 4;
            ECHK(hipFuncSetAttribute(reinterpret_cast<const void *>(&enc_attention_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem));
            hipLaunchKernelGGL(enc_attention_kernel, dim3((L + ATT_QT - 1) / ATT_QT, cf.n_heads, B), dim3(256), smem, st, Q, K, V, x_mask,
                               ek, ev, A, C, L, cf.n_heads, win);
        }
        ECHK(hipGetLastError());
        if ((rc = enc_conv(r, p + "conv_o", A, Z, false, false, nullptr, false))) return rc;
        snprintf(nm, sizeof nm, "encoder.norm_layers_1.%d", i);
        if ((rc = enc_ln(r, nm, x, Z, t1, C, true, false, false))) return rc;                          // LN(x * mask + y)
        snprintf(nm, sizeof nm, "encoder.ffn_layers.%d.", i);
        if ((rc = enc_conv(r, std::string(nm) + "conv_1", t1, H, true, false, nullptr, false))) return rc;
        if ((rc = enc_conv(r, std::string(nm) + "conv_2", H, Z, true, true, nullptr, true))) return rc; // conv_2(relu(h) * mask) * mask
        snprintf(nm, sizeof nm, "encoder.norm_layers_2.%d", i);
        if ((rc = enc_ln(r, nm, t1, Z, x, C, false, false, false))) return rc;                         // LN(x + y)
    }
    // x = x * x_mask (applied on load below)
    if (cf.mode == 1) return enc_conv(r, "term_proj", x, mu, true, false, nullptr, false);             // term_proj(x * x_mask)
    if ((rc = enc_conv(r, "proj_m", x, mu, true, false, nullptr, true))) return rc;                    // proj_m(x) * x_mask
    // ---- DurationPredictor (text_encoder.py:84-97)
    if ((rc = enc_conv(r, "proj_w.conv_1", x, H, true, false, nullptr, false))) return rc;
    if ((rc = enc_ln(r, "proj_w.norm_1", H, nullptr, H2, cf.filter_channels_dp, false, true, false))) return rc;
    if ((rc = enc_conv(r, "proj_w.conv_2", H2, H, true, false, nullptr, false))) return rc;
    if ((rc = enc_ln(r, "proj_w.norm_2", H, nullptr, H2, cf.filter_channels_dp, false, true, false))) return rc;
    return enc_conv(r, "proj_w.proj", H2, logw, true, false, nullptr, true);
}
