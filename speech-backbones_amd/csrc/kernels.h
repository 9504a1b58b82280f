// kernels.h -- host-visible launchers of every kernel in libgradtts_gfx950 (internal header).
#pragma once
#include "common.h"

namespace gtts {

// error text of the calling thread (plan.hip; returned by gtts_last_error)
int set_error(int code, const char *msg);

// ---- misc.hip
struct TimeMlpDesc {
    int dim;                 // time embedding width (dec_dim)
    size_t w0, b0, w2, b2;   // blob byte offsets of mlp.0 / mlp.2 (fp32, reference layout [out][in])
    int n;                   // number of ResnetBlocks
    int cout[32];
    size_t w[32], b[32];     // blob byte offsets of <R>.mlp.1.{weight,bias}
    int off[32];             // column of the block's bias vector inside a tb row
    int temb_off;            // column where the time embedding t = mlp(emb) (dim floats) is stored
    int semb_off;            // column where the raw sinusoidal embedding (dim floats) is stored, -1: not stored
    int tb_stride;
};

// one reverse step of the DiffVC sampler (DiffVC/model/diffusion.py:177-195); mode 0 = Grad-TTS Euler(-Maruyama)
struct VcStep {
    int mode;                // 1: 'pf', 2: 'em' / 'ml'
    float cm;                // 0.5*beta*h + omega
    float k1;                // 1 + kappa
    float bh;                // beta*h
    float sigma;
};

hipError_t launch_prep_input(const float *mu, const float *x, const float *s, void *x0, int B, int F, int T,
                             int nch, hipStream_t st, int act_bf16 = 0);
hipError_t launch_spk_mlp(const float *spk, const float *w0, const float *b0, const float *w2, const float *b2,
                          float *s, int B, int E, int F, hipStream_t st);
hipError_t launch_time_mlp(const float *t, const float *freq, float pe_scale, const unsigned char *blob,
                           const TimeMlpDesc &d, float *tb, int rows, hipStream_t st);
hipError_t launch_gn_finalize(const float *partials, int nparts, int groups, int C, int HW, const float *gamma,
                              const float *beta, float *sc, float *sh, int B, hipStream_t st);
hipError_t launch_tail_identity(const void *h, const void *x, const float *sc, const float *sh, const float *mask,
                                void *out, int B, int C, int H, int W, int T, int lvl, hipStream_t st, int act_bf16 = 0);
hipError_t launch_euler_step(float *xt, const float *mu, const float *est, const float *mask, const float *noise,
                             float beta, float h, int B, int F, int T, hipStream_t st);
hipError_t launch_mul_mask(const float *z, const float *mask, float *out, int B, int F, int T, hipStream_t st);
hipError_t launch_final_euler(const void *raw, const float *sc, const float *sh, const float *w, const float *bias,
                              const float *mask, int B, int C, int F, int T, float *est_out, float *xt, const float *mu,
                              const float *noise, float beta, float h, hipStream_t st, const VcStep *vc = nullptr,
                              int act_bf16 = 0);

// ---- vc.hip  (DiffVC-only pieces: RefBlock statistics / pooling, condition MLP, input assembly)
hipError_t launch_xt_ref(const float *ref, const float *mean_ref, const float *ref_mask, float *out, float w0, float w1,
                         int B, int F, int Tr, hipStream_t st);
hipError_t launch_instnorm_stats(const float *x, const float *gamma, const float *beta, float *sc, float *sh, int B, int C,
                                 int HW, hipStream_t st);
hipError_t launch_ref_pool(const float *raw, const float *sc, const float *sh, const float *ref_mask, float *S, int B,
                           int Ch, int F, int Tr, hipStream_t st);
hipError_t launch_vc_cond(const float *tb, int tb_stride, int semb_off, int dim, const float *S, const float *ref_mask,
                          const float *fw, const float *fb, const float *c, const float *w0, const float *b0,
                          const float *w2, const float *b2, float *cond, int B, int Ch, int dim_cond, int cdim, int F, int Tr,
                          int use_ref, hipStream_t st);
hipError_t launch_prep_vc(const float *mean, const float *x, const float *cond, float *x0, int B, int F, int T, int ncond,
                          hipStream_t st);

// ---- conv_mfma.hip
// GroupNorm partial slots per sample written by EPI_STATS for this layer (batch-size independent)
int conv_nparts(int mode, int cout, int Hout, int Wout);
bool conv_rowpair_stats(int mode, int cout, int Hout, int Wout);
bool conv_small_tiles(int mode, int cout, int Hout, int Wout, int B);

// ---- attn.hip  (LinearAttention, diffusion.py:82-100, folded: see attn.hip header)
// heads per workgroup of attn_ctx_kernel (C >= 128): two heads share one staged x tile.  A build-time choice that appears in the
// kernel's template arguments, hence in the names gtts_plan_op_info reports for the rocprofv3 / traffic.json joins.
#ifndef GTTS_ATTN_HPW
#define GTTS_ATTN_HPW 2
#endif
constexpr int ATTN_KCH = 2;                 // 16-channel chunks per LDS stage of the k/v projection
constexpr int ATTN_REC = 32 + 32 + 32 * 32; // floats per partial record: m[32], Z[32], ctx[32][32]
struct AttnGeom {
    int tiles;          // pixel tiles per sample (256 pixels; 64 in the head-per-wave kernel)
    int tps;            // tiles per slice (one workgroup walks a slice)
    int nslices;
    int nrec;           // partial records per (sample, head) = nslices
};
// C == 64 (level-0 and the last up-level attention, half of all attention work) runs the head-per-wave kernel: a workgroup
// owns a slice of 64-pixel tiles for ALL FOUR heads (wave = head), so x is loaded and split once instead of four times.
static inline bool attn_head_per_wave(int C) { return C == 64; }
static inline AttnGeom attn_geom(int HW, int C) {
    AttnGeom g;
    if (attn_head_per_wave(C)) {
        g.tiles = (HW + 63) / 64;
        int t = g.tiles / 64;
        g.tps = t < 1 ? 1 : (t > 64 ? 64 : t);
    } else {
        g.tiles = (HW + 255) / 256;
        int t = g.tiles / 16;
        g.tps = t < 1 ? 1 : (t > 16 ? 16 : t);
    }
    g.nslices = (g.tiles + g.tps - 1) / g.tps;
    g.nrec = g.nslices;
    return g;
}
static inline size_t attn_kv_packed_bytes(int C) {      // [head 4][stage][split 2][kg 2*KCH][64][8] bf16
    size_t nstage = (C + 16 * ATTN_KCH - 1) / (16 * ATTN_KCH);
    return 4 * nstage * 2 * (2 * ATTN_KCH) * 64 * 16;
}
// tail (nullable; C == 64, fp32 storage): the ResnetBlock's identity tail fused into the context pass -- x is then the tail's OUTPUT
// tensor, which this launch writes: x[c][n] = xin[c][n] * m + Mish(h[c][n] * esc[c] + esh[c]) * m (tail_identity_kernel's arithmetic)
struct AttnTail {
    const void *h, *xin;        // raw output of the block's second convolution; the block's input (the residual)
    const float *esc, *esh;     // [B][C] GroupNorm scale / shift of h
    const float *mask;          // [B][T]
    int W, T, lvl;
};
hipError_t launch_attn_ctx(const void *x, const unsigned char *wkv, float *partials, int B, int C, int HW, int nsplit,
                           hipStream_t st, int act_bf16 = 0, const AttnTail *tail = nullptr);
hipError_t launch_attn_merge(const float *partials, float *ctxn, int B, int nrec, hipStream_t st);
// wq [128][C], wout [C][128], bout [C], g [1] fp32 (reference layouts) -> per-sample packed 1x1 weights + bias
hipError_t launch_attn_fold(const float *ctxn, const float *wq, const float *wout, const float *bout, const float *g,
                            unsigned char *wpk, size_t wpk_bstride, float *biasb, int B, int C, hipStream_t st);

// ---- pack.hip
// status / tag: range record of the f16 + fp8 format (mode CONV_C3 | 32), see pack.hip; nullptr: no record
hipError_t launch_pack_conv(int mode, const float *w, unsigned char *dst, int cin, int cout, hipStream_t st, unsigned *status = nullptr, unsigned tag = 0);
// one descriptor per convolution of a batched pack (device-resident table; 64 bytes)
struct PackDesc {
    const float *w;
    void *dst;
    size_t total;            // (hi, lo) element pairs
    int mode, cin, cout, MT, nst, tps, nchunk, ncot, nkg, pad_;
};
void pack_describe(int mode, const float *w, void *dst, int cin, int cout, PackDesc *out);
hipError_t launch_pack_batch(const PackDesc *descs_dev, int n, int grid_x, hipStream_t st);
hipError_t launch_pack_attn_kv(const float *wqkv, unsigned char *dst, int C, hipStream_t st);
hipError_t launch_copy_f32(const float *src, float *dst, size_t n, hipStream_t st);

// ---- glue.hip (generate_path + aligned prior mean + terminal sample: tts.py:84-94, utils.py:26-39)
hipError_t launch_expand_alignment(const float *dur, const float *x_mask, const int *y_len, const float *mu_x,
                                   const float *noise, float temperature, float *attn, float *mu_y, float *z, int B, int F,
                                   int tx, int T, hipStream_t st);

hipError_t launch_log_prior(const float *mu_x, const float *y, float *out, int B, int F, int tx, int T, hipStream_t st);

// ---- mas.hip
hipError_t launch_mas(const float *value, const float *mask, const int *t_x, const int *t_y, int *path,
                      unsigned char *scratch, int b, int tx, int ty, hipStream_t st);

}  // namespace gtts
