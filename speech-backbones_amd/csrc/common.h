// common.h -- shared device helpers and host-side launch descriptors (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>

namespace gtts {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;   // 16-byte LDS / global unit

// ---------------------------------------------------------------------------------------------------
// Mish(x) = x * tanh(softplus(x)),  torch softplus: x > 20 -> x         (Grad-TTS/model/diffusion.py:16-18)
// tanh(log(1+e^x)) = ((1+e^x)^2 - 1) / ((1+e^x)^2 + 1) = n / (n + 2),  n = e^x (e^x + 2): one exp, one rcp,
// no cancellation for very negative x (n -> 2 e^x).  For x > 20 tanh(softplus) == 1 in fp32.
// Once n > 2^25 (x > ~8.7) n + 2 == n in fp32 and the ratio is 1 to an ulp, so clamping the exponent argument
// (no overflow: e^40 squared is 5e34) replaces the reference's x > 20 branch without a compare/select.
// Workgroup barrier that orders LDS traffic only.  __syncthreads() carries a full workgroup fence, which hipcc
// lowers to s_waitcnt vmcnt(0) lgkmcnt(0): every global prefetch in flight is drained at every barrier.  With the
// fence restricted to the local address space only lgkmcnt(0) is waited for and loads stay in flight across it.
// Use it where the data exchanged through the barrier lives in LDS (all main loops here).
__device__ __forceinline__ void lds_barrier() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}

// Fused GroupNorm finalize hand-off (conv_mfma.hip, conv_ws.hip).  What the product build relies on, instruction by instruction
// (gfx950, ROCm 7.2; MI355X_MICROARCH.md "Workgroup dispatch, XCD placement & inter-workgroup visibility"):
//   * the partial sums are stored with relaxed agent-scope atomic stores = `global_store_dword ... sc1`: write-through, the
//     line leaves this XCD's L2 towards memory instead of waiting dirty in it;
//   * `s_waitcnt vmcnt(0)` in inline asm (the compiler can neither drop nor move it) retires those stores before the wave
//     draws its ticket: relaxed agent-scope `global_atomic_add` with return, performed at the memory side;
//   * the workgroup that draws the last ticket of a sample reads every partial with relaxed agent-scope atomic loads =
//     `global_load ... sc1`: they bypass the CU's L1, and L2 cannot hold a stale copy of a line nobody on this XCD has read.
// This is the guide's "sc1 payload -> asm vmcnt(0) -> flag" form with the ticket as the flag; it is NOT the C++ memory model's
// release / acquire pair (a relaxed RMW orders nothing formally), which on gfx950 is `buffer_wbl2 sc1` (+15...50 % on every
// convolution: it writes back all dirty output lines of the XCD) and `buffer_inv sc1`.  -DGTTS_FENCED_FINALIZE=1 builds add
// exactly those two fences: the reference variant for A/B runs and for a compiler / chip on which the assumption breaks
// (tools/gpu_fenced_check.sh compares the two builds bit for bit).
#ifndef GTTS_FENCED_FINALIZE
#define GTTS_FENCED_FINALIZE 0
#endif

// Linear workgroup id -> position in an XCD-banded order: id % 8 is the XCD the hardware picks, id / 8 the order of
// arrival there.  Bijective for any n (the first n % 8 bands are one longer).  GTTS_XCD_BANDS=0 keeps the identity.
#ifndef GTTS_XCD_BANDS
#define GTTS_XCD_BANDS 1
#endif
__device__ __forceinline__ int xcd_slot(int id, int n) {
    if (!GTTS_XCD_BANDS) return id;
    const int q = n >> 3, r = n & 7, xcd = id & 7, k = id >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + k;
}

__device__ __forceinline__ float mish_f(float x) {
    // no implicit mul+add fusion in here: `n + 2` may or may not be contracted with the multiply that produced n (n has a
    // second use), and the choice can differ between template instances that must agree bit for bit (conv_mfma.hip)
#pragma clang fp contract(off)
    float e = __expf(fminf(x, 40.0f));
    float n = e * (e + 2.0f);
    return x * (n * __builtin_amdgcn_rcpf(n + 2.0f));
}

// v * m for a mask factor m that is 0 or 1 (or a mask value), with 0 * anything == 0 -- v_mul_legacy_f32.  The persistent convolution's
// 16-byte halo loads reach up to a frame in front of / three frames behind a tensor (conv_ws.hip: the neighbouring tensor's bytes, or
// workspace the caller never initialised); those positions carry m = 0, and an IEEE multiply would turn a NaN / Inf found there into a NaN
// of the result (found in round 6: after ONE overflowing call every later call on the same workspace returned NaN).  Finite operands:
// identical to v_mul_f32.
__device__ __forceinline__ float mul_mask0(float v, float m) {
    float r;
    asm("v_mul_legacy_f32 %0, %1, %2" : "=v"(r) : "v"(v), "v"(m));
    return r;
}

// The ResnetBlock identity tail of one element (Grad-TTS/model/diffusion.py:72,78): out = x m + Mish(GN(h)) m with GN(h) = h sc + sh.
// ONE definition for tail_identity_kernel (misc.hip) and the tail fused into the attention context pass (attn.hip): both forms of
// the same tensor must agree bit for bit, so nothing here is left to per-file contraction choices.
__device__ __forceinline__ float tail_value(float h, float x, float sc, float sh, float m) {
#pragma clang fp contract(off)
    return fmaf(mish_f(fmaf(h, sc, sh)), m, x * m);
}

// fp32 -> (hi, lo) bf16 pair with hi + lo == x to ~2^-17 relative (both round-to-nearest-even).
__device__ __forceinline__ void split_bf16(float x, __bf16 &h, __bf16 &l) {
    h = (__bf16)x;
    l = (__bf16)(x - (float)h);
}

// ---------------------------------------------------------------------------------------------------
// GTTS_PREC_F16F8: the "f16 + fp8 cross terms" split of an fp32 contraction (conv_mfma.hip, NSPLIT == 3).
//   x = xh + xl, xh = fp16(x) (RNE, 11 significant bits), xl = x - xh exactly (|xl| <= 2^-12 |x|);  w = wh + wl likewise.
//   w x  =  wh xh                      one v_mfma_f32_32x32x16_f16 per 16 channels (products of 22 bits: exact in the fp32 accumulate)
//         + w xl + wl x   (- wl xl)    ONE v_mfma_f32_32x32x64_f8f6f4 per 32 channels: K block 0 = q8(w) . q8(xl 2^S),
//                                      K block 1 = q8(wl 2^(S+D)) . q8(x 2^-D), q8 = fp8 e4m3 (4 significant bits).
// The cross terms are 2^-12 of the product, so their 2^-4 relative rounding leaves ~2^-17 per product -- the grade of the bf16x3
// split (measured on the chip, tools/probe/f8_probe2.hip: 1.3e-5 vs 4.2e-6 relative on a K = 1152 reduction) -- for 2/3 of its
// MFMA cycles (the fp8 instruction retires twice the K per cycle) and 1.53x its sustained rate under the power cap
// (tools/probe/f8_probe.hip).  Both fp8 products carry the factor 2^S; instead of the instruction's block scales the fp16 weights are
// stored pre-multiplied by 2^S (exact) so that the accumulator holds 2^S x the result and the epilogue multiplies by 2^-S (exact).
// Ranges: fp8 e4m3 tops out at 448 and the conversion instruction returns NaN beyond it (probed), so both operands saturate
// through v_med3_f32 first: |xl 2^S| <= |x| / 2 -- activations beyond 896 only lose their own cross term (fp16-grade for that
// element, finite); |x 2^-D| saturates beyond 7168.  Weights: |w| 2^S must stay below the fp16 maximum (|w| < 63).
constexpr int F8_S = 10;         // common factor 2^S of the two fp8 products and of the fp16 weights
constexpr int F8_D = 4;          // x enters K block 1 as x 2^-D, wl as wl 2^(S+D)
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(2))) short s16x2;
__device__ __forceinline__ float f8_sat(float v) { return __builtin_amdgcn_fmed3f(v, -448.0f, 448.0f); }
// two fp32 -> two fp8 e4m3 (RNE) in the low (hi_word = false) or high half of `old`
template <bool HI>
__device__ __forceinline__ int cvt2_fp8(float a, float b, int old) { return __builtin_amdgcn_cvt_pk_fp8_f32(a, b, old, HI); }
// the same of a / scale, b / scale in one instruction (v_cvt_scalef32_pk_fp8_f32: probed, tools/probe/f8_probe.hip -- it DIVIDES by
// the scale, rounds to nearest even like the plain conversion, and returns NaN beyond +-448 x scale as well)
template <bool HI>
__device__ __forceinline__ int cvt2_fp8_div(float a, float b, float scale, int old) {
    return __builtin_bit_cast(int, __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(__builtin_bit_cast(s16x2, old), a, b, scale, HI));
}
// the two cross-term operands of the f16 + fp8 split for a pair of values (x0, x1) with fp16 hi parts (h0, h1):
// lo: q8((x - h) 2^S) saturated, xq: q8(x 2^-D) saturated, packed into the low or high half of a word
template <bool HI>
__device__ __forceinline__ void f8_cross_pair(float x0, float x1, _Float16 h0, _Float16 h1, int &lo, int &xq) {
    constexpr float LS = 1.0f / (float)(1 << F8_S), LB = 448.0f * LS, XS = (float)(1 << F8_D), XB = 448.0f * XS;
    const float t0 = __builtin_amdgcn_fmed3f(x0 - (float)h0, -LB, LB), t1 = __builtin_amdgcn_fmed3f(x1 - (float)h1, -LB, LB);
    const float u0 = __builtin_amdgcn_fmed3f(x0, -XB, XB), u1 = __builtin_amdgcn_fmed3f(x1, -XB, XB);
    lo = cvt2_fp8_div<HI>(t0, t1, LS, lo);
    xq = cvt2_fp8_div<HI>(u0, u1, XS, xq);
}

// Activation range record of the f16 + fp8 split (GTTS_PREC_F16F8).  The staging code keeps the running maximum of |x| of everything
// it splits (one v_max3_f32 per value pair); a lane that saw |x| >= 1024 -- the smallest magnitude whose fp16 residual can exceed the
// fp8 cross-term range, i.e. from where on an element is carried at fp16 grade only -- adds one event and the maximum to the caller's
// record when its kernel ends: {unsigned events, bit pattern of max |x|} at the start of the workspace (gtts_workspace_status).
// NaN operands do not raise the running maximum (v_max_f32 returns the other operand); they are caught where they surface: a
// Block convolution whose GroupNorm statistics come out non-finite adds an event with max |x| = inf from its fused finalize.
constexpr float F8_ACT_LIMIT = 1024.0f;
__device__ __forceinline__ float f8_range_track(float m, float a, float b) { return fmaxf(fmaxf(m, fabsf(a)), fabsf(b)); }
__device__ __forceinline__ void f8_range_note(unsigned *rec, float m) {
    if (rec != nullptr && !(m < F8_ACT_LIMIT)) {       // (NaN counts)
        atomicAdd(rec, 1u);
        atomicMax(rec + 1, __builtin_bit_cast(unsigned, m) & 0x7fffffffu);
    }
}

// activation load / store through a buffer descriptor (per-lane byte offset + scalar byte offset), fp32 or bf16 storage
template <typename AT>
__device__ __forceinline__ float ld_act(__amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    if constexpr (sizeof(AT) == 4) return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff, soff, 0));
    else return __builtin_bit_cast(float, (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rs, voff, soff, 0) << 16);
}
// GTTS_OUT_NT: cache policy of the kernels' activation OUTPUT stores -- 2 = nt (streaming: the line is the first to leave the XCD's 4 MB
// L2, which then keeps the consumer's weights / halo rows instead of output lines no CU of this launch reads again), 0 = plain.
// Measured (round 6, same box, ms per U-Net call): nt on the persistent convolution's epilogue only (conv_ws.hip, GTTS_WS_EPI_AUX = 2)
// 6.228-6.234; nt on every kernel's output stores as well (this switch) 6.254-6.256 -> stays 0.
#ifndef GTTS_OUT_NT
#define GTTS_OUT_NT 0
#endif
template <typename AT>
__device__ __forceinline__ void st_act(float v, __amdgpu_buffer_rsrc_t rs, int voff, int soff) {
    if constexpr (sizeof(AT) == 4) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rs, voff, soff, GTTS_OUT_NT);
    else __builtin_amdgcn_raw_buffer_store_b16(__builtin_bit_cast(short, (__bf16)v), rs, voff, soff, GTTS_OUT_NT);
}

// Wave-wide reductions on the DPP path (full-rate VALU, no LDS crossbar): __shfl_xor lowers to ds_bpermute_b32 plus an
// lgkmcnt wait per step -- 12 dependent LDS round trips per 64-lane reduction, measured at ~11k cycles for the 16
// reductions of the conv epilogue.  Butterfly inside a 16-lane row (quad_perm, row_half_mirror, row_mirror), then the
// four row results are combined in a fixed order: deterministic, every lane gets the total.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_f(float v, int l) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l));
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_f<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_f<0x141>(v);     // row_half_mirror
    v += dpp_f<0x140>(v);     // row_mirror
    return (lane_f(v, 0) + lane_f(v, 16)) + (lane_f(v, 32) + lane_f(v, 48));
}
// V (8 or 16) independent wave-wide sums at once, "transposing": every step halves the number of live values instead of
// reducing each of them across all 64 lanes -- v_permlane32_swap pairs lanes (l, l + 32) for value pairs (k, k + V/2),
// v_permlane16_swap pairs (l, l + 16), four DPP steps finish inside the 16-lane rows: V/2 + V/4 swaps and V/2 + V/4 + V adds
// instead of 11 V instructions.  On return out[k] (k < V/4) holds, in EVERY lane of row r = lane >> 4, the total of value
// k + (V/4)(r & 1) + (V/2)(r >> 1).  Fixed association: deterministic, and the same for V = 8 and V = 16.
template <int V>
__device__ __forceinline__ void wave_sums_transposed(const float (&v)[V], float (&out)[V / 4]) {
    static_assert(V == 8 || V == 16, "8 or 16 values");
    float r1[V / 2];
#pragma unroll
    for (int k = 0; k < V / 2; ++k) {
        const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v[k]), __builtin_bit_cast(unsigned, v[k + V / 2]), false, false);
        const unsigned x0 = sw[0], x1 = sw[1];          // (copy the elements first: bit_cast of a vector element reads element 0)
        r1[k] = __builtin_bit_cast(float, x0) + __builtin_bit_cast(float, x1);
    }
#pragma unroll
    for (int k = 0; k < V / 4; ++k) {
        const auto sw = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, r1[k]), __builtin_bit_cast(unsigned, r1[k + V / 4]), false, false);
        const unsigned x0 = sw[0], x1 = sw[1];
        float t = __builtin_bit_cast(float, x0) + __builtin_bit_cast(float, x1);
        t += dpp_f<0xB1>(t);      // quad_perm [1,0,3,2]
        t += dpp_f<0x4E>(t);      // quad_perm [2,3,0,1]
        t += dpp_f<0x141>(t);     // row_half_mirror
        t += dpp_f<0x140>(t);     // row_mirror
        out[k] = t;
    }
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));
    v = fmaxf(v, dpp_f<0x4E>(v));
    v = fmaxf(v, dpp_f<0x141>(v));
    v = fmaxf(v, dpp_f<0x140>(v));
    return fmaxf(fmaxf(lane_f(v, 0), lane_f(v, 16)), fmaxf(lane_f(v, 32), lane_f(v, 48)));
}

// ---------------------------------------------------------------------------------------------------
// MFMA convolution launch descriptor (conv_mfma.hip).  One kernel family covers the 3x3 Block convs, the
// stride-2 Downsample, the 4x4/stride-2 ConvTranspose (as four 2x2 phase convs), and every 1x1 conv.
enum ConvMode { CONV_C3 = 0, CONV_DN = 1, CONV_UP = 2, CONV_P1 = 3, CONV_C7 = 4 };   // C7: Conv2d 7x7 pad 3 (DiffVC PostNet)
enum ConvPro {
    PRO_PLAIN = 0,   // v = x
    PRO_MASK = 1,    // v = x * mask                                   (Block input,    diffusion.py:57)
    PRO_GN = 2,      // v = (Mish(GN(x)) * mask + tbias) * mask        (block1 -> block2, diffusion.py:58,76,57)
    PRO_IGLU = 3     // v = (IN(x[c]) * sigmoid(IN(x[c + cin])) + tbias) * mask   (DiffVC RefBlock: InstanceNorm + GLU,
                     //      DiffVC/model/modules.py:140-157,160-165); the source tensor has 2*cin channels
};
enum ConvEpi {
    EPI_PLAIN = 0,   // out = acc + bias
    EPI_STATS = 1,   // out = acc + bias, plus per-(sample,group) partial sums for GroupNorm
    EPI_TAIL = 2,    // out = acc + bias + Mish(GN(h_raw)) * mask      (ResnetBlock tail with res_conv, :78)
    EPI_ATTN = 3     // out = acc + bias_b + x                          (Residual(Rezero(LinearAttention)))
};

struct ConvArgs {
    // input: channels [0,c0) come from src0, [c0,c0+c1) from src1 (torch.cat on dim 1 without the copy)
    const void *src0;       // activation tensors are fp32 or (act_bf16) bf16
    const void *src1;
    int c0, c1;
    int cin;            // c0 + c1
    int nchunk;         // ceil(cin / (16 * kch))   (set by launch_conv)
    int B, Hin, Win, Hout, Wout;
    const float *mask;  // [B][T] (level-0 mask); level-l column j is mask[b][j << lvl]
    int T;
    int lvl_in, lvl_out;
    int pro;
    const float *sc, *sh;   // PRO_GN: per-(b, input channel) GroupNorm scale / shift   [B][cin]
    const float *tb;        // PRO_GN: per-(b, input channel) time bias, row stride tb_stride (pre-offset)
    int tb_stride;
    // weights: packed bf16 blocks (see pack.hip); per-sample stride in bytes (0 = shared)
    const unsigned char *w;
    size_t w_bstride;
    const float *bias;      // [cout] (+ b * bias_bstride floats)
    size_t bias_bstride;
    int cout;
    int epi;
    void *out;              // [B][cout][Hout][Wout]
    float *partials;        // EPI_STATS: [B][nparts][groups][2]
    int nparts;
    int groups;
    const void *eh;         // EPI_TAIL: h_raw [B][cout][Hout][Wout]
    const float *esc, *esh; // EPI_TAIL: [B][cout]
    const void *eres;       // EPI_ATTN: residual [B][cout][Hout][Wout]
    int nsplit;             // 2: bf16x3 (hi/lo), 1: plain bf16
    int use_ws;             // plan option gtts_unet_cfg.conv_ws: eligible Block convolutions take conv_ws.hip
    const float *omask;     // EPI_PLAIN, conv_mfma.hip only: [B][Wout] column mask multiplied into the output (the data gradient
                            // of a masked convolution, train.hip); nullptr: none
    int act_bf16;           // 1: activation tensors are stored as bf16 (GTTS_PREC_BF16_STORE)
    int f16f8;              // 1: GTTS_PREC_F16F8 -- eligible Block convolutions (conv_f16f8_ok) run the f16 + fp8 split on weights
                            //    packed in that format; everything else of the plan stays bf16x3 (nsplit == 2)
    // EPI_STATS with the GroupNorm finalize fused in: the last workgroup of a sample to publish its partial sums (an
    // agent-scope ticket per sample) reduces them in a fixed order and writes the per-channel scale / shift.
    unsigned *ticket;       // [B] zero before the launch; reset to zero by the finalizing workgroup.  nullptr: not fused
    const float *gn_gamma, *gn_beta;   // [cout]
    float *gn_sc, *gn_sh;   // [B][cout]
    float gn_count;         // elements per group = (cout / groups) * Hout * Wout
    int tiles_x, tiles_y;
    int stat_rows;          // EPI_STATS: partial slots per (row pair, column block) instead of per tile (set by launch_cfg)
    unsigned *sat;          // f16 + fp8 staging: activation range record {events, bit pattern of max |x|} (f8_range_note); nullptr: none
};

// geometry of one conv configuration (compile-time in the kernel, mirrored on the host)
struct ConvGeom {
    int MT;     // output channels per workgroup
    int TR;     // output rows per workgroup (32 columns always)
    int nst;    // weight stages per chunk
    int tps;    // taps per stage
    int kch;    // 16-channel MFMA k-steps per chunk (chunk = 16 * kch input channels)
};
// host helper: how a layer is tiled (must match the template instantiations in conv_mfma.hip)
static inline ConvGeom conv_geom(int mode, int cin, int cout, int f16f8 = 0) {
    ConvGeom g;
    bool wide = cout > 64 && mode != CONV_C7;      // the 7x7 stage (7 taps) only fits LDS with the 64-cout tile
    g.MT = wide ? 128 : 64;
    g.kch = f16f8 ? 2 : 1;   // 16-channel chunks: measured faster than 32 (occupancy: 3 workgroups per CU beat fewer barriers);
                             // the f16 + fp8 split walks 32 (K of the fp8 instruction is 64 = 32 channels x 2 cross terms)
    (void)cin;
    if (mode == CONV_C7) { g.TR = 8; g.nst = 7; g.tps = 7; }
    else if (mode == CONV_DN) { g.TR = 4; g.nst = 3; g.tps = 3; }
    else if (mode == CONV_UP) { g.TR = wide ? 4 : 8; g.nst = 2; g.tps = 2; }
    else if (mode == CONV_P1) { g.TR = wide ? 4 : 8; g.nst = 1; g.tps = 1; }
    else { g.TR = wide ? 4 : 8; g.nst = 3; g.tps = 3; }
    return g;
}
static inline size_t conv_wblock_bytes(const ConvGeom &g) { return (size_t)g.tps * g.MT * 64 * g.kch; }
// packed bytes of one conv's weights: [phase][chunk][stage][cout tile] blocks
static inline size_t conv_packed_bytes(int mode, int cin, int cout) {
    ConvGeom g = conv_geom(mode, cin, cout);
    size_t nchunk = (cin + 16 * g.kch - 1) / (16 * g.kch), ncot = (cout + g.MT - 1) / g.MT;
    size_t phases = (mode == CONV_UP) ? 4 : 1;
    return phases * nchunk * g.nst * ncot * conv_wblock_bytes(g);
}

hipError_t launch_conv(int mode, const ConvArgs &a, hipStream_t st);
// GTTS_W64_RING=1 builds (A/B only): the 64-channel f16 + fp8 tile of conv_ws.hip takes its weights through per-wave LDS rings filled by
// LDS-DMA, in column stages (round 6; conv_ws.hip, "W64").  Built, parity-green, measured on one box against the register-load kernel:
// 275 / 261 us against 266 / 259 per level-0 launch -- not adopted (profiles/NEGATIVE_RESULTS.md).  The switch also selects the weight
// packing of those layers (pack.hip: column stages) and keeps them off the small-launch form.
#ifndef GTTS_W64_RING
#define GTTS_W64_RING 0
#endif
// conv_ws.hip: the persistent wave-specialised Block convolution.  GTTS_WS=0 builds (A/B only) keep every layer on conv_mfma.hip.
#ifndef GTTS_WS
#define GTTS_WS 1
#endif
bool conv_ws_eligible(int mode, int c0, int c1, int cout, int pro, int epi, int nsplit, int f16f8 = 0);
int conv_ws_nparts(int cout, int Hout, int Wout);      // GroupNorm partial slots per sample it writes (one per 32-frame x 5-row block)
bool conv_ws_small(int cout, int groups, int Hout, int Wout, int B, int f16f8 = 0);   // the launch takes the three-wave workgroup form (same arithmetic)
hipError_t launch_conv_ws(const ConvArgs &a, hipStream_t st);
// conv_up.hip: Upsample with the four output phases computed from one staged tile (fp32 storage, bf16x3)
bool conv_up4_eligible(const ConvArgs &a);
// GTTS_PREC_F16F8 plans: Upsample in the f16 + fp8 split (conv_up.hip); decides the packing of the layer's weights as well.  0: bf16x3 (A/B)
#ifndef GTTS_UP_F16F8
#define GTTS_UP_F16F8 1
#endif
bool conv_up4_f16f8_ok(int cin, int cout);
const char *conv_up4_f8_name();      // the instance launch_conv_up4 launches for such layers (per-op tables)
// conv_up_ws.hip: the wave-specialised form of the same (producer waves load, stage and STORE).  Built, parity-green, measured on one box
// against conv_up.hip's uniform-wave kernel: 114.5 / 132.3 us against 110 / 142.6 (128- / 64-channel layer) -- level; OFF (1: A/B builds)
#ifndef GTTS_UP_WS
#define GTTS_UP_WS 0
#endif
bool conv_up4_ws_ok(int cin, int cout);
const char *conv_up4_ws_name();
hipError_t launch_conv_up4_ws(const ConvArgs &a, hipStream_t st);
hipError_t launch_conv_up4(const ConvArgs &a, hipStream_t st);
// Block convolutions that take the f16 + fp8 split when the plan's precision is GTTS_PREC_F16F8 (conv_mfma.hip): 3x3, whole
// 32-channel chunks (a concatenated input splitting on one), mask / GroupNorm prologue, statistics epilogue, and an LDS
// footprint that leaves two workgroups per CU.  Decides the packing of the layer's weights as well (pack.hip).
// use_ws: the plan runs eligible layers on the persistent kernel (gtts_unet_cfg.conv_ws) -- 64-channel layers take the split only there.
bool conv_f16f8_ok(int mode, int c0, int c1, int cout, int pro, int epi, int use_ws);
bool conv_ws_f8_fits(int cin, int pro, int mt, int cout);
bool conv_small_tiles(int mode, int cout, int Hout, int Wout, int B);   // half-height tiles for launches smaller than the chip
bool conv_rowpair_stats(int mode, int cout, int Hout, int Wout);        // GroupNorm partial slots per row pair (batch-size independent)

}  // namespace gtts
