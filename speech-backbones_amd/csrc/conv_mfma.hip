// conv_mfma.hip -- implicit-GEMM convolutions on the bf16 MFMA pipe for gfx950 (MI355X).
//
// One kernel family covers every dense contraction of the score U-Net (Grad-TTS/model/diffusion.py):
//   CONV_C3  Conv2d 3x3 pad 1            Block            diffusion.py:52-58
//   CONV_DN  Conv2d 3x3 stride 2 pad 1   Downsample       diffusion.py:30-36
//   CONV_UP  ConvTranspose2d 4x4 s2 p1   Upsample         diffusion.py:21-27   (four 2x2 phase convolutions)
//   CONV_P1  Conv2d 1x1                  res_conv / folded LinearAttention output   diffusion.py:70,78,97-100
//
// Data layout: activations stay in the reference's NCHW fp32 layout ([B,C,mel-bin,frame], frame fastest), so
// global loads are coalesced along the mel-frame axis.  GEMM view per workgroup:
//     D[cout (MT)][pixel (TR x 32)] = sum_{tap, cin} W[cout][cin, tap] * X[cin][pixel + tap]
// with cin walked in chunks of 16*KCH (K of v_mfma_f32_32x32x16_bf16 is 16).  Per chunk the workgroup stages the
// halo tile of the chunk's input channels into LDS *through registers*, applying the producer's epilogue on the
// way (GroupNorm affine + Mish + mask + time bias: "apply-on-load", so normalised tensors never touch HBM) and
// splitting fp32 into bf16 hi/lo.  LDS image: [kgroup(2*KCH)][pixel][8 channels] x {hi, lo}: one 16-byte slot per
// (pixel, 8-channel group) -> both the staging ds_write_b128 and the MFMA B-fragment ds_read_b128 are
// conflict-free, and a tap is just a pixel offset.  Weights are pre-packed on the device (pack.hip) in
// exactly the LDS image order, one contiguous block per (chunk, stage, cout tile).
//
// Everything that selects code (mode, tiling, prologue, epilogue, precision) is a template parameter: the staging
// loop is straight-line code (unconditional loads from clamped addresses + selects, no exec-mask branches).
//
// Precision: NSPLIT == 2 computes hi*hi + hi*lo + lo*hi with fp32 accumulation (error ~2^-17 per product,
// i.e. fp32-grade: SURVEY.md section 0); NSPLIT == 1 is plain bf16; NSPLIT == 3 is the f16 + fp8 split of GTTS_PREC_F16F8
// (common.h: fp16 hi*hi on v_mfma_f32_32x32x16_f16 + both cross terms in one v_mfma_f32_32x32x64_f8f6f4 per 32 channels; 3x3 Block
// convolutions on whole 32-channel chunks only, KCH == 2).
//
// Wave tile: (MF x 32) output channels x (2 rows x 32 columns) pixels; a workgroup is WM x WN waves.
#include "common.h"
#include <algorithm>
#include <atomic>

// minimum waves per SIMD requested from the register allocator (= workgroups per CU with 4-wave workgroups)
#ifndef GTTS_C3_WAVES
#define GTTS_C3_WAVES 3
#endif
// single-pass bf16 kernels (config 3) have half the fragment / weight-staging registers and half the LDS
#ifndef GTTS_C3_WAVES_BF16
#define GTTS_C3_WAVES_BF16 3
#endif
#ifndef GTTS_DN_WAVES
#define GTTS_DN_WAVES 2
#endif
// the f16 + fp8 split (NSPLIT == 3, GTTS_PREC_F16F8) stages 32-channel chunks: 74-80 KB of LDS per workgroup, two per CU
#ifndef GTTS_F8_WAVES
#define GTTS_F8_WAVES 2
#endif
// 0 (A/B builds): 64-channel layers stay bf16x3 in GTTS_PREC_F16F8 plans
#ifndef GTTS_F8_WS64
#define GTTS_F8_WS64 1
#endif
#define GTTS_WAVES(MODE) ((MODE) == CONV_DN ? GTTS_DN_WAVES : GTTS_C3_WAVES)
#define GTTS_WAVES_NS(MODE, NSPLIT) ((MODE) == CONV_DN ? GTTS_DN_WAVES : ((NSPLIT) == 3 ? GTTS_F8_WAVES : ((NSPLIT) == 1 ? GTTS_C3_WAVES_BF16 : GTTS_C3_WAVES)))
// Diagnostics exist only in -DGTTS_DIAG builds (tools/abexp.sh, tools/trace_conv.py); the product library is compiled
// without it and the three switches below are then forced off, whatever else is on the command line.
// GTTS_EXP: timing-only ablations of the main loop (results are WRONG)
//   1 no MFMAs   2 no fragment ds_reads   3 no activation transform/ds_write   4 no weight staging   5 no barriers
#ifndef GTTS_DIAG
#undef GTTS_EXP
#undef GTTS_TRACE
#undef GTTS_WDMA
#undef GTTS_ADBUF
#undef GTTS_LDS_MIN
#endif
// GTTS_LDS_MIN (diagnostic builds): minimum dynamic LDS bytes per workgroup, to force 1 (> 80 KB) or 2 (> 54 KB) workgroups
// per CU when tracing the un-contended phase times
#ifndef GTTS_LDS_MIN
#define GTTS_LDS_MIN 0
#endif
// GTTS_C3_LDS_MIN (any build): minimum dynamic LDS bytes of the bf16x3 Block convolutions (3x3, GroupNorm statistics
// epilogue) only -- caps how many of THEM a CU holds (> 54 KB: two) while leaving its registers and wave slots to the
// bandwidth-bound kernels of the other sub-batch streams.  0 = off.
#ifndef GTTS_C3_LDS_MIN
#define GTTS_C3_LDS_MIN 0
#endif
#ifndef GTTS_EXP
#define GTTS_EXP 0
#endif
// 0 = the round-1 conditional prefetches (kept for A/B builds only)
#ifndef GTTS_UNCOND_PF
#define GTTS_UNCOND_PF 1
#endif
// GTTS_PRIV=1 (measured, not adopted; kept buildable): private weight slices for the bf16x3 3x3 convolutions -- see the
// kernel comment.  All parity tests pass; 4 of the 6 workgroup barriers per chunk and every wait in front of an MFMA are
// gone, and the time does not move: 128-cout GroupNorm kernel 216 vs 221 us, 128-cout mask kernel 205 vs 205, the 64-cout
// kernels 241 vs 231 and 211 vs 204 (they stage every weight twice), 7.40 vs 7.38 ms per U-Net call.  Together with the
// unconditional-prefetch result (exact vmcnt, no change) this rules out waits and barriers as what holds the loop at ~80 % of
// the matrix pipe; what is left is issue time of the non-MFMA instructions, which this variant does not reduce (it adds
// two fragment reads per tap).
#ifndef GTTS_PRIV
#define GTTS_PRIV 0
#endif
// launches with fewer workgroups than this use half-height 3x3 tiles (conv_small_tiles).  Measured: 256 also catches the
// 5-utterance sub-batches of the B = 16 sampler (200 workgroups) and costs 1 % there; 128 and 192 keep all of the B = 1 gain
// (1.73 -> 1.57 ms per U-Net call) at no cost for B = 16.
#ifndef GTTS_SMALL_WGS
#define GTTS_SMALL_WGS 128
#endif
#ifndef GTTS_PRIV_WAVES
#define GTTS_PRIV_WAVES 3
#endif
#define GTTS_SYNC() do { if (GTTS_EXP != 5) lds_barrier(); } while (0)     // LDS-only fence: prefetches stay in flight

// GTTS_TRACE=1 (diagnostic builds only): per-wave s_memtime phase sums of the 3x3 GroupNorm kernel, read back with
// gtts_debug_trace().  Phases: 0 top-of-chunk barrier, 1 activation transform + LDS write, 2 weight wait + LDS write,
// 3 barrier after the weight write, 4 prefetch issue + fragment reads + MFMAs, 5 inter-stage barrier, 6 whole loop.
// GTTS_WDMA=1: where LDS allows two weight-stage buffers at three workgroups per CU (the 64-cout tile of the 3x3
// conv), the packed weight stage goes global -> LDS directly (buffer_load_dwordx4 ... lds): no staging VGPRs, no
// ds_write_b128, and -- the stage being double-buffered -- one barrier per weight stage instead of two.  Verified
// correct on MI355X (all parity tests) and speed-neutral (214.6 vs 215.9 us): waits and barriers of one wave are
// covered by the other two waves of the SIMD; what adds to the MFMA time is VALU work, which this does not change.
// Off by default (it costs 12 KB of LDS); kept as the building block for DMA-staged activations.
#ifndef GTTS_WDMA
#define GTTS_WDMA 0
#endif
#ifndef GTTS_TRACE
#define GTTS_TRACE 0
#endif
// GTTS_ADBUF=1 (diagnostic builds; measured, not adopted): the 128-cout 3x3 kernel double-buffers its activation image
// (13 KB more LDS, still three workgroups per CU) and transforms / stages chunk c+1 inside the last weight stage of chunk c,
// branch-free so that the transform shares the MFMAs' basic block.  hipcc still emits the transform as one VALU block in
// front of the stage's MFMAs (forcing the interleave with sched_group_barrier spills at 168 VGPRs), and the variant
// measures 256 vs 238 us on the GroupNorm-prologue kernel and 221 vs 220 us on the mask-prologue kernel: the staging phase
// cannot be hidden inside a wave from HIP source at this register budget.
#ifndef GTTS_ADBUF
#define GTTS_ADBUF 0
#endif
#ifndef GTTS_TRACE_CIN
#define GTTS_TRACE_CIN 128      // traced layer: cin == cout == this
#endif
// (s_setprio by phase -- staging high or MFMA high -- was measured: no effect, +-0.5 %.)
#if GTTS_TRACE
__device__ unsigned long long g_conv_trace[64 * 4 * 8 + 64 * 4 * 2 + 64 * 4 * 4];   // + [wg][wave]{prologue, epilogue} + epilogue parts      // zero-initialised; rewritten by every traced launch
extern "C" int gtts_debug_trace(unsigned long long *dst, int n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_conv_trace), sizeof(unsigned long long) * (n < 3584 ? n : 3584));
}
#define TR_MARK(ph) do { if (tr_on) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); tr_sum[ph] += t_ - tr_last; tr_last = t_; } } while (0)
#else
#define TR_MARK(ph) do { } while (0)
#endif

// Several instances of this kernel must produce the same bits for the same layer (regular and half-height tiles are picked by
// launch size, and results must not depend on how utterances are batched).  hipcc's default -ffp-contract=fast fuses a multiply
// with a following add where it sees fit, and it can see fit differently in different instances (measured on an experimental
// third instance of the 3x3 kernel: five more v_pk_fma_f32 in the statistics epilogue, GroupNorm shifts off by one ulp): every
// fused multiply-add in this file is therefore written as fmaf / fma explicitly, and implicit contraction is off.
#pragma clang fp contract(off)

namespace gtts {

template <int MODE, int WM, int WN, int MF, int KCH, int NF = 2>
struct ConvCfg {
    static constexpr int MT = WM * MF * 32;
    static constexpr int TR = WN * NF;
    static constexpr int TC = 32;
    static constexpr int NST = MODE == CONV_P1 ? 1 : (MODE == CONV_UP ? 2 : (MODE == CONV_C7 ? 7 : 3));
    static constexpr int HALO = MODE == CONV_C7 ? 3 : 1;    // C3 / C7: zero padding on each side
    static constexpr int TPS = NST;
    static constexpr int HR = MODE == CONV_P1 ? TR : (MODE == CONV_DN ? 2 * TR + 1 : TR + 2 * HALO);
    static constexpr int HC = MODE == CONV_P1 ? TC : (MODE == CONV_DN ? 2 * TC + 1 : TC + 2 * HALO);
    static constexpr int NPIX = HR * HC;
    static constexpr int NKG = 2 * KCH;                     // 8-channel groups per chunk (chunk = 16*KCH channels)
    static constexpr int AITER = (NKG * NPIX + 255) / 256;  // (pixel, kgroup) staging items per thread
    static constexpr int WBLK16 = TPS * MT * 2 * NKG;       // 16-byte units per weight block
    static constexpr int WITER = WBLK16 / 256;
    static_assert(WBLK16 % 256 == 0, "weight block must be a whole number of 256 x 16-byte rows");
};

static inline int conv_npar(int pro) { return pro == PRO_GN ? 3 : (pro == PRO_IGLU ? 5 : 0); }
template <int MODE, int WM, int FULLC>
struct ConvAdbuf { static constexpr bool on = GTTS_ADBUF && !GTTS_TRACE && GTTS_EXP == 0 && !GTTS_WDMA && MODE == CONV_C3 && WM == 2 && FULLC; };
template <int MODE, int WM, int FULLC>
struct ConvWdma { static constexpr bool on = GTTS_WDMA && !GTTS_TRACE && GTTS_EXP == 0 && MODE == CONV_C3 && WM == 1 && FULLC; };

static inline size_t conv_smem_bytes(int npix, int nkg, int wblk16, int cin, int pro, int mt) {
    size_t cpad = (size_t)((cin + 8 * nkg - 1) / (8 * nkg)) * 8 * nkg;
    return (size_t)npix * nkg * 16 * 2 + (size_t)wblk16 * 16 + (size_t)conv_npar(pro) * cpad * 4 + 4 * 2 * 4 * 2 * 4 +
           (size_t)3 * mt * 4;
}

// FULLC = 1: cin is a multiple of 16 (and a concatenated input splits on a 16-channel boundary), so there is no
// channel padding and every chunk comes from one source.  Then all global loads are buffer loads -- per-lane byte
// offset fixed per staging item, per-chunk/channel offset in an SGPR -- and need no VALU address arithmetic, and the
// zero padding of the halo is produced by the mask factor alone (out-of-image items read offset 0 and get m = 0).
// AT = storage type of the activation tensors (float, or __bf16 for the bf16-storage mode of BASELINE config 3: every
// activation is read / written as bf16, the accumulators and the GroupNorm statistics stay fp32).
//
// Wave tile: (MF x 32) output channels x (NF rows x 32 columns).  PRIV = 1 ("private weight slices", 3x3 only): every
// wave owns MF x 32 output channels for the whole tile height (WM = 4, WN = 1 for the 128-channel tile), so the weights
// a wave multiplies with are read by no other wave.  Each wave then copies ITS rows of the packed stage into its own
// LDS region and nobody has to wait for anybody: the two workgroup barriers per weight stage disappear (2 barriers per
// chunk -- around the shared activation image -- instead of 6).  LDS executes the DS operations of one wave in issue
// order, which makes the weight path a register-free ring: right after the fragment reads of tap j are issued the
// wave overwrites slot j with tap j of the NEXT stage (prefetched one stage ahead into wregs) and re-issues the global
// load of the stage after that, so a weight fragment is in LDS a whole stage before its first reader and no wait on a
// ds_write or a global load sits in front of an MFMA.
template <int MODE, int WM, int WN, int MF, int KCH, int PRO, int EPI, int NSPLIT, int FULLC, typename AT, int NF = 2, int PRIV = 0>
__global__ __launch_bounds__(256, (PRO == PRO_IGLU || MODE == CONV_C7) ? 2 : (PRIV ? GTTS_PRIV_WAVES : GTTS_WAVES_NS(MODE, NSPLIT))) void conv_mfma_kernel(const ConvArgs a) {
    constexpr int AB = (int)sizeof(AT);      // bytes per stored activation
    using C = ConvCfg<MODE, WM, WN, MF, KCH, NF>;
    static_assert(!PRIV || (MODE == CONV_C3 && FULLC && KCH == 1 && MF == 1), "private weight slices: 3x3, whole chunks, one fragment row per wave");
    static_assert(NSPLIT != 3 || (MODE == CONV_C3 && FULLC && KCH == 2 && !PRIV && sizeof(AT) == 4 && (PRO == PRO_MASK || PRO == PRO_GN)),
                  "f16 + fp8 split: 3x3 Block convolutions on whole 32-channel chunks, fp32 storage");
    static_assert(NSPLIT != 3 || (!ConvWdma<MODE, WM, FULLC>::on && !ConvAdbuf<MODE, WM, FULLC>::on), "f16 + fp8 split: plain staging only");
    constexpr int MT = C::MT, TR = C::TR, TC = C::TC, NST = C::NST, TPS = C::TPS, NKG = C::NKG;
    constexpr int HC = C::HC, NPIX = C::NPIX, AITER = C::AITER, WBLK16 = C::WBLK16;
    // single-pass bf16 (NSPLIT == 1) multiplies with the hi halves only: the block's first half ([split][tap][kg][MT] order) is
    // all it stages -- half the weight loads and LDS writes of a chunk
    constexpr int NW16 = (NSPLIT == 1 && !ConvWdma<MODE, WM, FULLC>::on) ? WBLK16 / 2 : WBLK16;
    constexpr int WITER = (NW16 + 255) / 256;

#if GTTS_TRACE
    const unsigned long long tr_entry = __builtin_amdgcn_s_memtime();
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4 *s_ah = reinterpret_cast<u32x4 *>(smem);      // [NKG][NPIX]  hi
    u32x4 *s_al = s_ah + NKG * NPIX;                    // [NKG][NPIX]  lo
    constexpr bool WDMA = ConvWdma<MODE, WM, FULLC>::on;
    constexpr bool ADBUF = ConvAdbuf<MODE, WM, FULLC>::on;
    constexpr int AIMG = 2 * NKG * NPIX;                // one activation image (hi + lo) in 16-byte units
    u32x4 *s_w = s_al + NKG * NPIX + (ADBUF ? AIMG : 0);   // [split][tap][kg][MT]  (WDMA: two such buffers)
    const int cpad = a.nchunk * 8 * NKG;
    // PRO_GN: [3][cpad] scale, shift, time bias; PRO_IGLU: [5][cpad] scale_a, shift_a, time bias, scale_b, shift_b
    constexpr int WLDS16 = PRIV ? 4 * NSPLIT * TPS * NKG * MF * 32 : WBLK16;   // PRIV: four private regions
    float *s_par = reinterpret_cast<float *>(s_w + (WDMA ? 2 : 1) * WLDS16);
    constexpr int NPAR = PRO == PRO_GN ? 3 : (PRO == PRO_IGLU ? 5 : 0);
    float *s_red = s_par + NPAR * cpad;                       // [4 waves][MF][4 octets][2]
    float *s_epi = s_red + 4 * 2 * 4 * 2;                     // [3][MT]: bias, (EPI_TAIL) GN scale, shift

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kg_l = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    // Workgroup -> (sample, pixel tile, cout tile / phase).  The hardware deals consecutive workgroup ids round-robin
    // to the 8 XCDs (each with its own L2); xcd_slot() turns that into 8 contiguous bands, and inside a band the
    // workgroups that read the same input tile (cout tiles, transposed-conv phases) and then the row-neighbour tiles
    // (shared halo rows) are adjacent, so the re-reads hit the XCD's L2 instead of HBM.
    const int ny = gridDim.x / (a.tiles_x * a.tiles_y * a.B);       // cout tiles (x 4 phases)
    int wg = xcd_slot(blockIdx.x, gridDim.x);
    const int by = wg % ny; wg /= ny;
    const int tile = wg % (a.tiles_x * a.tiles_y);
    const int b = wg / (a.tiles_x * a.tiles_y);
    const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
    int cot = by, phase = 0;
    if (MODE == CONV_UP) { phase = cot & 3; cot >>= 2; }
    const int ncot = (a.cout + MT - 1) / MT;
    const int ph_y = phase >> 1, ph_x = phase & 1;
    const int y0 = ty * TR, x0 = tx * TC;
    const int iy0 = MODE == CONV_P1 ? y0 : (MODE == CONV_DN ? 2 * y0 - 1 : y0 - C::HALO);
    const int ix0 = MODE == CONV_P1 ? x0 : (MODE == CONV_DN ? 2 * x0 - 1 : x0 - C::HALO);
    const int HWin = a.Hin * a.Win;

    // ---- staging items: geometry is chunk-invariant.  Out-of-image items load from offset 0 (valid memory)
    // and are zeroed by a select.
    int it_goff[AITER];     // offset inside a channel plane (clamped to 0 when outside the image); FULLC: byte offset
                            // of (first channel of the item's 8-group, pixel) inside the chunk
    float it_m[AITER];      // mask value at the item's frame; < 0 encodes "outside the image / no item"
#pragma unroll
    for (int it = 0; it < AITER; ++it) {
        const int idx = tid + it * 256;
        const bool has = idx < NKG * NPIX;
        const int kg = idx / NPIX;
        const int p = idx - kg * NPIX;
        const int pr = p / HC, pc = p - pr * HC;
        const int gy = iy0 + pr, gx = ix0 + pc;
        const bool in = has && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
        it_goff[it] = in ? gy * a.Win + gx : 0;
        if (FULLC) it_goff[it] = (it_goff[it] + min(kg, NKG - 1) * 8 * HWin) * AB;
        float m = 1.f;
        if (PRO != PRO_PLAIN) m = a.mask[(size_t)b * a.T + ((size_t)(in ? gx : 0) << a.lvl_in)];
        it_m[it] = in ? m : -1.f;
    }

    float araw[AITER][8];
    float brawst[PRO == PRO_IGLU ? AITER : 1][8];      // PRO_IGLU: the gate half (channel c + cin)
    // buffer descriptors (FULLC): built per chunk from wave-uniform scalars (readfirstlane keeps them in SGPRs --
    // a descriptor the compiler believes divergent is loaded through a waterfall loop)
    const int srcC0 = PRO == PRO_IGLU ? 2 * a.cin : a.c0;
    const AT *sbase0 = reinterpret_cast<const AT *>(a.src0) + (size_t)b * srcC0 * HWin;
    const AT *sbase1 = a.c1 > 0 ? reinterpret_cast<const AT *>(a.src1) + (size_t)b * a.c1 * HWin : sbase0;
    auto uniform_rsrc = [](const void *p, int bytes) {
        const unsigned long long u = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo), 0,
                                                 __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
    };
    auto load_act = [&](int chunk) {
        if constexpr (FULLC) {
            const int cb = chunk * (8 * NKG);
            const bool first = cb < a.c0 || PRO == PRO_IGLU;
            const __amdgpu_buffer_rsrc_t rs0 =
                uniform_rsrc(first ? sbase0 : sbase1, (first ? srcC0 : a.c1) * HWin * AB);
            const int soff = (first ? cb : cb - a.c0) * HWin * AB;
#pragma unroll
            for (int it = 0; it < AITER; ++it) {
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int so = soff + i * HWin * AB;
                    araw[it][i] = ld_act<AT>(rs0, it_goff[it], so);
                    if (PRO == PRO_IGLU) brawst[it][i] = ld_act<AT>(rs0, it_goff[it], so + a.cin * HWin * AB);
                }
            }
            return;
        }
#pragma unroll
        for (int it = 0; it < AITER; ++it) {
            const int idx = tid + it * 256;
            const int kg = min(idx / NPIX, NKG - 1);
            const int cbase = chunk * (8 * NKG) + kg * 8;
            const int nval = min(max(a.cin - cbase, 0), 8);        // valid channels of this 8-group
            const int cb0 = nval > 0 ? cbase : 0;
            const AT *pl;
            if (PRO == PRO_IGLU) pl = reinterpret_cast<const AT *>(a.src0) + ((size_t)b * 2 * a.cin + cb0) * HWin;
            else pl = (cb0 < a.c0) ? reinterpret_cast<const AT *>(a.src0) + ((size_t)b * a.c0 + cb0) * HWin
                                   : reinterpret_cast<const AT *>(a.src1) + ((size_t)b * a.c1 + (cb0 - a.c0)) * HWin;
            pl += it_goff[it];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int ii = min(i, max(nval, 1) - 1);            // clamp: always a valid address
                araw[it][i] = (float)pl[(size_t)ii * HWin];          // zeroing of i >= nval happens at use
                if (PRO == PRO_IGLU) brawst[it][i] = (float)pl[(size_t)(ii + a.cin) * HWin];
            }
        }
    };

    const unsigned char *wbase = a.w + (size_t)b * a.w_bstride;
    constexpr int PSEG = NSPLIT * TPS * NKG;              // PRIV: (split, tap, kgroup) segments of MF*32 rows per wave
    constexpr int PITER = PRIV ? PSEG * MF * 32 / 64 : 1; // 16-byte items per lane and stage (NKG = 2: item i = (split, tap) i)
    u32x4 wregs[PRIV ? PITER : WITER];
    const int wtotal = (MODE == CONV_UP ? 4 : 1) * a.nchunk * NST * ncot * WBLK16 * 16;       // bytes of this conv's blocks
    const __amdgpu_buffer_rsrc_t rsw = uniform_rsrc(wbase, wtotal);
    auto load_w = [&](int chunk, int stage) {
        const int blk = ((phase * a.nchunk + chunk) * NST + stage) * ncot + cot;
        if constexpr (WDMA) {
            // straight into LDS buffer (global stage index & 1): lane l of a wave lands at the wave's base + 16 l
            u32x4 *dst = s_w + ((chunk * NST + stage) & 1) * WBLK16 + wave * 64;
#pragma unroll
            for (int i = 0; i < WITER; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void *)(dst + i * 256), 16,
                                                         (tid + i * 256) * 16, blk * (WBLK16 * 16), 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < WITER; ++i)
                wregs[i] = __builtin_amdgcn_raw_buffer_load_b128(rsw, min(tid + i * 256, NW16 - 1) * 16, blk * (WBLK16 * 16), 0);
        }
    };

    // PRIV: tap j of global stage g (= chunk * NST + stage, clamped to the last one) -> wregs[j] (hi), wregs[TPS + j] (lo)
    u32x4 *s_wp = s_w + wave * (PSEG * MF * 32);          // this wave's private region: [split][tap][kg][MF*32 rows]
    auto load_w_tap = [&](int g, int j) {
        const int gl = min(g, a.nchunk * NST - 1);
        const int blk = gl * ncot + cot;                   // phase == 0 for CONV_C3
#pragma unroll
        for (int sp = 0; sp < NSPLIT; ++sp) {
            const int i = sp * TPS + j;
            // per-lane offset shared by all items; the (split, tap) item offset rides in the scalar offset
            wregs[i] = __builtin_amdgcn_raw_buffer_load_b128(rsw, (kg_l * MT + wm * MF * 32 + l31) * 16,
                                                             blk * (WBLK16 * 16) + i * (NKG * MT * 16), 0);
        }
    };

    f32x16 acc[MF][NF];
#pragma unroll
    for (int mi = 0; mi < MF; ++mi)
#pragma unroll
        for (int ni = 0; ni < NF; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    // (ragged-channel first layers keep the conditional form: the extra live range spills there)
    constexpr bool UNCOND_PF = GTTS_UNCOND_PF && FULLC;
    // activations first, weights second: the same queue order as on the loop's back edge, so the waits at the loop head
    // are exact on both paths
    if constexpr (PRIV) {
        load_act(0);
#pragma unroll
        for (int j = 0; j < TPS; ++j) load_w_tap(0, j);
    } else if (UNCOND_PF) { load_act(0); load_w(0, 0); }
    else { load_w(0, 0); load_act(0); }

    // (filled after the first tile's loads are in flight: the parameter loads share one memory round trip with them
    // instead of adding two serialized ones in front)
    // ---- per-(sample, channel) prologue parameters -> LDS (visible after the first barrier)
    if (PRO == PRO_GN || PRO == PRO_IGLU) {
        const int cs = PRO == PRO_IGLU ? 2 * a.cin : a.cin;      // channels of the raw source tensor
        for (int i = tid; i < cpad; i += 256) {
            const bool ok = i < a.cin;
            const int ic = ok ? i : 0;
            const float v0 = a.sc[(size_t)b * cs + ic], v1 = a.sh[(size_t)b * cs + ic];
            const float v2 = a.tb ? a.tb[(size_t)b * a.tb_stride + ic] : 0.f;
            s_par[i] = ok ? v0 : 0.f;
            s_par[cpad + i] = ok ? v1 : 0.f;
            s_par[2 * cpad + i] = ok ? v2 : 0.f;
            if (PRO == PRO_IGLU) {
                const float v3 = a.sc[(size_t)b * cs + a.cin + ic], v4 = a.sh[(size_t)b * cs + a.cin + ic];
                s_par[3 * cpad + i] = ok ? v3 : 0.f;
                s_par[4 * cpad + i] = ok ? v4 : 0.f;
            }
        }
    }

    // ---- per-output-channel epilogue parameters -> LDS (read after many barriers)
    for (int i = tid; i < MT; i += 256) {
        const int co = cot * MT + i;       // host guarantees cout % MT == 0
        s_epi[i] = a.bias[(size_t)b * a.bias_bstride + co];
        if (EPI == EPI_TAIL) {
            s_epi[MT + i] = a.esc[(size_t)b * a.cout + co];
            s_epi[2 * MT + i] = a.esh[(size_t)b * a.cout + co];
        }
    }


    const int m0 = wm * MF * 32;
#if GTTS_TRACE
    const bool tr_on = MODE == CONV_C3 && PRO == PRO_GN && a.cin == GTTS_TRACE_CIN && a.cout == GTTS_TRACE_CIN && (blockIdx.x % 97) == 5 && blockIdx.x / 97 < 64;
    unsigned long long tr_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tr_last = __builtin_amdgcn_s_memtime();
    const unsigned long long tr_t0 = tr_last;
#endif

    [[maybe_unused]] float f8_vmax = 0.f;      // f16 + fp8 split: running max |x| of everything this lane split (range record, common.h)
    // ---- transform + split + stage the activation tile of a chunk (straight-line code) into image (dh, dl)
    auto stage_act = [&](int chunk, u32x4 *dh, u32x4 *dl) {
#pragma unroll
        for (int it = 0; it < (GTTS_EXP == 3 ? 0 : AITER); ++it) {
            const int idx = tid + it * 256;
            const int kg = min(idx / NPIX, NKG - 1);
            const int p = idx - (idx / NPIX) * NPIX;
            const int pr = p / HC, pc = p - pr * HC;
            int lc = pc;
            if (MODE == CONV_DN) lc = (pc & 1) ? 33 + (pc >> 1) : (pc >> 1);   // column-parity planes
            const bool has = idx < NKG * NPIX;
            const float mraw = it_m[it];
            const bool inb = mraw >= 0.f;
            const float m = inb ? mraw : 0.f;
            const int cb = chunk * (8 * NKG) + kg * 8;
            const int nval = min(max(a.cin - cb, 0), 8);
            float sc[8], sh[8], tb[8], scb[8], shb[8];
            if (PRO == PRO_GN || PRO == PRO_IGLU) {
                const float4 *q = reinterpret_cast<const float4 *>(s_par + cb);
                const float4 *q1 = reinterpret_cast<const float4 *>(s_par + cpad + cb);
                const float4 *q2 = reinterpret_cast<const float4 *>(s_par + 2 * cpad + cb);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float4 u = q[h], u1 = q1[h], u2 = q2[h];
                    sc[4 * h + 0] = u.x; sc[4 * h + 1] = u.y; sc[4 * h + 2] = u.z; sc[4 * h + 3] = u.w;
                    sh[4 * h + 0] = u1.x; sh[4 * h + 1] = u1.y; sh[4 * h + 2] = u1.z; sh[4 * h + 3] = u1.w;
                    tb[4 * h + 0] = u2.x; tb[4 * h + 1] = u2.y; tb[4 * h + 2] = u2.z; tb[4 * h + 3] = u2.w;
                }
                if (PRO == PRO_IGLU) {
                    const float4 *q3 = reinterpret_cast<const float4 *>(s_par + 3 * cpad + cb);
                    const float4 *q4 = reinterpret_cast<const float4 *>(s_par + 4 * cpad + cb);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float4 u3 = q3[h], u4 = q4[h];
                        scb[4 * h + 0] = u3.x; scb[4 * h + 1] = u3.y; scb[4 * h + 2] = u3.z; scb[4 * h + 3] = u3.w;
                        shb[4 * h + 0] = u4.x; shb[4 * h + 1] = u4.y; shb[4 * h + 2] = u4.z; shb[4 * h + 3] = u4.w;
                    }
                }
            }
            bf16x8 vh, vl;
            [[maybe_unused]] float vv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v = araw[it][i];
                if (!FULLC) v = (inb && i < nval) ? v : 0.f;     // FULLC: zeroing comes from m == 0 (see kernel comment)
                if (PRO == PRO_MASK) {
                    v *= m;
                } else if (PRO == PRO_GN) {
                    const float y = fmaf(v, sc[i], sh[i]);
                    v = fmaf(mish_f(y), m, tb[i]) * m;
                } else if (PRO == PRO_IGLU) {
                    const float ga = fmaf(v, sc[i], sh[i]);
                    const float gb = fmaf(brawst[it][i], scb[i], shb[i]);
                    const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-gb));     // sigmoid
                    v = fmaf(ga, sg, tb[i]) * m;
                    if (!FULLC) v = (inb && i < nval) ? v : 0.f;
                }
                if constexpr (NSPLIT == 3) {
                    vv[i] = v;
                } else if constexpr (NSPLIT > 1) {
                    __bf16 h, l;
                    split_bf16(v, h, l);
                    vh[i] = h;
                    vl[i] = l;
                } else {
                    vh[i] = (__bf16)v;           // single-pass bf16: no lo plane (neither computed nor written)
                }
            }
            if constexpr (NSPLIT == 3) {
                // f16 + fp8 split (common.h): hi plane fp16 [kg][pixel][8 ch]; cross-term plane [g][pixel][16 ch] fp8 with
                // g = plane * 2 + (16-channel half): plane 0 = q8(xl 2^S), plane 1 = q8(x 2^-D); this item owns 8 of the 16 bytes
                f16x8 fh;
                int lw[2] = {0, 0}, xw[2] = {0, 0};
#pragma unroll
                for (int i = 0; i < 8; i += 2) {
                    const _Float16 h0 = (_Float16)vv[i], h1 = (_Float16)vv[i + 1];
                    f8_vmax = f8_range_track(f8_vmax, vv[i], vv[i + 1]);
                    fh[i] = h0;
                    fh[i + 1] = h1;
                    if (i & 2) f8_cross_pair<true>(vv[i], vv[i + 1], h0, h1, lw[i >> 2], xw[i >> 2]);
                    else f8_cross_pair<false>(vv[i], vv[i + 1], h0, h1, lw[i >> 2], xw[i >> 2]);
                }
                if (has) {
                    const int slot = pr * HC + lc;
                    dh[kg * NPIX + slot] = __builtin_bit_cast(u32x4, fh);
                    typedef __attribute__((ext_vector_type(2))) int i32x2;
                    i32x2 *d8 = reinterpret_cast<i32x2 *>(dl);
                    i32x2 q0, q1;
                    q0[0] = lw[0]; q0[1] = lw[1];
                    q1[0] = xw[0]; q1[1] = xw[1];
                    d8[(((kg >> 1)) * NPIX + slot) * 2 + (kg & 1)] = q0;
                    d8[((2 + (kg >> 1)) * NPIX + slot) * 2 + (kg & 1)] = q1;
                }
            } else
            if constexpr (ADBUF) {
                // branch-free (lanes without an item write a scratch slot): keeps the transform in the MFMAs' basic block
                const int slot = kg * NPIX + pr * HC + lc;
                u32x4 *ph = has ? dh + slot : reinterpret_cast<u32x4 *>(s_red);
                u32x4 *pl = has ? dl + slot : reinterpret_cast<u32x4 *>(s_red) + 1;
                *ph = *reinterpret_cast<u32x4 *>(&vh);
                *pl = *reinterpret_cast<u32x4 *>(&vl);
            } else if (has) {
                const int slot = kg * NPIX + pr * HC + lc;
                dh[slot] = *reinterpret_cast<u32x4 *>(&vh);
                if constexpr (NSPLIT > 1) dl[slot] = *reinterpret_cast<u32x4 *>(&vl);
            }
        }
    };
    if constexpr (PRIV) {
        // ---- private-slice main loop: 2 workgroup barriers per chunk (both around the shared activation image)
        // stage (0, 0) goes straight into the ring, stage (0, 1) (or (1, 0)) waits in wregs
#pragma unroll
        for (int i = 0; i < PITER; ++i) s_wp[i * 64 + lane] = wregs[i];
#pragma unroll
        for (int j = 0; j < TPS; ++j) load_w_tap(1, j);
        int chunk = 0;
        do {       // (bottom-tested by hand: hipcc left the exit test at the top of this loop and then copied all 64
                   //  accumulator registers into the exit block's set and back on every iteration)
            GTTS_SYNC();                               // every wave is done with the previous chunk's image (and s_par is written)
            stage_act(chunk, s_ah, s_al);
            GTTS_SYNC();
            load_act(min(chunk + 1, a.nchunk - 1));    // unconditional: see the comment on prefetches below
#pragma unroll
            for (int stage = 0; stage < NST; ++stage) {
                const int g = chunk * NST + stage;
#pragma unroll
                for (int j = 0; j < TPS; ++j) {
                    bf16x8 wh[MF], wl[MF];
#pragma unroll
                    for (int mi = 0; mi < MF; ++mi) {
                        const int wi = (j * NKG + kg_l) * (MF * 32) + mi * 32 + l31;
                        wh[mi] = *reinterpret_cast<const bf16x8 *>(&s_wp[wi]);
                        if (NSPLIT > 1) wl[mi] = *reinterpret_cast<const bf16x8 *>(&s_wp[wi + TPS * NKG * MF * 32]);
                    }
                    // slot j is free as soon as the reads above are ISSUED (DS operations of a wave execute in order):
                    // refill it with tap j of the next stage, re-issue the global load of the stage after that
#pragma unroll
                    for (int sp = 0; sp < NSPLIT; ++sp) s_wp[(sp * TPS + j) * 64 + lane] = wregs[sp * TPS + j];
                    load_w_tap(g + 2, j);
#pragma unroll
                    for (int np = 0; np < NF; np += 2) {
                        bf16x8 xh[2], xl[2];
#pragma unroll
                        for (int q = 0; q < 2; ++q) {
                            const int r = wn * NF + np + q;
                            const int xi = kg_l * NPIX + (r + stage) * HC + j + l31;
                            xh[q] = *reinterpret_cast<const bf16x8 *>(&s_ah[xi]);
                            if (NSPLIT > 1) xl[q] = *reinterpret_cast<const bf16x8 *>(&s_al[xi]);
                        }
#pragma unroll
                        for (int mi = 0; mi < MF; ++mi)
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                if (NSPLIT > 1) {
                                    acc[mi][np + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[mi], xh[q], acc[mi][np + q], 0, 0, 0);
                                    acc[mi][np + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[mi], xl[q], acc[mi][np + q], 0, 0, 0);
                                }
                                acc[mi][np + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[mi], xh[q], acc[mi][np + q], 0, 0, 0);
                            }
                    }
                    // keep hipcc from hoisting the next taps' fragment reads over this tap's MFMAs: the register budget
                    // (168 at three waves per SIMD) has room for one tap's fragments, not for two
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } while (++chunk < a.nchunk);
    }
    if constexpr (ADBUF) {
        GTTS_SYNC();                                   // s_par is visible
        stage_act(0, s_ah, s_al);
        if (1 < a.nchunk) load_act(1);
    }
    for (int chunk = 0; chunk < (PRIV ? 0 : a.nchunk); ++chunk) {
        if constexpr (!ADBUF) GTTS_SYNC();   // previous chunk's MFMAs are done with s_a* / s_w (and s_par is written)
        TR_MARK(0);
#if GTTS_TRACE
        __builtin_amdgcn_s_waitcnt(0x0f70);     // vmcnt(0): separates the load wait (phase 7) from the transform (phase 1)
        TR_MARK(7);
#endif
        if constexpr (!ADBUF) stage_act(chunk, s_ah, s_al);
        TR_MARK(1);
#pragma unroll
        for (int stage = 0; stage < NST; ++stage) {
            const u32x4 *s_wc = s_w;                        // weight buffer read by this stage's MFMAs
            if constexpr (WDMA) {
                // this stage's weights were DMA'd during the previous stage; every wave waits for its own pieces
                // (and, conservatively, every other load in flight) and the barrier publishes them -- and, at
                // stage 0, the activation image written above.  The same barrier frees the other buffer.
                __builtin_amdgcn_s_waitcnt(0x0f70);         // vmcnt(0)
                GTTS_SYNC();
                s_wc = s_w + ((chunk * NST + stage) & 1) * WBLK16;
            } else {
                if (stage > 0 || ADBUF) { GTTS_SYNC(); TR_MARK(5); }   // previous stage's MFMAs are done with s_w
#pragma unroll
                for (int i = 0; i < (GTTS_EXP == 4 ? 0 : WITER); ++i)
                    if (NW16 % 256 == 0 || tid + i * 256 < NW16) s_w[tid + i * 256] = wregs[i];
                TR_MARK(2);
                GTTS_SYNC();
                TR_MARK(3);
            }
            // ---- prefetch behind the MFMAs: next weight block (one stage ahead) and, as early as the staging
            // registers are free again, the next activation chunk (a whole chunk of MFMAs ahead)
            // Both prefetches are UNCONDITIONAL (the last chunk re-requests its own blocks; the registers are never read).
            // A prefetch under `if (chunk + 1 < nchunk)` puts a control-flow merge between the loads and the next
            // stage's s_waitcnt: hipcc then has to assume the shorter queue and emits vmcnt(5) for the weight stage,
            // which also waits for 11 of the 16 activation loads issued one stage (36 MFMAs) earlier -- the HBM latency
            // of the NEXT chunk's tile was exposed once per chunk.  Straight-line code gets the exact vmcnt(21).
            if (GTTS_EXP == 4) {
            } else if (stage + 1 < NST) load_w(chunk, stage + 1);
            else if (ADBUF || UNCOND_PF) load_w(min(chunk + 1, a.nchunk - 1), 0);
            else if (chunk + 1 < a.nchunk) load_w(chunk + 1, 0);
            if (!ADBUF && stage == 0) {
                if (UNCOND_PF) load_act(min(chunk + 1, a.nchunk - 1));
                else if (chunk + 1 < a.nchunk) load_act(chunk + 1);
            }
            if (ADBUF && stage == NST - 1) {
                // next chunk's image goes to the other buffer while this stage's MFMAs run (its last readers finished two
                // barriers ago); the chunk after that is prefetched into the freed staging registers.  Unconditional (the
                // last chunk restages itself into the idle buffer) so that the transform shares a basic block with the
                // MFMAs and the scheduler can place its VALU work between them.
                const int nc = min(chunk + 1, a.nchunk - 1);
                stage_act(nc, s_ah + ((chunk + 1) & 1) * AIMG, s_al + ((chunk + 1) & 1) * AIMG);
                load_act(min(chunk + 2, a.nchunk - 1));
            }
            const u32x4 *s_xh = s_ah + (ADBUF ? (chunk & 1) * AIMG : 0), *s_xl = s_al + (ADBUF ? (chunk & 1) * AIMG : 0);

#pragma unroll
            for (int j = 0; j < TPS; ++j) {
                int po[NF];
#pragma unroll
                for (int ni = 0; ni < NF; ++ni) {
                    const int r = wn * NF + ni;
                    if (MODE == CONV_C3 || MODE == CONV_C7) po[ni] = (r + stage) * HC + j;
                    else if (MODE == CONV_DN) po[ni] = (2 * r + stage) * HC + (j == 1 ? 33 : (j >> 1));
                    else if (MODE == CONV_UP) {
                        int dy = ph_y == 0 ? (stage == 0 ? 0 : -1) : (stage == 0 ? 1 : 0);
                        int dx = ph_x == 0 ? (j == 0 ? 0 : -1) : (j == 0 ? 1 : 0);
                        po[ni] = (r + 1 + dy) * HC + 1 + dx;
                    } else po[ni] = r * HC;
                }
                if constexpr (NSPLIT == 3) {
                    // 32 channels of one tap: two fp16 k-steps (hi * hi) + one fp8 K = 64 step (both cross terms); the per-accumulator
                    // order k-step 0, k-step 1, fp8 is the same in every instance of a layer (batch-size independent results)
                    f16x8 fwh[2][MF], fxh[2][NF];
                    i32x8 w8[MF], x8[NF];
#pragma unroll
                    for (int kc = 0; kc < 2; ++kc) {
#pragma unroll
                        for (int mi = 0; mi < MF; ++mi)
                            fwh[kc][mi] = *reinterpret_cast<const f16x8 *>(&s_wc[(j * NKG + kc * 2 + kg_l) * MT + m0 + mi * 32 + l31]);
#pragma unroll
                        for (int ni = 0; ni < NF; ++ni)
                            fxh[kc][ni] = *reinterpret_cast<const f16x8 *>(&s_xh[(kc * 2 + kg_l) * NPIX + po[ni] + l31]);
                    }
#pragma unroll
                    for (int mi = 0; mi < MF; ++mi) {
                        const int wi = TPS * NKG * MT + (j * NKG + kg_l * 2) * MT + m0 + mi * 32 + l31;
                        const u32x4 q0 = s_wc[wi], q1 = s_wc[wi + MT];
                        w8[mi][0] = (int)q0[0]; w8[mi][1] = (int)q0[1]; w8[mi][2] = (int)q0[2]; w8[mi][3] = (int)q0[3];
                        w8[mi][4] = (int)q1[0]; w8[mi][5] = (int)q1[1]; w8[mi][6] = (int)q1[2]; w8[mi][7] = (int)q1[3];
                    }
#pragma unroll
                    for (int ni = 0; ni < NF; ++ni) {
                        const int xi = (kg_l * 2) * NPIX + po[ni] + l31;
                        const u32x4 q0 = s_xl[xi], q1 = s_xl[xi + NPIX];
                        x8[ni][0] = (int)q0[0]; x8[ni][1] = (int)q0[1]; x8[ni][2] = (int)q0[2]; x8[ni][3] = (int)q0[3];
                        x8[ni][4] = (int)q1[0]; x8[ni][5] = (int)q1[1]; x8[ni][6] = (int)q1[2]; x8[ni][7] = (int)q1[3];
                    }
#pragma unroll
                    for (int mi = 0; mi < MF; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NF; ++ni) {
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwh[0][mi], fxh[0][ni], acc[mi][ni], 0, 0, 0);
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fwh[1][mi], fxh[1][ni], acc[mi][ni], 0, 0, 0);
                            acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(w8[mi], x8[ni], acc[mi][ni], 0, 0, 0, 0, 0, 0);
                        }
                } else
#pragma unroll
                for (int kc = 0; kc < KCH; ++kc) {
                    bf16x8 wh[MF], wl[MF], xh[NF], xl[NF];
#if GTTS_EXP == 2
                    for (int mi = 0; mi < MF; ++mi) { wh[mi] = __builtin_bit_cast(bf16x8, wregs[0]); wl[mi] = wh[mi]; }
                    for (int ni = 0; ni < NF; ++ni) { xh[ni] = __builtin_bit_cast(bf16x8, wregs[1]); xl[ni] = xh[ni]; }
#else
#pragma unroll
                    for (int mi = 0; mi < MF; ++mi) {
                        int wi = (j * NKG + kc * 2 + kg_l) * MT + m0 + mi * 32 + l31;
                        wh[mi] = *reinterpret_cast<const bf16x8 *>(&s_wc[wi]);
                        if (NSPLIT > 1) wl[mi] = *reinterpret_cast<const bf16x8 *>(&s_wc[wi + TPS * NKG * MT]);
                    }
#pragma unroll
                    for (int ni = 0; ni < NF; ++ni) {
                        int xi = (kc * 2 + kg_l) * NPIX + po[ni] + l31;
                        xh[ni] = *reinterpret_cast<const bf16x8 *>(&s_xh[xi]);
                        if (NSPLIT > 1) xl[ni] = *reinterpret_cast<const bf16x8 *>(&s_xl[xi]);
                    }
#endif
#if GTTS_EXP == 1
#pragma unroll
                    for (int mi = 0; mi < MF; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NF; ++ni) {      // consume the fragments with a handful of VALU ops
                            const u32x4 p = __builtin_bit_cast(u32x4, wh[mi]) ^ __builtin_bit_cast(u32x4, xh[ni]);
                            u32x4 q = p;
                            if (NSPLIT > 1) q = __builtin_bit_cast(u32x4, wl[mi]) ^ __builtin_bit_cast(u32x4, xl[ni]);
                            acc[mi][ni][0] += __builtin_bit_cast(float, (p[0] ^ p[1] ^ p[2] ^ p[3] ^ q[0] ^ q[1] ^ q[2] ^ q[3]) & 0x3fffffffu);
                        }
#else
                    // (Issuing the three bf16x3 passes pass-major over the four accumulators, pinned with sched_barrier -- every
                    // same-accumulator pair then exactly four issue slots apart instead of hipcc's mix of 2..8 -- measured
                    // 7.164 vs 7.176 ms per U-Net call over three alternating repeats: no difference; not kept.)
#pragma unroll
                    for (int mi = 0; mi < MF; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NF; ++ni) {
                            if (NSPLIT > 1) {
                                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[mi], xh[ni], acc[mi][ni], 0, 0, 0);
                                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[mi], xl[ni], acc[mi][ni], 0, 0, 0);
                            }
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[mi], xh[ni], acc[mi][ni], 0, 0, 0);
                        }
#endif
                }
            }
            TR_MARK(4);
        }
    }
#if GTTS_TRACE
    if (tr_on && lane == 0) {
        tr_sum[6] = __builtin_amdgcn_s_memtime() - tr_t0;
        for (int i = 0; i < 8; ++i) g_conv_trace[((blockIdx.x / 97) * 4 + wave) * 8 + i] = tr_sum[i];
        g_conv_trace[2048 + ((blockIdx.x / 97) * 4 + wave) * 2] = tr_t0 - tr_entry;
    }
#endif

    if constexpr (NSPLIT == 3) f8_range_note(a.sat, f8_vmax);
    // ---------------------------------------------------------------- epilogue
    const int HWout = a.Hout * a.Wout;
    float st1[MF][4], st2[MF][4];
#pragma unroll
    for (int mi = 0; mi < MF; ++mi)
#pragma unroll
        for (int q = 0; q < 4; ++q) { st1[mi][q] = 0.f; st2[mi][q] = 0.f; }

    // Output (and the fused-tail / residual input) go through buffer descriptors of this sample's tensors: per-lane
    // byte offset = pixel + the lane's 4-channel sub-row, per-channel offset in an SGPR -> no 64-bit VALU address
    // arithmetic per access.
    const int out_bytes = a.cout * HWout * AB;
    const __amdgpu_buffer_rsrc_t rs_out = uniform_rsrc(reinterpret_cast<AT *>(a.out) + (size_t)b * a.cout * HWout, out_bytes);
    const __amdgpu_buffer_rsrc_t rs_ex = uniform_rsrc(
        reinterpret_cast<const AT *>(EPI == EPI_TAIL ? a.eh : (EPI == EPI_ATTN ? a.eres : (const void *)a.out)) + (size_t)b * a.cout * HWout,
        out_bytes);
    const int ch0 = __builtin_amdgcn_readfirstlane(cot * MT + m0);     // first channel of this wave's fragments
#pragma unroll
    for (int ni = 0; ni < NF; ++ni) {
        const int r = wn * NF + ni;
        int oy = y0 + r, ox = x0 + l31;
        if (MODE == CONV_UP) { oy = 2 * oy + ph_y; ox = 2 * ox + ph_x; }
        const bool pix_ok = oy < a.Hout && ox < a.Wout;
        if (pix_ok) {      // one exec-mask region per row; straight-line code inside
            float m_out = 0.f;
            if (EPI == EPI_TAIL) m_out = a.mask[(size_t)b * a.T + ((size_t)ox << a.lvl_out)];
            if (EPI == EPI_PLAIN && a.omask) m_out = a.omask[(size_t)b * a.Wout + ox];
            const int voff = (oy * a.Wout + ox + 4 * kg_l * HWout) * AB;
#pragma unroll
            for (int mi = 0; mi < MF; ++mi) {
                float ex[16];
                if (EPI == EPI_TAIL || EPI == EPI_ATTN) {
#pragma unroll
                    for (int rg = 0; rg < 16; ++rg) {
                        const int soff = (ch0 + mi * 32 + (rg & 3) + 8 * (rg >> 2)) * HWout * AB;
                        ex[rg] = ld_act<AT>(rs_ex, voff, soff);
                    }
                }
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const int col = m0 + mi * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * kg_l;   // channel inside the tile
                    float v;
                    if constexpr (NSPLIT == 3) v = fmaf(acc[mi][ni][rg], 1.0f / (float)(1 << F8_S), s_epi[col]);   // accumulators hold 2^S x the sum (common.h)
                    else v = acc[mi][ni][rg] + s_epi[col];
                    if (EPI == EPI_TAIL) {
                        const float y = fmaf(ex[rg], s_epi[MT + col], s_epi[2 * MT + col]);
                        v = fmaf(mish_f(y), m_out, v);
                    } else if (EPI == EPI_ATTN) {
                        v += ex[rg];
                    } else if (EPI == EPI_PLAIN) {
                        if (a.omask) v *= m_out;
                    }
                    const int soff = (ch0 + mi * 32 + (rg & 3) + 8 * (rg >> 2)) * HWout * AB;
                    st_act<AT>(v, rs_out, voff, soff);
                    if (EPI == EPI_STATS) {
                        // octet rg>>2 of this 32-channel fragment (static index); octets -> groups below
                        st1[mi][rg >> 2] += v;
                        st2[mi][rg >> 2] = fmaf(v, v, st2[mi][rg >> 2]);
                    }
                }
            }
        }
    }

#if GTTS_TRACE
    unsigned long long tr_e[4] = {0, 0, 0, 0};
    if (tr_on) tr_e[0] = __builtin_amdgcn_s_memtime();
#endif
    if (EPI == EPI_STATS) {
        // wave reduce -> LDS -> fixed-order combine: one partial per (workgroup, group), deterministic
        const int gs = a.cout / a.groups;                   // channels per GroupNorm group
        // (s_red is not touched by the main loop: no barrier needed before writing it; the one below orders LDS only,
        // a full __syncthreads() would also wait for the 64 output stores of every lane to complete)
        {
            // all 8 MF wave sums in one transposing reduction (common.h); value index = which * 4 MF + mi * 4 + q
            constexpr int V = 8 * MF;
            float vals[V], tot[V / 4];
#pragma unroll
            for (int mi = 0; mi < MF; ++mi)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    vals[mi * 4 + q] = st1[mi][q];
                    vals[4 * MF + mi * 4 + q] = st2[mi][q];
                }
            wave_sums_transposed<V>(vals, tot);
            if ((lane & 15) == 0) {
                const int r = lane >> 4;
#pragma unroll
                for (int k = 0; k < V / 4; ++k) {
                    const int vi = k + (V / 4) * (r & 1) + (V / 2) * (r >> 1);
                    const int which = vi / (4 * MF), mi = (vi % (4 * MF)) >> 2, q = vi & 3;
                    s_red[((wave * MF + mi) * 4 + q) * 2 + which] = tot[k];
                }
            }
        }
#if GTTS_TRACE
        if (tr_on) tr_e[1] = __builtin_amdgcn_s_memtime();
#endif
        lds_barrier();
#if GTTS_TRACE
        if (tr_on) tr_e[2] = __builtin_amdgcn_s_memtime();
#endif
        const int gpw = MT / gs > 0 ? MT / gs : 1;     // groups covered by this workgroup
        // Partial slots: one per (tile, group) -- or, for layers that small launches tile with half-height tiles
        // (a.stat_rows), one per (ROW PAIR, 32-column block, group): a wave row covers exactly one row pair in either tiling,
        // its wave-level sums are the same numbers in the same order, so the statistics (and with them every output) are
        // bit-identical whatever the batch size picked.  All slots of a workgroup are written by wave 0 (gpw * WN <= 64).
        const int nsl = a.stat_rows ? WN : 1;
        if (tid < gpw * nsl) {
            const int gl = a.stat_rows ? tid % gpw : tid, wsel = a.stat_rows ? tid / gpw : -1;
            const int g = (cot * MT) / gs + gl;        // global group index
            const int prow = ty * WN + wsel;           // row pair of this wave row (NF == 2)
            if (g < a.groups && (wsel < 0 || 2 * prow < a.Hout)) {
                // this group's octets inside the tile (gs is a multiple of 8; a group wider than the tile covers all of it):
                // octet o belongs to fragment o / 4 = (wave row wmm, mi) and was summed by the WN waves of that row
                const int noct = (gs < MT ? gs : MT) >> 3, o0 = gl * noct;
                float s1 = 0.f, s2 = 0.f;
                for (int o = o0; o < o0 + noct; ++o) {
                    const int f = o >> 2, q = o & 3, wmm = f / MF, mi = f - wmm * MF;
#pragma unroll
                    for (int wc = 0; wc < WN; ++wc) {
                        if (wsel >= 0 && wc != wsel) continue;
                        const int w = wmm * WN + wc;
                        s1 += s_red[((w * MF + mi) * 4 + q) * 2 + 0];
                        s2 += s_red[((w * MF + mi) * 4 + q) * 2 + 1];
                    }
                }
                const int slot = a.stat_rows ? prow * a.tiles_x + tx : tile;
                float *p = a.partials + (((size_t)b * a.nparts + slot) * a.groups + g) * 2;
                if (a.ticket != nullptr) {
                    // write-through (sc1) stores: visible device-wide once vmcnt has drained, WITHOUT a release fence --
                    // an agent-scope release is buffer_wbl2, which writes back every dirty line of this XCD's L2 (all the
                    // output tiles just stored by every workgroup on the XCD): measured +15...50 % on the whole kernel
                    __hip_atomic_store(p, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(p + 1, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                } else {
                    p[0] = s1;
                    p[1] = s2;
                }
            }
        }
        // ---- fused GroupNorm finalize (diffusion.py:53: eps 1e-5, biased variance).  Hand-off in the write-through form
        // of the agent-scope recipe: the wave that stored the partials (sc1 stores) drains its stores and draws a ticket
        // (relaxed, agent scope); the workgroup that draws the last ticket of its sample reads ALL partials of the sample
        // with sc1 loads and reduces them in a fixed order (fp64), so the result does not depend on which workgroup happens
        // to be last.
        if (a.ticket != nullptr && wave == 0) {
#if GTTS_FENCED_FINALIZE
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // the textbook form (see common.h): buffer_wbl2 sc1
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(a.ticket + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = __builtin_amdgcn_readfirstlane(old);
            const unsigned total = (unsigned)(a.tiles_x * a.tiles_y * ncot);
            if (old == total - 1) {
#if GTTS_FENCED_FINALIZE
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
                // This tail is serial (the last workgroup of the last sample runs it alone), so it is organised for few
                // dependent memory round trips: 16 lanes walk the slots of a PAIR of groups, two 8-byte sc1 loads per slot
                // and eight slots in flight per lane; then 8 lanes per group finish.  Fixed order: deterministic.
                const int g = lane >> 3, sub = lane & 7;          // final owner: 8 lanes per group, 8 groups
                double s1 = 0.0, s2 = 0.0;
                if (a.groups == 8) {
                    const int gp = lane >> 4, s16 = lane & 15;    // group pair (2 gp, 2 gp + 1), slot residue
                    double t1[2] = {0.0, 0.0}, t2[2] = {0.0, 0.0};
                    const float *pp = a.partials + ((size_t)b * a.nparts * 8 + 2 * gp) * 2;
#pragma unroll 8
                    for (int i = s16; i < a.nparts; i += 16) {
#pragma unroll
                        for (int h = 0; h < 2; ++h) {
                            const unsigned long long u = __hip_atomic_load(
                                reinterpret_cast<const unsigned long long *>(pp + ((size_t)i * 8 + h) * 2), __ATOMIC_RELAXED,
                                __HIP_MEMORY_SCOPE_AGENT);
                            t1[h] += (double)__builtin_bit_cast(float, (unsigned)u);
                            t2[h] += (double)__builtin_bit_cast(float, (unsigned)(u >> 32));
                        }
                    }
                    // lanes (gp, s16) -> group 2 gp + (s16 >> 3) keeps its own half and takes the partner's (s16 ^ 8)
                    const int hsel = s16 >> 3;
                    const double give1 = hsel ? t1[0] : t1[1], give2 = hsel ? t2[0] : t2[1];
                    s1 = (hsel ? t1[1] : t1[0]) + __shfl_xor(give1, 8, 64);
                    s2 = (hsel ? t2[1] : t2[0]) + __shfl_xor(give2, 8, 64);
                } else if (g < a.groups) {
                    const float *pp = a.partials + ((size_t)b * a.nparts * a.groups + g) * 2;
                    // (sum, sum of squares) pairs as one 8-byte sc1 load each, eight in flight per lane
#pragma unroll 8
                    for (int i = sub; i < a.nparts; i += 8) {
                        const unsigned long long u = __hip_atomic_load(
                            reinterpret_cast<const unsigned long long *>(pp + (size_t)i * a.groups * 2), __ATOMIC_RELAXED,
                            __HIP_MEMORY_SCOPE_AGENT);
                        s1 += (double)__builtin_bit_cast(float, (unsigned)u);
                        s2 += (double)__builtin_bit_cast(float, (unsigned)(u >> 32));
                    }
                }
#pragma unroll
                for (int o = 1; o < 8; o <<= 1) {
                    s1 += __shfl_xor(s1, o, 64);
                    s2 += __shfl_xor(s2, o, 64);
                }
                const double mean = s1 / (double)a.gn_count;
                double var = fma(-mean, mean, s2 / (double)a.gn_count);
                if (var < 0.0) var = 0.0;
                const double rstd = 1.0 / sqrt(var + 1e-5);
                // non-finite statistics = a non-finite output of this convolution (in GTTS_PREC_F16F8: an activation beyond the fp16 half's
                // range, or what such a value turned into downstream): one event with max |x| = inf in the call's range record (common.h);
                // the staging kernels' own running maximum ignores NaN (v_max_f32), so this is where a NaN becomes visible to the caller
                if (g < a.groups && sub == 0 && !(fabs(s1) < 1.0e300 && s2 < 1.0e300)) f8_range_note(a.sat, __builtin_inff());
                if (g < a.groups) {
                    for (int c = g * gs + sub; c < (g + 1) * gs; c += 8) {
                        const double sc = (double)a.gn_gamma[c] * rstd;
                        a.gn_sc[(size_t)b * a.cout + c] = (float)sc;
                        a.gn_sh[(size_t)b * a.cout + c] = (float)fma(-mean, sc, (double)a.gn_beta[c]);
                    }
                }
                if (lane == 0) __hip_atomic_store(a.ticket + b, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
            }
        }
    }
#if GTTS_TRACE
    if (tr_on && lane == 0) {
        const unsigned long long tend = __builtin_amdgcn_s_memtime();
        g_conv_trace[2048 + ((blockIdx.x / 97) * 4 + wave) * 2 + 1] = tend - tr_t0 - tr_sum[6];
        unsigned long long *q = g_conv_trace + 2560 + ((blockIdx.x / 97) * 4 + wave) * 4;
        q[0] = tr_e[0] - tr_t0 - tr_sum[6];      // bias + stores issued
        q[1] = tr_e[1] - tr_e[0];                // wave reductions + s_red
        q[2] = tr_e[2] - tr_e[1];                // barrier
        q[3] = tend - tr_e[2];                   // final combine
    }
#endif
}


template <int MODE, int WM, int WN, int MF, int KCH, int PRO, int EPI, int NSPLIT, int FULLC, typename AT = float, int NF = 2, int PRIV = 0>
static hipError_t launch_cfg(const ConvArgs &a_in, hipStream_t st) {
    using C = ConvCfg<MODE, WM, WN, MF, KCH, NF>;
    ConvArgs a = a_in;
    a.nchunk = (a.cin + 16 * KCH - 1) / (16 * KCH);
    const int th = (MODE == CONV_UP) ? a.Hin : a.Hout;   // tile space: input resolution for UP
    const int tw = (MODE == CONV_UP) ? a.Win : a.Wout;
    a.tiles_x = (tw + C::TC - 1) / C::TC;
    a.tiles_y = (th + C::TR - 1) / C::TR;
    const int ncot = (a.cout + C::MT - 1) / C::MT;
    a.stat_rows = (EPI == EPI_STATS && conv_rowpair_stats(MODE, a.cout, a.Hout, a.Wout)) ? 1 : 0;
    if (a.stat_rows && NF != 2) return hipErrorInvalidValue;       // a wave row must be one row pair (not the PRIV experiment)
    dim3 grid(a.tiles_x * a.tiles_y * ncot * (MODE == CONV_UP ? 4 : 1) * a.B);
    if (a.cout % C::MT != 0) return hipErrorInvalidValue;   // epilogue assumes whole output-channel tiles
    // buffer descriptors address one sample's tensor with 32-bit byte offsets
    const size_t lim = (size_t)1 << 31;
    const size_t in_c = (size_t)(PRO == PRO_IGLU ? 2 * a.cin : std::max(a.c0, a.c1));
    if (in_c * a.Hin * a.Win * sizeof(AT) >= lim || (size_t)a.cout * a.Hout * a.Wout * sizeof(AT) >= lim) return hipErrorInvalidValue;
    constexpr int WLDS16 = PRIV ? 4 * NSPLIT * C::TPS * C::NKG * MF * 32 : C::WBLK16;
    size_t smem = conv_smem_bytes(C::NPIX, C::NKG, WLDS16, a.cin, PRO, C::MT) +
                  (ConvWdma<MODE, WM, FULLC>::on ? (size_t)C::WBLK16 * 16 : 0) +
                  (ConvAdbuf<MODE, WM, FULLC>::on ? (size_t)C::NPIX * C::NKG * 16 * 2 : 0);
    if (smem < (size_t)GTTS_LDS_MIN) smem = (size_t)GTTS_LDS_MIN;
    if (MODE == CONV_C3 && NSPLIT == 2 && (EPI == EPI_STATS || EPI == EPI_PLAIN) && smem < (size_t)GTTS_C3_LDS_MIN) smem = (size_t)GTTS_C3_LDS_MIN;
    // hipFuncSetAttribute is per device: remember the largest size set on each device (atomics: launches may come
    // from several host threads; setting the attribute twice is harmless)
    static std::atomic<size_t> attr_set[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (smem > attr_set[dev].load(std::memory_order_relaxed)) {
        hipError_t e = hipFuncSetAttribute(
            reinterpret_cast<const void *>(&conv_mfma_kernel<MODE, WM, WN, MF, KCH, PRO, EPI, NSPLIT, FULLC, AT, NF, PRIV>),
            hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        attr_set[dev].store(smem, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((conv_mfma_kernel<MODE, WM, WN, MF, KCH, PRO, EPI, NSPLIT, FULLC, AT, NF, PRIV>), grid, dim3(256), smem, st, a);
    return hipGetLastError();
}

// See common.h.  LDS of the uniform-wave form: one 32-channel activation image (fp16 plane + fp8 plane), one weight stage of three
// taps, the prologue's per-channel parameters.
bool conv_f16f8_ok(int mode, int c0, int c1, int cout, int pro, int epi, int use_ws) {
    const int cin = c0 + c1;
    if (mode != CONV_C3 || epi != EPI_STATS || (pro != PRO_MASK && pro != PRO_GN)) return false;
    if (cin % 32 != 0 || (c1 != 0 && c0 % 32 != 0)) return false;
    // 64-channel layers: only on the persistent kernel (conv_ws.hip: four consumer + eight producer waves).  The uniform-wave 64-channel
    // tile was built and measured (round 5, same box, us per launch at B = 16): 219.6 / 209.9 (mask / GroupNorm prologue) against
    // 214.2 / 186.5 in bf16x3 -- two workgroups per CU (68 KB of LDS at 32-channel chunks) instead of three, twice the staging per MFMA,
    // and LDS fragment traffic that no longer hides behind the shorter MFMA phase; those instances are gone.
    if (cout == 64) return GTTS_WS && GTTS_F8_WS64 && use_ws && cin >= 64 && conv_ws_f8_fits(cin, pro, 64, cout);
    if (cout % 128 != 0) return false;
    const ConvGeom g = conv_geom(mode, cin, cout, 1);
    const int npix = (g.TR + 2) * 34, nkg = 2 * g.kch;
    // (two workgroups per CU up to 256 input channels with the GroupNorm prologue; wider layers -- DiffVC -- still fit one)
    return conv_smem_bytes(npix, nkg, g.tps * g.MT * 2 * nkg, cin, pro, g.MT) <= (size_t)160 * 1024;
}

// Small launches (B = 1, what Grad-TTS/inference.py runs): a 3x3 layer whose regular tiling yields fewer than GTTS_SMALL_WGS
// workgroups (half the CUs) is tiled with half-height tiles (128 x (2 x 32) / 64 x (4 x 32)): twice the workgroups, each half as long.
// The kernel is latency-bound in that regime (one workgroup per CU, one wave per SIMD), so the time roughly halves.
bool conv_small_tiles(int mode, int cout, int Hout, int Wout, int B) {
    if (mode != CONV_C3 || B <= 0) return false;
    ConvGeom g = conv_geom(mode, 64, cout);
    const long wgs = (long)B * ((Wout + 31) / 32) * ((Hout + g.TR - 1) / g.TR) * ((cout + g.MT - 1) / g.MT);
    return wgs < GTTS_SMALL_WGS;
}
// Layers that a small launch would tile with half-height tiles keep their GroupNorm partial sums per row pair (see the
// kernel's statistics epilogue): a property of the layer geometry, NOT of the batch size, so that results do not depend on
// how utterances are batched.
bool conv_rowpair_stats(int mode, int cout, int Hout, int Wout) { return conv_small_tiles(mode, cout, Hout, Wout, 1); }
// GroupNorm partial slots per sample that EPI_STATS writes for this layer
int conv_nparts(int mode, int cout, int Hout, int Wout) {
    if (conv_rowpair_stats(mode, cout, Hout, Wout)) return ((Wout + 31) / 32) * ((Hout + 1) / 2);
    ConvGeom g = conv_geom(mode, 64, cout);
    return ((Wout + 31) / 32) * ((Hout + g.TR - 1) / g.TR);
}

// (mode, tiling) x (prologue, epilogue) x precision -> template instance; must agree with conv_geom() in common.h.
// Only the combinations the op program uses are instantiated:
//   C3: (MASK | GN, STATS)   DN, UP: (MASK, PLAIN)   P1: (MASK, TAIL) | (PLAIN, ATTN) | (MASK, PLAIN: training)
template <int MODE, int WM, int WN, int MF, int PRO, int EPI>
static hipError_t launch_prec(const ConvArgs &a, hipStream_t st) {
    const bool fullc = a.cin % 16 == 0 && (a.c1 == 0 || a.c0 % 16 == 0);
    if constexpr (MODE == CONV_C3 && EPI == EPI_STATS && (PRO == PRO_MASK || PRO == PRO_GN)) {
        // GTTS_PREC_F16F8: the layer's weights are packed in the f16 + fp8 format exactly when conv_f16f8_ok says so (plan.hip)
        if (a.f16f8 && conv_f16f8_ok(MODE, a.c0, a.c1, a.cout, PRO, EPI, a.use_ws)) {
            if (a.act_bf16 || a.nsplit != 2) return hipErrorInvalidValue;
            if constexpr (WM == 2) {      // (conv_f16f8_ok: 128-channel cout tiles only)
                if (conv_small_tiles(MODE, a.cout, a.Hout, a.Wout, a.B))      // half-height tiles, as below
                    return launch_cfg<MODE, 4, 1, 1, 2, PRO, EPI, 3, 1, float, 2, 0>(a, st);
                return launch_cfg<MODE, WM, WN, MF, 2, PRO, EPI, 3, 1, float, 2, 0>(a, st);
            }
            return hipErrorInvalidValue;
        }
    }
    if constexpr (MODE == CONV_C3 && PRO != PRO_IGLU) {
        // half-height tiles for small launches (conv_small_tiles): same cout tile, waves re-arranged to 32 channels x 2 rows
        if (fullc && !a.act_bf16 && a.nsplit > 1 && conv_small_tiles(MODE, a.cout, a.Hout, a.Wout, a.B)) {
            if constexpr (WM == 2) return launch_cfg<MODE, 4, 1, 1, 1, PRO, EPI, 2, 1, float, 2, 0>(a, st);
            else return launch_cfg<MODE, 2, 2, 1, 1, PRO, EPI, 2, 1, float, 2, 0>(a, st);
        }
    }
    // ragged channel counts only occur on first layers (stacked input, 1-channel reference): PRO_MASK variants
    constexpr bool ragged_ok = PRO == PRO_MASK && (MODE == CONV_C3 || MODE == CONV_P1);
#ifdef GTTS_LEAN      // A/B builds: a code object without the single-pass bf16 / bf16-storage instances (do unlaunched kernels cost time?)
    if (a.act_bf16 || a.nsplit == 1) return hipErrorInvalidValue;
#endif
#ifndef GTTS_LEAN
    if (a.act_bf16) {
        // bf16 storage (BASELINE config 3): single-pass bf16 MFMA only, Grad-TTS op set only (no InstanceNorm-GLU convs)
        if constexpr (PRO != PRO_IGLU && !(MODE == CONV_C3 && EPI == EPI_PLAIN)) {
            if (a.nsplit > 1) return hipErrorInvalidValue;
            if (fullc) return launch_cfg<MODE, WM, WN, MF, 1, PRO, EPI, 1, 1, __bf16>(a, st);
            if constexpr (ragged_ok) return launch_cfg<MODE, WM, WN, MF, 1, PRO, EPI, 1, 0, __bf16>(a, st);
        }
        return hipErrorInvalidValue;
    }
#endif
    if constexpr (GTTS_PRIV && MODE == CONV_C3 && PRO != PRO_IGLU) {
        // bf16x3 3x3 convolutions on whole chunks: private weight slices (same workgroup tiles, waves re-arranged to
        // 32 channels x 4 rows each): 128 x (4 x 32) as 4 x 1 waves, 64 x (8 x 32) as 2 x 2 waves
        if (fullc && a.nsplit > 1) {
            if constexpr (WM == 2) return launch_cfg<MODE, 4, 1, 1, 1, PRO, EPI, 2, 1, float, 4, 1>(a, st);
            else return launch_cfg<MODE, 2, 2, 1, 1, PRO, EPI, 2, 1, float, 4, 1>(a, st);
        }
    }
#ifdef GTTS_LEAN
    if (fullc) return launch_cfg<MODE, WM, WN, MF, 1, PRO, EPI, 2, 1>(a, st);
    if constexpr (ragged_ok) return launch_cfg<MODE, WM, WN, MF, 1, PRO, EPI, 2, 0>(a, st);
#else
    if (fullc)
        return a.nsplit > 1 ? launch_cfg<MODE, WM, WN, MF, 1, PRO, EPI, 2, 1>(a, st)
                            : launch_cfg<MODE, WM, WN, MF, 1, PRO, EPI, 1, 1>(a, st);
    if constexpr (ragged_ok)
        return a.nsplit > 1 ? launch_cfg<MODE, WM, WN, MF, 1, PRO, EPI, 2, 0>(a, st)
                            : launch_cfg<MODE, WM, WN, MF, 1, PRO, EPI, 1, 0>(a, st);
#endif
    return hipErrorInvalidValue;
}

hipError_t launch_conv(int mode, const ConvArgs &a, hipStream_t st) {
    const bool wide = a.cout > 64;
    // Block convolutions on whole 16-channel chunks: the persistent wave-specialised kernel (conv_ws.hip)
    if (a.use_ws && conv_ws_eligible(mode, a.c0, a.c1, a.cout, a.pro, a.epi, a.nsplit, a.f16f8)) return launch_conv_ws(a, st);
    switch (mode) {
        case CONV_C3:
            if (a.epi == EPI_PLAIN) {          // DiffVC RefBlock convolutions (InstanceNorm statistics are a separate pass)
                if (a.pro == PRO_MASK)
                    return wide ? launch_prec<CONV_C3, 2, 2, 2, PRO_MASK, EPI_PLAIN>(a, st)
                                : launch_prec<CONV_C3, 1, 4, 2, PRO_MASK, EPI_PLAIN>(a, st);
                if (a.pro == PRO_IGLU)
                    return wide ? launch_prec<CONV_C3, 2, 2, 2, PRO_IGLU, EPI_PLAIN>(a, st)
                                : launch_prec<CONV_C3, 1, 4, 2, PRO_IGLU, EPI_PLAIN>(a, st);
                break;
            }
            if (a.epi != EPI_STATS) break;
            if (a.pro == PRO_MASK)
                return wide ? launch_prec<CONV_C3, 2, 2, 2, PRO_MASK, EPI_STATS>(a, st)
                            : launch_prec<CONV_C3, 1, 4, 2, PRO_MASK, EPI_STATS>(a, st);
            if (a.pro == PRO_GN)
                return wide ? launch_prec<CONV_C3, 2, 2, 2, PRO_GN, EPI_STATS>(a, st)
                            : launch_prec<CONV_C3, 1, 4, 2, PRO_GN, EPI_STATS>(a, st);
            break;
        case CONV_C7:          // DiffVC PostNet Block (postnet.py:15-23): 64-cout tiles, fp32 storage, bf16x3
            if (a.epi != EPI_STATS || a.act_bf16 || a.nsplit != 2) break;
            if (a.cin % 16 != 0) break;
            if (a.pro == PRO_MASK) return launch_cfg<CONV_C7, 1, 4, 2, 1, PRO_MASK, EPI_STATS, 2, 1>(a, st);
            if (a.pro == PRO_GN) return launch_cfg<CONV_C7, 1, 4, 2, 1, PRO_GN, EPI_STATS, 2, 1>(a, st);
            break;
        case CONV_DN:
            if (a.pro != PRO_MASK || a.epi != EPI_PLAIN) break;
            return wide ? launch_prec<CONV_DN, 2, 2, 2, PRO_MASK, EPI_PLAIN>(a, st)
                        : launch_prec<CONV_DN, 2, 2, 1, PRO_MASK, EPI_PLAIN>(a, st);
        case CONV_UP:
            if (a.pro != PRO_MASK || a.epi != EPI_PLAIN) break;
            if (conv_up4_eligible(a)) return launch_conv_up4(a, st);
            // a GTTS_PREC_F16F8 plan packed this layer's weights in the f16 + fp8 format: only conv_up.hip reads it -- fail, never misread
            if (a.f16f8 && conv_up4_f16f8_ok(a.cin, a.cout)) return hipErrorInvalidValue;
            return wide ? launch_prec<CONV_UP, 2, 2, 2, PRO_MASK, EPI_PLAIN>(a, st)
                        : launch_prec<CONV_UP, 1, 4, 2, PRO_MASK, EPI_PLAIN>(a, st);
        case CONV_P1:
            if (a.pro == PRO_MASK && a.epi == EPI_PLAIN)        // training: res_conv / to_qkv / to_out and their data gradients (train.hip)
                return wide ? launch_prec<CONV_P1, 2, 2, 2, PRO_MASK, EPI_PLAIN>(a, st)
                            : launch_prec<CONV_P1, 1, 4, 2, PRO_MASK, EPI_PLAIN>(a, st);
            if (a.pro == PRO_MASK && a.epi == EPI_TAIL)
                return wide ? launch_prec<CONV_P1, 2, 2, 2, PRO_MASK, EPI_TAIL>(a, st)
                            : launch_prec<CONV_P1, 1, 4, 2, PRO_MASK, EPI_TAIL>(a, st);
            if (a.pro == PRO_PLAIN && a.epi == EPI_ATTN)
                return wide ? launch_prec<CONV_P1, 2, 2, 2, PRO_PLAIN, EPI_ATTN>(a, st)
                            : launch_prec<CONV_P1, 1, 4, 2, PRO_PLAIN, EPI_ATTN>(a, st);
            break;
    }
    return hipErrorInvalidValue;
}

}  // namespace gtts
