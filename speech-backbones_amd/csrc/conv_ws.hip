// conv_ws.hip -- the Block 3x3 convolution (Grad-TTS/model/diffusion.py:49-58) as a PERSISTENT, WAVE-SPECIALISED kernel.
//
// Why a second kernel: conv_mfma.hip runs uniform waves (every wave loads, transforms, stages and multiplies); its
// matrix pipe is 82 % busy inside the loop and 0.44 of peak over the launch (profiles/r02_conv_findings.txt): the VALU
// work of the apply-on-load prologue (GroupNorm affine + Mish + mask + time bias + bf16 hi/lo split) sits in the same
// instruction stream as the MFMAs, prologue / epilogue are exposed per tile and 2560 tiles quantise on 768 slots.
// Here a workgroup is 8 waves on one CU (one workgroup per CU, 256 registers per lane):
//   waves 0-3  CONSUMERS, one per SIMD: nothing but fragment reads (ds_read_b128), weight-fragment loads and MFMAs.
//              Wave tile (MF x 32) channels x (NF rows x 32 frames) with MF x NF = 10 accumulators (160 registers);
//              the three bf16x3 passes of a tap are issued pass-major over the ten accumulators, so two MFMAs on the
//              same accumulator are ten issue slots apart (the per-accumulator order wl.xh, wh.xl, wh.xh -- and with it
//              every output bit -- is that of conv_mfma.hip).  Weights never touch LDS: they are packed in fragment
//              order (pack.hip), so a lane's A fragment is one coalesced 16-byte buffer load, prefetched one tap
//              (30 MFMAs) ahead.
//   waves 4-7  PRODUCERS: global loads of the halo tile of the next 16-channel chunk, the producer's epilogue
//              (apply-on-load), the hi/lo split and the ds_write_b128 into a ring of activation images; their VALU
//              instructions issue beside the consumers' MFMAs (separate pipes) instead of between them.  The first
//              producer wave also combines the GroupNorm partial sums of a finished tile and runs the fused finalize.
// One s_barrier per chunk hands an image over (ring of RING images, producers RING-1 chunks ahead).  The grid is
// min(tiles, CUs) persistent workgroups walking tiles in XCD-banded order: the launch prologue is paid once per CU, and
// the 10-row tiles divide the 40 / 20 mel-bin levels, so B = 16 x 1024 frames is a whole number of rounds on 256 CUs.
// Small launches (B = 1 ... 4: fewer regular tiles than CUs) take a three-wave workgroup -- one consumer wave on a
// 32-channel x 5-row tile, two producer waves -- so that 4x as many CUs have work; see conv_ws_small().
//
// A GroupNorm partial slot is one consumer wave's sums over a (32-frame, 5-row) block in a fixed order in both forms, and the
// accumulation order of an output never depends on the form, so results do not depend on how utterances are batched.
#include "common.h"
#include "kernels.h"
#include <algorithm>
#include <atomic>
#include <type_traits>

// GTTS_WS_TRACE (diagnostic builds only, -DGTTS_DIAG): per-wave s_memtime sums of one layer (cin == cout == GTTS_TRACE_CIN)
// read back with gtts_debug_trace_ws().  Consumer waves: [0] barrier wait, [1] chunk loops, [2] tile epilogues, [3] items, [4] total.
// Producer waves: [0] barrier wait, [1] staging (load wait + transform + LDS write), [2] re-request (+ tile setup), [3] finish_tile.
// GTTS_WS_EXP (diagnostic builds only): timing ablations, results are WRONG.  1: the consumers never reload weights,
// 2: the producers never re-request activations, 3: the producers skip transform + LDS write, 4: the consumers never re-read
// B fragments, 5: no MFMAs, 6: no output stores, 7: no weight loads in the first three taps of a tile (f16 + fp8 form: 1, 4, 5, 6 in the consumer loop / epilogue; 2, 3 are common)
#ifndef GTTS_DIAG
#undef GTTS_WS_TRACE
#undef GTTS_WS_EXP
#endif
#ifndef GTTS_WS_EXP
#define GTTS_WS_EXP 0
#endif
#ifndef GTTS_WS_TRACE
#define GTTS_WS_TRACE 0
#endif
#ifndef GTTS_TRACE_CIN
#define GTTS_TRACE_CIN 128
#endif
#if GTTS_WS_TRACE
__device__ unsigned long long g_ws_trace[64 * 16 * 8];      // [workgroup][wave (up to 12)][slot]
extern "C" int gtts_debug_trace_ws(unsigned long long *dst, int n) {
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_ws_trace), sizeof(unsigned long long) * (n < 8192 ? n : 8192));
}
#define WT_NOW() (tr_on ? __builtin_amdgcn_s_memtime() : 0ull)
#define WT_ADD(slot, t1, t0) do { if (tr_on) tr_sum[slot] += (t1) - (t0); } while (0)
#else
#define WT_NOW() 0ull
#define WT_ADD(slot, t1, t0) do { } while (0)
#endif

// the eight-wave and three-wave forms of this kernel must agree bit for bit (results do not depend on the batch): every fused
// multiply-add is explicit and implicit contraction is off (see conv_mfma.hip)
#pragma clang fp contract(off)

namespace gtts {

// producer waves of the f16 + fp8 64-channel tile: 8 (three waves per SIMD, 168 registers: the consumers' fragment window shrinks to
// 96 cycles) or 4 (256 registers, 224-cycle window; the producers then bound the tile)
#ifndef GTTS_WS64_NPW
#define GTTS_WS64_NPW 8
#endif
// consumers of the f16 + fp8 form: weight sets in flight and fragment lead of the 64-channel tile (168 registers: 3 sets fit with a
// 96-cycle lead, 2 sets with 160), fragment lead of the 128-channel tile
#ifndef GTTS_WS64_NWS
#define GTTS_WS64_NWS 3
#endif
// producer waves of the small-launch form in f16 + fp8 (one consumer wave; bf16x3: two)
#ifndef GTTS_WS_SMALL_NPW8
#define GTTS_WS_SMALL_NPW8 2
#endif
#ifndef GTTS_WS64_LEAD
#define GTTS_WS64_LEAD 96
#endif
// the 64-channel f16 + fp8 tile (W64): sweep pipelining (0: a sweep reads its A fragments and first B rows itself; 1: the previous sweep
// prefetches them at input row GTTS_W64_PFROW; 2: A fragments only), B rows fetched GTTS_W64_DIST rows ahead, ring refill after row
// GTTS_W64_DMAROW (>= 2: every A fragment of the sweep has been consumed by then)
// cache policy of the f16 + fp8 epilogue's output stores (A/B builds): 0 plain, 2 nt (streaming), 16 sc1 (write-through), 17 sc0 sc1
#ifndef GTTS_WS_EPI_AUX
#define GTTS_WS_EPI_AUX 2      // measured (same box, ms per U-Net call): plain 6.432-6.457, nt 6.351, sc1 6.439, sc0 sc1 6.427
#endif
#ifndef GTTS_W64_PF
#define GTTS_W64_PF 0
#endif
#ifndef GTTS_W64_DIST
#define GTTS_W64_DIST 2
#endif
// cache policy of the f16 + fp8 epilogue's output stores (A/B builds): 0 plain, 2 nt (streaming), 16 sc1 (write-through), 17 sc0 sc1
#ifndef GTTS_WS_EPI_AUX
#define GTTS_WS_EPI_AUX 2      // measured (same box, ms per U-Net call): plain 6.432-6.457, nt 6.351, sc1 6.439, sc0 sc1 6.427
#endif
#ifndef GTTS_W64_PFROW
#define GTTS_W64_PFROW 5
#endif
#ifndef GTTS_W64_DMAROW
#define GTTS_W64_DMAROW 2
#endif
#ifndef GTTS_WS_LEAD
#define GTTS_WS_LEAD 160
#endif
// De-phased tile epilogues.  Persistent workgroups walk equal tiles in lock-step, so all 256 CUs reach their tile epilogue together
// and the chip's write path (6.0-6.8 TB/s: 13-14 B/clk per CU when every CU stores, 28 at 128 CUs, 33 / 50 at 64 with dword / 16-byte
// stores -- tools/probe/mem_probe.hip, profiles/r06_mem_probe.txt) bounds it: 10.7k cycles per 128-channel tile while the MFMAs idle.
// With GTTS_WS_DEPHASE = P > 1 phase groups the workgroups of group g (both halves / all quarters of every XCD) start
// g x (bytes a workgroup stores per tile) / GTTS_WS_DEPHASE_BPC cycles late: meant to leave only 1 / P of the CUs storing at any time.
// MEASURED (round 6, profiles/r06_dephase_trace.txt): it does not hold -- the late group catches up (level 0: started 2.9k cycles
// late, its first epilogue begins 1.1k after the early group's; the HBM-bound launch re-locks the phases) or stays 3.4k behind where
// an epilogue lasts 9k (level 2), and per-tile epilogue time does not move (5 277 vs 5 257; 8 867-9 153 vs 9 277 cycles):
// 153.3 vs 153.1 us on the dominant kernel, 6.52-6.56 ms per call with 0 / 2 / 4 groups.  Kept buildable, OFF.
#ifndef GTTS_WS_DEPHASE
#define GTTS_WS_DEPHASE 0
#endif
#ifndef GTTS_WS_DEPHASE_BPC
#define GTTS_WS_DEPHASE_BPC 28
#endif
#ifndef GTTS_WS_DEPHASE_MIN
#define GTTS_WS_DEPHASE_MIN 2
#endif
template <int WM, int WN, int MF, int NF, int NKGT = 2>
struct WsCfg {
    static constexpr int NCW = WM * WN;          // consumer wave ROWS x COLUMNS of the statistics layout: 4, 2 (64-channel tile) or 1
    // waves that actually run: the f16 + fp8 form (NKGT == 4) maps one wave to a 32-channel block (x a 5-row band for the 64-channel
    // tile), so the 64-channel tile still has four consumers; it takes eight producer waves (twice the staging per MFMA)
    static constexpr int NCWP = (NKGT == 4 && WM == 1 && WN == 2) ? 4 : NCW;
    static constexpr int NPW = NCW == 1 ? (NKGT == 4 ? GTTS_WS_SMALL_NPW8 : 2) : ((NKGT == 4 && WM == 1 && WN == 2) ? GTTS_WS64_NPW : NCW);   // producer waves (small form: a 32-channel
                                                     // consumer tile takes 4.3k cycles per chunk, one producer wave needs 6k to stage it)
    static constexpr int NT = (NCWP + NPW) * 64; // threads per workgroup
    // waves per SIMD the registers are budgeted for: 3 (168 registers) for the 12-wave 64-channel tile, and for the five-wave small
    // form of the f16 + fp8 split -- two of its workgroups per CU are ten waves; at 2 waves per SIMD (256 registers) only one fits
    static constexpr int WPS = (NT >= 768 || (NCW == 1 && NT > 256)) ? 3 : 2;
    static constexpr int MT = WM * MF * 32;      // output channels per workgroup
    static constexpr int TR = WN * NF;           // output rows per workgroup
    static constexpr int HR = TR + 2, HC = 34;   // halo tile
    static constexpr int NPIX = HR * HC;
    static constexpr int NKG = NKGT;             // 8-channel groups per chunk: 2 (16 channels), or 4 (32) for the f16 + fp8 split
    static_assert((NCW == 4 && MF == 2) || (NCW == 1 && MF == 1) || (NKGT == 4 && NCW == 2 && MF == 2 && WM == 1),
                  "four consumer waves of 64 channels, one of 32, or (f16 + fp8) the 64-channel tile");
};

// nsplit planes (1: hi; 2: hi + lo, or fp16 hi + fp8 cross-term operands) of nkg 8-channel groups per ring slot
// wq16: 16-byte units of the weight ring (the f16 + fp8 64-channel tile keeps two (chunk, column stage) weight blocks in LDS)
constexpr int WS64_WST16 = 3 * 64 * 2 * 4;      // one block: [split][tap = ky][kg / g][64 couts] x 16 B = 24 KB
static inline size_t ws_smem_bytes(int npix, int nsplit, int ring, int cin, int pro, int cout, int mf, int ncw, int nkg = 2, int wq16 = 0) {
    const size_t cpad = (size_t)((cin + 31) / 32) * 32;
    return (size_t)ring * nsplit * nkg * npix * 16 + (size_t)wq16 * 16 + (pro == PRO_GN ? (size_t)2 * 3 * cpad * 4 : 0) +
           (size_t)2 * ncw * mf * 8 * 4 + (size_t)((cout + 63) / 64) * 64 * 4;
}

// NSPLIT == 3: the f16 + fp8 split of GTTS_PREC_F16F8 (common.h) on 32-channel chunks: plane 0 of an image holds fp16 hi values
// [kg 0..3][pixel][8 channels], plane 1 the fp8 cross-term operands [g = plane * 2 + half][pixel][16 channels]; the consumers issue
// two fp16 k-steps and one fp8 K = 64 step per tap (2/3 of the bf16x3 MFMA cycles, 1.53x its sustained rate).
template <int WM, int WN, int MF, int NF, int PRO, int NSPLIT, typename AT, int RING>
__global__ __launch_bounds__((WsCfg<WM, WN, MF, NF, NSPLIT == 3 ? 4 : 2>::NT), (WsCfg<WM, WN, MF, NF, NSPLIT == 3 ? 4 : 2>::WPS))
void conv3x3_ws_kernel(const ConvArgs a) {
    constexpr bool F8 = NSPLIT == 3;
    using C = WsCfg<WM, WN, MF, NF, F8 ? 4 : 2>;
    constexpr int AB = (int)sizeof(AT);
    constexpr int MT = C::MT, TR = C::TR, HC = C::HC, NPIX = C::NPIX, NKG = C::NKG, NCW = C::NCW;
    constexpr int CH = 8 * NKG;                      // input channels per chunk (ring item)
    constexpr int NPL = F8 ? 2 : NSPLIT;             // planes per image
    constexpr int NCWP = C::NCWP;                    // consumer waves that run (== NCW except for the f16 + fp8 64-channel tile)
    constexpr int NCT = NCWP * 64, NPT = C::NPW * 64;  // consumer / producer threads
    constexpr int PLANE16 = NKG * NPIX;              // 16-byte units of one plane (hi or lo) of an image
    constexpr int IMG16 = NPL * PLANE16;             // ... of one ring slot
    static_assert(!F8 || (AB == 4 && RING == 2), "f16 + fp8 split: fp32 storage, two 32-channel images");
    constexpr int D = RING - 1;                      // producers run D chunks ahead
    // packed weight block (chunk, stage, cout tile of the PACKING: 128 channels for cout > 64): [split][tap][kg][MTP] x 16 B
    const int MTP = a.cout > 64 ? 128 : 64;
    const int WBLK16 = 3 * MTP * 2 * NKG;
    static_assert(PRO == PRO_MASK || PRO == PRO_GN, "Block prologues only");

    // W64 (GTTS_W64_RING builds only -- measured, not adopted: same box, us per level-0 64 -> 64 launch with the GroupNorm / mask
    // prologue: 266.4 / 258.5 for the register-load loop below against 274.6-275.4 / 259.8-262.5 for this form; ups.1.0.b1 189 vs 193):
    // the 12-wave 64-channel tile of the f16 + fp8 form.  Its weights do NOT come through the vector L1 tap by tap: the CU's L1
    // returns loads in order, and behind the producers' HBM-miss halo loads an L2-hit fragment load waits ~1000-1500 cycles at 32 KB in
    // flight, ~3000 at this tile's 52 KB (tools/probe/mem_probe.hip; round 5: chunk loop 10.25k cycles for 5.76k of MFMA issue, 7.05k
    // without weight reloads).  Instead every consumer wave streams its block's A fragments of one (chunk, COLUMN stage) -- the three taps
    // (ky = 0..2) of one kx -- into 12 KB of LDS of its own with LDS-DMA (buffer_load ... lds: no registers, a whole stage = 1920 MFMA
    // cycles ahead of their use) and reads them from there; see the consumer loop.  Column stages also let a B fragment serve three taps.
    constexpr bool W64 = GTTS_W64_RING && F8 && C::NCWP != C::NCW;
    constexpr int WQ16 = W64 ? 2 * WS64_WST16 : 0;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4 *s_img = reinterpret_cast<u32x4 *>(smem);                      // [RING][split][kg][NPIX]
    [[maybe_unused]] u32x4 *s_wq = s_img + RING * IMG16;                 // W64: [4 consumer waves][12 pieces][64 lanes] A fragments
    const int cpad = a.nchunk * CH;
    float *s_par = reinterpret_cast<float *>(s_img + RING * IMG16 + WQ16);   // PRO_GN: [2 (tile parity)][3][cpad] scale, shift, time bias
    float *s_red = s_par + (PRO == PRO_GN ? 2 * 3 * cpad : 0);           // [2 (tile parity)][NCW waves][MF][4 octets][2]
    float *s_epi = s_red + 2 * NCW * MF * 8;                             // [cout] bias of every output channel, written once (prologue)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kg_l = lane >> 5;

    // ---- persistent tile walk: workgroup position p in the XCD-banded order, tiles p, p + G, p + 2G, ...  (the 32 workgroups
    // of an XCD work on 32 consecutive tiles at any time: cout tiles of one pixel tile, then row neighbours -> shared halo
    // rows and the weights hit that XCD's L2)
    const int ncot = a.cout / MT;
    const int tps = a.tiles_x * a.tiles_y;              // pixel tiles per sample
    const int ntiles = a.B * tps * ncot;
    const int G = gridDim.x;
    const int pos = xcd_slot(blockIdx.x, G);
    const int my_tiles = pos < ntiles ? (ntiles - pos + G - 1) / G : 0;
    const int nchunk = a.nchunk;
    const int nitems = my_tiles * nchunk;
    const int HW = a.Hin * a.Win;                       // 3x3, pad 1: output geometry == input geometry
    struct TileId { int b, ty, tx, cot; };
    auto decode = [&](int k) {                          // k-th tile of this workgroup
        int t = pos + k * G;
        t = t < ntiles ? t : ntiles - 1;
        TileId r;
        r.cot = t % ncot; t /= ncot;
        r.ty = t % a.tiles_y; t /= a.tiles_y;
        r.tx = t % a.tiles_x;
        r.b = t / a.tiles_x;
        return r;
    };
    auto uniform_rsrc = [](const void *p, int bytes) {
        const unsigned long long u = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo), 0,
                                                 __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
    };

#if GTTS_WS_TRACE
    const bool tr_on = a.cin == GTTS_TRACE_CIN && a.cout == GTTS_TRACE_CIN && (blockIdx.x & 3) == 1;
    unsigned long long tr_sum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long tr_entry = WT_NOW();
#endif
    if (GTTS_WS_DEPHASE > 1 && G >= 8 * GTTS_WS_DEPHASE && my_tiles >= GTTS_WS_DEPHASE_MIN) {
        const int grp = (blockIdx.x >> 3) % GTTS_WS_DEPHASE;          // blockIdx % 8 is the XCD: every XCD has CUs in every group
        if (grp != 0) {
            const long long wait = (long long)grp * (MT * TR * 32 * AB / GTTS_WS_DEPHASE_BPC) * 2 / GTTS_WS_DEPHASE;
            const long long t0 = (long long)__builtin_amdgcn_s_memtime();
            while ((long long)__builtin_amdgcn_s_memtime() - t0 < wait) __builtin_amdgcn_s_sleep(16);
        }
    }
    if (wave < NCWP) {
        // =================================================================================== CONSUMERS
        __builtin_amdgcn_s_setprio(3);         // MFMA issue wins the per-SIMD arbitration against the producers' VALU stream
        const int wm = wave / WN, wn = wave % WN;
        const int m0 = wm * MF * 32;
        const int ncotp = a.cout / MTP, cpp = MTP / MT;             // packing tiles; workgroup cout tiles per packing tile
        const int wtotal = nchunk * 3 * ncotp * WBLK16 * 16;
        const __amdgpu_buffer_rsrc_t rsw = uniform_rsrc(a.w, wtotal);
        const int w_lane = (kg_l * MTP + m0 + l31) * 16;            // lane's row inside a (split, tap, kg) segment of the packing tile
        f32x16 acc[MF][NF];
        // Issue order of a tap (bf16x3): pass 1 wl.xh, pass 2 wh.xh, pass 3 wh.xl, each pass over all MF x NF accumulators
        // (two MFMAs on one accumulator are MF*NF issue slots apart).  Everything a pass multiplies was requested at least
        // one pass (10 MFMAs, 320 cycles) earlier: xl of this tap at the start of pass 1, xh of the NEXT tap at the start
        // of pass 3 (xh is dead after pass 2), the next tap's wl at the start of pass 2 (wl is dead after pass 1) and
        // its wh -- into the second wh register set -- at the start of pass 1.  sched_barrier(0) between the passes pins
        // that order (hipcc otherwise hoists the loads of all nine taps and spills); the waits it inserts are exact.
        bf16x8 wh[MF], wl[MF], xh[NF];
        TileId tl = decode(0);
        // B-fragment base of this lane inside a plane (16-byte units): rows wn*NF.., column l31
        const int x_lane = kg_l * NPIX + wn * NF * HC + l31;
        auto wload_one = [&](bf16x8 (&w)[MF], int mi, int sp, int chunk, int stage, int tap, int cot) {
            if (GTTS_WS_EXP == 1 && (chunk | stage | tap) != 0) return;
            const int blk = (chunk * 3 + stage) * ncotp + cot / cpp;
            const int w_voff = w_lane + (cot % cpp) * MT * 16;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(
                rsw, w_voff + mi * 32 * 16, blk * (WBLK16 * 16) + ((sp * 3 + tap) * NKG * MTP) * 16, 0);
            w[mi] = __builtin_bit_cast(bf16x8, v);
        };
        auto wload1 = [&](bf16x8 (&w)[MF], int sp, int chunk, int stage, int tap, int cot) {
            if (GTTS_WS_EXP == 1 && (chunk | stage | tap) != 0) return;
            const int blk = (chunk * 3 + stage) * ncotp + cot / cpp;
            const int w_voff = w_lane + (cot % cpp) * MT * 16;
#pragma unroll
            for (int mi = 0; mi < MF; ++mi) {
                const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(
                    rsw, w_voff + mi * 32 * 16, blk * (WBLK16 * 16) + ((sp * 3 + tap) * NKG * MTP) * 16, 0);
                w[mi] = __builtin_bit_cast(bf16x8, v);
            }
        };
        // ------------------------------------------------------------ tile epilogue: bias, store, GroupNorm partial sums
        auto ws_epilogue = [&](int par) {
                const int y0 = tl.ty * TR + wn * NF, ox = tl.tx * 32 + l31;
                const int out_bytes = a.cout * HW * AB;
                const __amdgpu_buffer_rsrc_t rs_out =
                    uniform_rsrc(reinterpret_cast<AT *>(a.out) + (size_t)tl.b * a.cout * HW, out_bytes);
                const int ch0 = tl.cot * MT + m0;
                const bool col_ok = ox < a.Wout;
                float st1[MF][4], st2[MF][4];
#pragma unroll
                for (int mi = 0; mi < MF; ++mi)
#pragma unroll
                    for (int q = 0; q < 4; ++q) { st1[mi][q] = 0.f; st2[mi][q] = 0.f; }
                const float *bias_l = s_epi + tl.cot * MT + m0 + 4 * kg_l;
                // (Measured and dropped: a 4 x 4 transpose inside every lane quad -- DPP exchanges + bit-selects -- so that a lane
                // holds four consecutive frames of one channel and stores 16 bytes.  A quarter of the store instructions, but
                // each then touches 8 channel rows instead of 2: epilogue 12.9k vs 9.9k cycles per 128-channel tile.)
#pragma unroll
                for (int mi = 0; mi < MF; ++mi) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float bv[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) bv[i] = bias_l[mi * 32 + 8 * q + i];
                        const int soff = (ch0 + mi * 32 + 8 * q) * HW * AB;
#pragma unroll
                        for (int ni = 0; ni < NF; ++ni) {
                            const int oy = y0 + ni;
                            if (oy >= a.Hout) continue;
                            float r[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                if constexpr (F8) r[i] = fmaf(acc[mi][ni][4 * q + i], 1.0f / (float)(1 << F8_S), bv[i]);   // accumulators hold 2^S x the sum (common.h)
                                else r[i] = acc[mi][ni][4 * q + i] + bv[i];
                            }
                            if (col_ok) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    st1[mi][q] += r[i];
                                    st2[mi][q] = fmaf(r[i], r[i], st2[mi][q]);
                                }
                                const int voff = (oy * a.Wout + ox + 4 * kg_l * HW) * AB;
#pragma unroll
                                for (int i = 0; i < 4; ++i) st_act<AT>(r[i], rs_out, voff, soff + i * HW * AB);
                            }
                        }
                    }
                }
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
                    for (int kk = 0; kk < V / 4; ++kk) {
                        const int vi = kk + (V / 4) * (r & 1) + (V / 2) * (r >> 1);
                        const int which = vi / (4 * MF), mi = (vi % (4 * MF)) >> 2, q = vi & 3;
                        s_red[par * (NCW * MF * 8) + ((wave * MF + mi) * 4 + q) * 2 + which] = tot[kk];
                    }
                }
        };
        // the layer's whole bias vector, once: a load per tile (as before round 6) put an s_waitcnt vmcnt(0) behind the previous tile's
        // epilogue stores at every tile start -- the consumers then waited for up to 63 stores per wave to be acknowledged before their
        // first MFMA of the next tile instead of leaving them to drain behind it
        for (int c = tid; c < a.cout; c += NCT) s_epi[c] = a.bias[c];
        lds_barrier();                                              // (P) prologue barrier: s_par of tile 0 is written
        int k = 0, cc = 0, slot = 0;
        if constexpr (F8) {
            // ---------------------------------------------------------------- f16 + fp8 consumer loop (32 channels per item)
            // Wave mapping of this form: a wave owns ONE 32-channel block of the tile's output channels for FR rows -- all 10 rows of a
            // 128-channel tile (four blocks, four waves), one 5-row band of a 64-channel tile (two blocks x two bands), the 5 rows of the
            // small form -- i.e. FR accumulators and HALF the weight fragments per tap of the bf16x3 form's 64-channel x 5-row mapping
            // (16 registers: two fp16 k-steps + one fp8 operand).  The weights are therefore kept NWS sets deep: the set of tap t + NWS - 1
            // is requested at the start of tap t, >= 1280 cycles ahead of its first use.  Measured on the first version of this loop
            // (64 x 5 mapping, 8 fragment loads per tap 640-1100 cycles ahead): 16.1k cycles per item for 11.5k of MFMA issue, 12.6k
            // without weight reloads, 12.3k with the loads and NO MFMAs -- an L2 round trip under this load is ~1400 cycles
            // (profiles/r05_ws_f16f8_ablations.txt); with the sets below: 12.5k.  Activation fragments are each used by one MFMA only; they
            // stream through a rolling window LEAD cycles ahead (sched_barrier pins the issue order; hipcc reuses the registers of dead
            // fragments).  A tap is three passes over the FR accumulators: fp16 k-step 0, fp16 k-step 1, fp8 (both cross terms) -- per
            // accumulator always in this order, in every form of the kernel.
            constexpr int CB = MT / 32;                              // 32-channel blocks of the tile
            constexpr int NBW = NCWP / CB;                           // row bands that different waves own (2: the 64-channel tile)
            constexpr int FR = TR / NBW;                             // rows (accumulators) per wave
            constexpr int NS = 3 * FR;                               // MFMA slots per tap
            constexpr int NWS = FR >= 10 ? 2 : GTTS_WS64_NWS;        // weight sets in flight (a tap is 128 FR cycles)
            constexpr int LEAD = C::WPS >= 3 ? GTTS_WS64_LEAD : GTTS_WS_LEAD;   // cycles between a fragment's ds_read and its MFMA (three waves per SIMD: 168 registers)
            const int cbk = wave % CB, bnd = wave / CB;              // wave -> (channel block, band)
            const int fm0 = cbk * 32;
            f32x16 facc[FR];
            const int wl_h = (kg_l * MTP + fm0 + l31) * 16;         // lane's row in a (tap, kg) segment of the fp16 plane
            const int wl_8 = (kg_l * 2 * MTP + fm0 + l31) * 16;     // ... of the fp8 plane: g = kg_l * 2 + q
            struct WSet { f16x8 a, b; i32x8 w8; };
            auto wload = [&](WSet &w, int chunk, int stage, int tap, int cot) {
                const int blk = (chunk * 3 + stage) * ncotp + cot / cpp;
                const int cofs = (cot % cpp) * MT * 16;
                const int so = blk * (WBLK16 * 16);
                const u32x4 va = __builtin_amdgcn_raw_buffer_load_b128(rsw, wl_h + cofs, so + ((tap * NKG) * MTP) * 16, 0);
                const u32x4 vb = __builtin_amdgcn_raw_buffer_load_b128(rsw, wl_h + cofs, so + ((tap * NKG + 2) * MTP) * 16, 0);
                const u32x4 q0 = __builtin_amdgcn_raw_buffer_load_b128(rsw, wl_8 + cofs, so + (((3 + tap) * NKG) * MTP) * 16, 0);
                const u32x4 q1 = __builtin_amdgcn_raw_buffer_load_b128(rsw, wl_8 + cofs, so + (((3 + tap) * NKG + 1) * MTP) * 16, 0);
                w.a = __builtin_bit_cast(f16x8, va);
                w.b = __builtin_bit_cast(f16x8, vb);
                w.w8[0] = (int)q0[0]; w.w8[1] = (int)q0[1]; w.w8[2] = (int)q0[2]; w.w8[3] = (int)q0[3];
                w.w8[4] = (int)q1[0]; w.w8[5] = (int)q1[1]; w.w8[6] = (int)q1[2]; w.w8[7] = (int)q1[3];
            };
            // start cycle of MFMA slot s of a tap (slots [0, FR): k-step 0, [FR, 2 FR): k-step 1, 32 cycles each; [2 FR, 3 FR): fp8, 64 each)
            auto slot_t = [](int s2) { return s2 < 2 * FR ? 32 * s2 : 64 * FR + 64 * (s2 - 2 * FR); };
            constexpr int TAPC = 128 * FR;                           // cycles per tap
            // slot of THIS tap at whose start the fragment of slot u (u >= NS: slot u - NS of the next tap) is requested: the last slot
            // starting at least LEAD cycles before u does; -1: before this tap began (requested at the item's start instead)
            auto issue_slot = [&](int u) {
                const int tu = u < NS ? slot_t(u) : TAPC + slot_t(u - NS);
                int r = -1;
                for (int q = 0; q < NS; ++q)
                    if (slot_t(q) + LEAD <= tu) r = q;
                return r;
            };
            WSet wq[NWS];                                            // wq[0]: the current tap's set; wq[k]: tap + k
            const int xl0 = kg_l * NPIX + bnd * FR * HC + l31;       // lane's fragment column in a plane; rows start at this wave's band
            // tile epilogue of this mapping: bias, store, GroupNorm partial sums per 5-row band, written to s_red in the layout of the
            // bf16x3 form (wave row = (channel-block pair) * WN + band, fragment = block inside the pair), so that finish_tile is common
            auto f8_epilogue = [&](int par) {
                const int oxx = tl.tx * 32 + l31;
                const int out_bytes = a.cout * HW * AB;
                const __amdgpu_buffer_rsrc_t rs_out =
                    uniform_rsrc(reinterpret_cast<AT *>(a.out) + (size_t)tl.b * a.cout * HW, out_bytes);
                const int ch0 = tl.cot * MT + fm0;
                const bool col_ok = oxx < a.Wout;
                const float *bias_l = s_epi + tl.cot * MT + fm0 + 4 * kg_l;
#pragma unroll
                for (int lb = 0; lb < FR / 5; ++lb) {
                    const int band = bnd * (FR / 5) + lb;             // 5-row band of the tile
                    float st1[4], st2[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { st1[q] = 0.f; st2[q] = 0.f; }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        float bv[4];
#pragma unroll
                        for (int i = 0; i < 4; ++i) bv[i] = bias_l[8 * q + i];
                        const int soff = (ch0 + 8 * q) * HW * AB;
#pragma unroll
                        for (int rr = 0; rr < 5; ++rr) {
                            const int r = lb * 5 + rr;
                            const int oy = tl.ty * TR + band * 5 + rr;
                            if (oy >= a.Hout) continue;
                            float v[4];
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = fmaf(facc[r][4 * q + i], 1.0f / (float)(1 << F8_S), bv[i]);   // accumulators hold 2^S x the sum
                            if (col_ok) {
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    st1[q] += v[i];
                                    st2[q] = fmaf(v[i], v[i], st2[q]);
                                }
                                const int voff = (oy * a.Wout + oxx + 4 * kg_l * HW) * AB;
#pragma unroll
                                for (int i = 0; i < 4; ++i) {
                                    if (GTTS_WS_EXP == 6 && v[i] != 12345.678f) continue;      // (ablation: no output stores)
                                    if constexpr (AB == 4 && GTTS_WS_EPI_AUX != 0) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v[i]), rs_out, voff, soff + i * HW * AB, GTTS_WS_EPI_AUX);
                                    else st_act<AT>(v[i], rs_out, voff, soff + i * HW * AB);
                                }
                            }
                        }
                    }
                    float vals[8], tot[2];
#pragma unroll
                    for (int q = 0; q < 4; ++q) { vals[q] = st1[q]; vals[4 + q] = st2[q]; }
                    wave_sums_transposed<8>(vals, tot);
                    if ((lane & 15) == 0) {
                        const int rw = lane >> 4;
                        const int wold = (cbk / MF) * WN + band, mi = cbk % MF;
#pragma unroll
                        for (int kk = 0; kk < 2; ++kk) {
                            const int vi = kk + 2 * (rw & 1) + 4 * (rw >> 1);
                            const int which = vi >> 2, q = vi & 3;
                            s_red[par * (NCW * MF * 8) + ((wold * MF + mi) * 4 + q) * 2 + which] = tot[kk];
                        }
                    }
                }
            };
            if constexpr (W64) {
                // ------------------------------------------------------ 64-channel tile: weights from a private LDS ring, column stages
                // An item is three COLUMN stages j = kx = 0..2 of two sweeps each; one workgroup barrier per item (the image hand-over),
                // nothing else between waves.  Every consumer wave keeps ITS 32-channel block's weights of one (chunk, kx) -- twelve
                // A fragments of 1 KB: (ky, k-step) x 6 for the fp16 plane, (ky, half) x 6 for the fp8 plane -- in 12 KB of LDS of its own and
                // refills each half BEHIND its own reads: once the six fp16 fragments of stage s are in registers (their first MFMAs
                // have issued) the wave requests the fp16 fragments of stage s + 1 into the same 6 KB by LDS-DMA, likewise the fp8 half in
                // the fp8 sweep -- a whole stage (1920 MFMA cycles) ahead of their use, no registers, no other wave involved (the two
                // bands of a block each fetch their copy: L2 hits).  s_waitcnt vmcnt(6) before a sweep's fragment reads = "everything but
                // the six pieces requested last has landed" (loads return in order).  At a tile's end the two pending refills are
                // drained with vmcnt(0) BEFORE the epilogue's 80 stores enter the queue, and the first stage of the next tile skips its
                // two waits (a counted wait would otherwise sit behind those stores).
                // (History, same box: weights as per-tap register loads through the L1 303 / 286 us per level-0 launch (GroupNorm / mask
                // prologue); one shared two-block ring with all twelve waves in three barriers per item 303 / 286 -- consumers and producers
                // then wait for each other's slowest third; the shared ring with a consumer-only arrival counter 279 / 260.)
                // Sweeps: input rows R = 0..6 of the wave's band, one B fragment (pair) per row fetched two rows ahead and used by every
                // (output row r = R - ky, ky) pair: 84 + 36 fragment reads per item and wave instead of 180 + 36 weight loads.
                // Order per accumulator (every band, every batch size -- the small-launch form does not take 64-channel layers in
                // f16 + fp8): chunk, kx; then ky 0..2: k-step 0, k-step 1; then ky 0..2: fp8.
                typedef __attribute__((address_space(3))) void *lds_vp;
                u32x4 *wq = s_wq + wave * (WS64_WST16 / 2);            // this wave's 12 KB: pieces 0..5 fp16 (ky * 2 + k-step), 6..11 fp8 (ky * 2 + half)
                const int src16 = (kg_l * 64 + fm0 + l31) * 16;       // lane's byte offset inside a (tap, kg pair) segment of the packed fp16 plane ...
                const int src8 = ((3 * NKG + kg_l * 2) * 64 + fm0 + l31) * 16;   // ... and of the fp8 plane (g = kg_l * 2 + half)
                auto dma16 = [&](int chunk, int stage) {              // fp16 A fragments of (chunk, kx = stage): piece = ky * 2 + k-step
                    const int so = (chunk * 3 + stage) * (WS64_WST16 * 16);
#pragma unroll
                    for (int p = 0; p < 6; ++p)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_vp)(wq + p * 64), 16, src16, so + (((p >> 1) * NKG + (p & 1) * 2) * 64) * 16, 0, 0);
                };
                auto dma8 = [&](int chunk, int stage) {               // fp8 A fragments: piece = 6 + ky * 2 + half
                    const int so = (chunk * 3 + stage) * (WS64_WST16 * 16);
#pragma unroll
                    for (int p = 0; p < 6; ++p)
                        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_vp)(wq + (6 + p) * 64), 16, src8, so + (((p >> 1) * NKG + (p & 1)) * 64) * 16, 0, 0);
                };
                // The six sweeps of an item are software-pipelined: while a sweep works on its last input rows (R == 5) it reads the NEXT
                // sweep's six A fragments and first two B rows (behind a vmcnt(6) for the refill they come from), so a sweep starts with
                // its operands in registers -- without that every sweep opened on an exposed LDS round trip (A + two B rows, ~300-400
                // cycles, six times per item: level-0 launches 273 / 267 us).  Only the first sweep of an item reads its two B rows after the
                // item barrier (the image is not complete before it).
                u32x4 Acur[6], Anext[6], Bn[GTTS_W64_DIST][2];
                auto readA = [&](u32x4 (&A)[6], int piece0) {
#pragma unroll
                    for (int p = 0; p < 6; ++p) A[p] = wq[(piece0 + p) * 64 + lane];
                };
                if (nitems > 0) {
                    dma16(0, 0);
                    dma8(0, 0);
                    if (GTTS_W64_PF != 0) {
                        asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                        readA(Acur, 0);
                    }
                    if (GTTS_WS_EXP == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (ablation: the ring is filled once)
                }
                for (int i = 0; i < nitems; ++i) {
                    const int par = k & 1;
                    const u32x4 *xh_p = s_img + slot * IMG16 + xl0;                                  // fp16 plane, k-step 0 (k-step 1: + 2 NPIX)
                    const u32x4 *x8_p = s_img + slot * IMG16 + PLANE16 + xl0 + kg_l * NPIX;          // fp8 plane: g = 2 kg_l (second half: + NPIX)
                    slot = slot + 1 == RING ? 0 : slot + 1;
                    const bool last_c = cc + 1 == nchunk;
                    [[maybe_unused]] const unsigned long long tw0 = WT_NOW();
                    lds_barrier();                  // image of item i is complete; everybody is done with item i - 1
                    [[maybe_unused]] const unsigned long long tw1 = WT_NOW();
                    WT_ADD(0, tw1, tw0);
                    auto readB = [&](u32x4 (&b)[2], int R, int j, bool f8s) {      // the B fragment (pair) of input row R at column shift j
                        if (!f8s) { b[0] = xh_p[R * HC + j]; b[1] = xh_p[2 * NPIX + R * HC + j]; }
                        else { b[0] = x8_p[R * HC + j]; b[1] = x8_p[NPIX + R * HC + j]; }
                    };
                    if (GTTS_W64_PF == 1) {
#pragma unroll
                        for (int d = 0; d < GTTS_W64_DIST; ++d) readB(Bn[d], d, 0, false);
                    }
                    if (cc == 0) {
#pragma unroll
                        for (int r = 0; r < FR; ++r)
#pragma unroll
                            for (int e = 0; e < 16; ++e) facc[r][e] = 0.f;
                    }
                    const bool fresh = i != 0 && cc == 0;             // first stage after an epilogue: both ring halves landed before it
#pragma unroll
                    for (int s = 0; s < 6; ++s) {
                        const int j = s >> 1;
                        const bool f8s = (s & 1) != 0;
                        const int nch = j == 2 ? (last_c ? 0 : cc + 1) : cc, nst = j == 2 ? 0 : j + 1;      // the next stage (after the last item: block 0 again, never read)
                        u32x4 Bw[FR + 2][2];
                        if (GTTS_W64_PF == 0) {
                            [[maybe_unused]] const unsigned long long tv0 = WT_NOW();
                            if (GTTS_WS_EXP != 1 && !(fresh && j == 0)) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                            WT_ADD(7, WT_NOW(), tv0);
                            readA(Acur, f8s ? 6 : 0);
                        }
                        if (GTTS_W64_PF == 1) {
#pragma unroll
                            for (int d = 0; d < GTTS_W64_DIST; ++d) { Bw[d][0] = Bn[d][0]; Bw[d][1] = Bn[d][1]; }
                        } else {
#pragma unroll
                            for (int d = 0; d < GTTS_W64_DIST; ++d) readB(Bw[d], d, j, f8s);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int R = 0; R < FR + 2; ++R) {
                            if (R + GTTS_W64_DIST < FR + 2) readB(Bw[R + GTTS_W64_DIST], R + GTTS_W64_DIST, j, f8s);
                            if (GTTS_W64_PF != 0 && R == GTTS_W64_PFROW) {
                                // operands of the next sweep: its A fragments come from the ring half refilled one stage ago
                                [[maybe_unused]] const unsigned long long tv0 = WT_NOW();
                                if (GTTS_WS_EXP != 1 && !(fresh && s == 0)) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                                WT_ADD(7, WT_NOW(), tv0);
                                readA(Anext, f8s ? 0 : 6);
                                if (GTTS_W64_PF == 1 && s < 5) {
#pragma unroll
                                    for (int d = 0; d < GTTS_W64_DIST; ++d) readB(Bn[d], d, (s + 1) >> 1, !f8s);
                                }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            if (!f8s) {
#pragma unroll
                                for (int st = 0; st < 3; ++st)
                                    if (R - st >= 0 && R - st < FR)
                                        facc[R - st] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, Acur[st * 2]), __builtin_bit_cast(f16x8, Bw[R][0]), facc[R - st], 0, 0, 0);
#pragma unroll
                                for (int st = 0; st < 3; ++st)
                                    if (R - st >= 0 && R - st < FR)
                                        facc[R - st] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, Acur[st * 2 + 1]), __builtin_bit_cast(f16x8, Bw[R][1]), facc[R - st], 0, 0, 0);
                            } else {
                                i32x8 b8;
                                b8[0] = (int)Bw[R][0][0]; b8[1] = (int)Bw[R][0][1]; b8[2] = (int)Bw[R][0][2]; b8[3] = (int)Bw[R][0][3];
                                b8[4] = (int)Bw[R][1][0]; b8[5] = (int)Bw[R][1][1]; b8[6] = (int)Bw[R][1][2]; b8[7] = (int)Bw[R][1][3];
#pragma unroll
                                for (int st = 0; st < 3; ++st)
                                    if (R - st >= 0 && R - st < FR) {
                                        i32x8 a8;
                                        a8[0] = (int)Acur[st * 2][0]; a8[1] = (int)Acur[st * 2][1]; a8[2] = (int)Acur[st * 2][2]; a8[3] = (int)Acur[st * 2][3];
                                        a8[4] = (int)Acur[st * 2 + 1][0]; a8[5] = (int)Acur[st * 2 + 1][1]; a8[6] = (int)Acur[st * 2 + 1][2]; a8[7] = (int)Acur[st * 2 + 1][3];
                                        facc[R - st] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, facc[R - st], 0, 0, 0, 0, 0, 0);
                                    }
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            // every A fragment of this sweep has been consumed by an issued MFMA (ky = 2 first at R = 2): its ring half may be refilled
                            if (R == GTTS_W64_DMAROW && GTTS_WS_EXP != 1) { if (!f8s) dma16(nch, nst); else dma8(nch, nst); }
                        }
                        if (GTTS_W64_PF != 0) {
#pragma unroll
                            for (int p = 0; p < 6; ++p) Acur[p] = Anext[p];
                        }
                    }
                    [[maybe_unused]] const unsigned long long tw2 = WT_NOW();
                    WT_ADD(1, tw2, tw1);
                    WT_ADD(3, 1ull, 0ull);
                    if (!last_c) { ++cc; continue; }
#if GTTS_WS_TRACE
                    if (tr_on) {      // ([7]: cycles in the sweeps' vmcnt waits)
                        if (k == 0) tr_sum[5] = tw2 - tr_entry;
                        tr_sum[6] = tw2 - tr_entry;
                    }
#endif
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the next tile's first two refills land before the stores queue up behind them
                    WT_ADD(7, WT_NOW(), tw2);
                    f8_epilogue(par);
                    cc = 0;
                    ++k;
                    tl = decode(k);
                    WT_ADD(2, WT_NOW(), tw2);
                }
            } else
            for (int i = 0; i < nitems; ++i) {
                [[maybe_unused]] const unsigned long long tw0 = WT_NOW();
                lds_barrier();                                      // image of item i is complete; everybody is done with item i - 1
                [[maybe_unused]] const unsigned long long tw1 = WT_NOW();
                WT_ADD(0, tw1, tw0);
                const int par = k & 1;
                const u32x4 *xh_p = s_img + slot * IMG16 + xl0;                                  // fp16 plane, k-step 0 (k-step 1: + 2 NPIX)
                const u32x4 *x8_p = s_img + slot * IMG16 + PLANE16 + xl0 + kg_l * NPIX;          // fp8 plane: g = 2 kg_l (second half: + NPIX)
                slot = slot + 1 == RING ? 0 : slot + 1;
                const bool last_c = cc + 1 == nchunk;
                const int ncn = last_c ? cc : cc + 1;               // (the last chunk re-requests its own first taps: never used)
                // weights of tap index u of this item (u >= 9: tap u - 9 of the next chunk)
                auto wload_tap = [&](WSet &w, int u) {
                    if (u < 9) wload(w, cc, u / 3, u % 3, tl.cot);
                    else wload(w, ncn, (u - 9) / 3, (u - 9) % 3, tl.cot);
                };
                // (ablation 7: taps 0..2 of every tile but the first run on stale weight registers -- what a tile start that issues no
                // vector load behind the previous tile's stores for three taps would gain)
                const bool abl7 = GTTS_WS_EXP == 7 && i != 0 && cc == 0;
                if (cc == 0) {
                    // a tile starts cold: its first weight sets are requested here (one exposed round trip per tile)
                    if (!abl7) {
#pragma unroll
                    for (int q = 0; q + 1 < NWS; ++q) wload_tap(wq[q], q);
                    }
#pragma unroll
                    for (int r = 0; r < FR; ++r)
#pragma unroll
                        for (int e = 0; e < 16; ++e) facc[r][e] = 0.f;
                }
                // activation fragments of the tap in flight: one variable per slot (dead ones are reused by the register allocator)
                f16x8 fa[FR], fb[FR];
                u32x4 f8l[FR], f8h[FR];
                auto fetch = [&](int u, int st, int j) {             // fragment of slot u (< NS) of tap (st, j)
                    const int r = u % FR, off = (r + st) * HC + j;
                    if (u < FR) fa[r] = *reinterpret_cast<const f16x8 *>(xh_p + off);
                    else if (u < 2 * FR) fb[r] = *reinterpret_cast<const f16x8 *>(xh_p + 2 * NPIX + off);
                    else { f8l[r] = x8_p[off]; f8h[r] = x8_p[NPIX + off]; }
                };
                // fragments of tap (0, 0) whose request slot lies before the tap
#pragma unroll
                for (int u = 0; u < NS; ++u)
                    if (issue_slot(u) < 0) fetch(u, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int st = 0; st < 3; ++st) {
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        const bool last_t = st == 2 && j == 2;
                        const int nst = j == 2 ? st + 1 : st, nj = j == 2 ? 0 : j + 1;      // next tap inside the chunk
                        // the set of tap + NWS - 1, NWS - 1 taps ahead
                        if (GTTS_WS_EXP != 1 && !(abl7 && st == 0 && j < 2)) wload_tap(wq[NWS - 1], st * 3 + j + NWS - 1);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int s2 = 0; s2 < NS; ++s2) {
                            // requests due at this slot: later slots of this tap, then the first slots of the next tap (same image)
#pragma unroll
                            for (int u = s2 + 1; u < 2 * NS; ++u) {
                                if (issue_slot(u) != s2) continue;
                                if (u < NS) fetch(u, st, j);
                                else if (!last_t && issue_slot(u - NS) < 0) fetch(u - NS, nst, nj);     // (the others are requested inside their own tap)
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            const int r = s2 % FR;
#if GTTS_WS_EXP == 5 && defined(__HIP_DEVICE_COMPILE__)
                            // (diagnostic builds, device pass only: on the host pass an asm with a "v" operand silently drops the kernel's stub)
                            if (s2 < FR) asm volatile("" ::"v"(wq[0].a), "v"(fa[r]));
                            else if (s2 < 2 * FR) asm volatile("" ::"v"(wq[0].b), "v"(fb[r]));
                            else asm volatile("" ::"v"(wq[0].w8), "v"(f8l[r]), "v"(f8h[r]));
                            if (false)
#endif
                            if (s2 < FR) {
                                facc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[0].a, fa[r], facc[r], 0, 0, 0);
                            } else if (s2 < 2 * FR) {
                                facc[r] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wq[0].b, fb[r], facc[r], 0, 0, 0);
                            } else {
                                i32x8 b8;
                                b8[0] = (int)f8l[r][0]; b8[1] = (int)f8l[r][1]; b8[2] = (int)f8l[r][2]; b8[3] = (int)f8l[r][3];
                                b8[4] = (int)f8h[r][0]; b8[5] = (int)f8h[r][1]; b8[6] = (int)f8h[r][2]; b8[7] = (int)f8h[r][3];
                                facc[r] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wq[0].w8, b8, facc[r], 0, 0, 0, 0, 0, 0);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        // rotate the sets (renames in the unrolled code; copies only on the loop's back edge)
#pragma unroll
                        for (int q = 0; q + 1 < NWS; ++q) wq[q] = wq[q + 1];
                    }
                }
                [[maybe_unused]] const unsigned long long tw2 = WT_NOW();
                WT_ADD(1, tw2, tw1);
                WT_ADD(3, 1ull, 0ull);
                if (!last_c) { ++cc; continue; }
#if GTTS_WS_TRACE
                if (tr_on) {      // [5] start of the first, [6] of the last tile epilogue since kernel entry, [7] phase group
                    if (k == 0) tr_sum[5] = tw2 - tr_entry;
                    tr_sum[6] = tw2 - tr_entry;
                    tr_sum[7] = (blockIdx.x >> 3) % (GTTS_WS_DEPHASE > 1 ? GTTS_WS_DEPHASE : 1);
                }
#endif
                f8_epilogue(par);
                cc = 0;
                ++k;
                tl = decode(k);
                WT_ADD(2, WT_NOW(), tw2);
            }
        } else
        for (int i = 0; i < nitems; ++i) {
            [[maybe_unused]] const unsigned long long tw0 = WT_NOW();
            lds_barrier();                                          // image of item i is complete; everybody is done with item i - 1
            [[maybe_unused]] const unsigned long long tw1 = WT_NOW();
            WT_ADD(0, tw1, tw0);
            const int par = k & 1;
            const u32x4 *xh_p = s_img + slot * IMG16 + x_lane;
            const u32x4 *xl_p = xh_p + PLANE16;
            slot = slot + 1 == RING ? 0 : slot + 1;
            const u32x4 *xn_p = s_img + slot * IMG16 + x_lane;     // next item's image (RING == 3: complete since the last barrier)
            if (cc == 0) {
                // a tile starts cold: first fragments requested here (one exposed round trip per tile)
                wload1(wh, 0, 0, 0, 0, tl.cot);
                if (NSPLIT > 1) wload1(wl, 1, 0, 0, 0, tl.cot);
#pragma unroll
                for (int ni = 0; ni < NF; ++ni) xh[ni] = *reinterpret_cast<const bf16x8 *>(xh_p + ni * HC);
#pragma unroll
                for (int mi = 0; mi < MF; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NF; ++ni)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
            } else if (RING < 3) {
#pragma unroll
                for (int ni = 0; ni < NF; ++ni) xh[ni] = *reinterpret_cast<const bf16x8 *>(xh_p + ni * HC);
            }
            const bool last_c = cc + 1 == nchunk;
            const int ncn = last_c ? cc : cc + 1;                   // (the last chunk re-requests its own first tap: never used)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int st = 0; st < 3; ++st) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const bool last_t = st == 2 && j == 2;
                    const int nst = j == 2 ? st + 1 : st, nj = j == 2 ? 0 : j + 1;      // next tap inside the chunk
                    bf16x8 whn[MF], xl[NF], xhn[NF];
                    if (GTTS_WS_EXP == 1) {
#pragma unroll
                        for (int mi = 0; mi < MF; ++mi) whn[mi] = wh[mi];
                    }
                    // ---- pass 1
                    if (last_t) wload1(whn, 0, ncn, 0, 0, tl.cot); else wload1(whn, 0, cc, nst, nj, tl.cot);
                    if (NSPLIT > 1) {
#pragma unroll
                        for (int ni = 0; ni < NF; ++ni) {
                            if (GTTS_WS_EXP == 4) xl[ni] = xh[ni];
                            else xl[ni] = *reinterpret_cast<const bf16x8 *>(xl_p + (ni + st) * HC + j);
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int mi = 0; mi < MF; ++mi) {
#pragma unroll
                            for (int ni = 0; ni < NF; ++ni)
                                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[mi], xh[ni], acc[mi][ni], 0, 0, 0);
                            __builtin_amdgcn_sched_barrier(0);
                            // wl[mi] is dead: request the next tap's right away (25 MFMAs ahead of its first use)
                            if (last_t) wload_one(wl, mi, 1, ncn, 0, 0, tl.cot); else wload_one(wl, mi, 1, cc, nst, nj, tl.cot);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                        // ---- pass 2
                    }
#pragma unroll
                    for (int mi = 0; mi < MF; ++mi)
#pragma unroll
                        for (int ni = 0; ni < NF; ++ni)
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[mi], xh[ni], acc[mi][ni], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    // ---- pass 3 (single-pass bf16: nothing left but the prefetch)
                    if (GTTS_WS_EXP == 4) {
#pragma unroll
                        for (int ni = 0; ni < NF; ++ni) xhn[ni] = xh[ni];
                    } else if (last_t) {
                        if (RING >= 3) {
#pragma unroll
                            for (int ni = 0; ni < NF; ++ni) xhn[ni] = *reinterpret_cast<const bf16x8 *>(xn_p + ni * HC);
                        }
                    } else {
#pragma unroll
                        for (int ni = 0; ni < NF; ++ni) xhn[ni] = *reinterpret_cast<const bf16x8 *>(xh_p + (ni + nst) * HC + nj);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (NSPLIT > 1) {
#pragma unroll
                        for (int mi = 0; mi < MF; ++mi)
#pragma unroll
                            for (int ni = 0; ni < NF; ++ni)
                                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[mi], xl[ni], acc[mi][ni], 0, 0, 0);
                    }
#pragma unroll
                    for (int mi = 0; mi < MF; ++mi) wh[mi] = whn[mi];
                    if (!last_t || RING >= 3) {
#pragma unroll
                        for (int ni = 0; ni < NF; ++ni) xh[ni] = xhn[ni];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            [[maybe_unused]] const unsigned long long tw2 = WT_NOW();
            WT_ADD(1, tw2, tw1);
            WT_ADD(3, 1ull, 0ull);
            if (!last_c) { ++cc; continue; }

            ws_epilogue(par);
            cc = 0;
            ++k;
            tl = decode(k);
            WT_ADD(2, WT_NOW(), tw2);
        }
        lds_barrier();                                              // (F) the last tile's wave sums are in s_red
    } else {
        // =================================================================================== PRODUCERS
        const int ptid = tid - NCT;
        // A staging item is (8-channel group, halo row, GROUP OF FOUR consecutive frames): eight 16-byte loads (one per
        // channel; 8 bytes in bf16 storage) instead of thirty-two dword loads -- the texture path spends its address cycles
        // per instruction, and with dword loads the producers' requests alone kept it busy for a third of a chunk, in front of
        // the consumers' weight-fragment loads.  Nine groups cover the 34 halo frames (the last one half); a group that
        // straddles the image edge reads the neighbouring row's frames (valid memory) and the mask factor zeroes them.
        constexpr int NG = 9;                         // frame groups per halo row
        constexpr int NLI = NKG * C::HR * NG;         // lane-items per chunk
        constexpr int LITER = (NLI + NPT - 1) / NPT;
        typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
        using RawV = typename std::conditional<AB == 4, u32x4, u32x2>::type;
        struct Raw { RawV x[LITER][8]; };
        Raw rawA, rawB;           // two chunks in flight: HBM latency is covered by a whole chunk of the consumers' MFMAs
        int it_goff[LITER];       // load side: byte offset of (first channel of the 8-group, row, first frame of the group) in a chunk
        float m_nxt[LITER][4];    // mask factors of the tile being REQUESTED (0 outside the image: the conv's zero padding) ...
        float m_cur[LITER][4];    // ... and of the tile being STAGED (at most one tile behind)
        int lk = 0, lc = 0, loaded = 0;      // tile ordinal / chunk of the next item to REQUEST, items requested
        int sk = 0, sc_ = 0, staged = 0, ps = 0;   // ... of the next item to STAGE, items staged, ring slot
        [[maybe_unused]] float vmax = 0.f;   // f16 + fp8 split: running max |x| of everything this lane split (range record, common.h)
        __amdgpu_buffer_rsrc_t rs0, rs1;
        auto setup_tile = [&](const TileId &t) {
            const int iy0 = t.ty * TR - 1, ix0 = t.tx * 32 - 1;
#pragma unroll
            for (int it = 0; it < LITER; ++it) {
                const int idx = ptid + it * NPT;
                const bool has = idx < NLI;
                const int kg = min(idx / (C::HR * NG), NKG - 1);
                const int rem = idx - (idx / (C::HR * NG)) * (C::HR * NG);
                const int pr = rem / NG, g = rem - pr * NG;
                const int gy = iy0 + pr, gx0 = ix0 + 4 * g;
                const bool row_in = has && gy >= 0 && gy < a.Hin;
                // (+16: the descriptors below start 16 bytes in front of the sample's tensor -- the group left of the image edge
                // starts one frame before its row, and a negative offset is out of range for all four frames of the load)
                it_goff[it] = ((row_in ? gy : 0) * a.Win + gx0 + kg * 8 * HW) * AB + 16;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int gx = gx0 + j;
                    const bool in = row_in && gx >= 0 && gx < a.Win;
                    const float m = a.mask[(size_t)t.b * a.T + ((size_t)(in ? gx : 0) << a.lvl_in)];
                    m_nxt[it][j] = in ? m : 0.f;
                }
            }
            const AT *sbase0 = reinterpret_cast<const AT *>(a.src0) + (size_t)t.b * a.c0 * HW;
            const AT *sbase1 = a.c1 > 0 ? reinterpret_cast<const AT *>(a.src1) + (size_t)t.b * a.c1 * HW : sbase0;
            rs0 = uniform_rsrc(reinterpret_cast<const char *>(sbase0) - 16, a.c0 * HW * AB + 16);
            rs1 = uniform_rsrc(reinterpret_cast<const char *>(sbase1) - 16, (a.c1 > 0 ? a.c1 : a.c0) * HW * AB + 16);
        };
        auto write_params = [&](const TileId &t, int par) {
            if constexpr (PRO == PRO_GN) {
                float *sp = s_par + par * 3 * cpad;
                for (int i = ptid; i < cpad; i += NPT) {
                    sp[i] = a.sc[(size_t)t.b * a.cin + i];
                    sp[cpad + i] = a.sh[(size_t)t.b * a.cin + i];
                    sp[2 * cpad + i] = a.tb ? a.tb[(size_t)t.b * a.tb_stride + i] : 0.f;
                }
            }
        };
        // Request side.  load_prep: at a tile switch the new tile's geometry, and its per-channel parameters into
        // s_par[lk & 1] (the tile two back, which used that half, was staged completely long ago; the first reader is two
        // staging steps = two barriers away); returns the scalar offset of the chunk.  load_one: one channel of one item.
        struct LoadCtx { __amdgpu_buffer_rsrc_t rs; int soff; };
        LoadCtx Lc;               // past the last item the previous request is simply repeated (never read): the loads stay
                                  // unconditional straight-line code, so the waits hipcc counts in front of the transforms are exact
        auto load_prep = [&]() {
            if (loaded >= nitems) return;
            if (lc == 0) {
                const TileId t = decode(lk);
                setup_tile(t);
                write_params(t, lk & 1);
            }
            const int cb = lc * CH;
            const bool first = cb < a.c0;
            Lc.rs = first ? rs0 : rs1;
            Lc.soff = (first ? cb : cb - a.c0) * HW * AB;
            ++loaded;
            if (++lc == nchunk) { lc = 0; ++lk; }
        };
        auto load_one = [&](Raw &R, int voff, int it, int i) {
            if (GTTS_WS_EXP == 2) return;
            if constexpr (AB == 4) R.x[it][i] = __builtin_amdgcn_raw_buffer_load_b128(Lc.rs, voff, Lc.soff + i * HW * AB, 0);
            else R.x[it][i] = __builtin_amdgcn_raw_buffer_load_b64(Lc.rs, voff, Lc.soff + i * HW * AB, 0);
        };
        // Stage one item out of R into image dst and re-request R (the item two ahead) CHANNEL BY CHANNEL: the 16-byte load of
        // channel i is issued as soon as its four frames are transformed.  Requested in one burst (32-64 KB per workgroup),
        // the producers' loads queue in front of the consumers' weight-fragment loads in the CU's texture path for longer than
        // those are prefetched ahead (measured: chunk loop 9.3k cycles without the burst, 10.0k / 12.4k with it).
        auto stage_and_load = [&](Raw &R, int chunk, int par, u32x4 *dst) {
            const float *sp = s_par + par * 3 * cpad;
#pragma unroll
            for (int it = 0; it < LITER; ++it) {
                const int idx = ptid + it * NPT;
                const bool has = idx < NLI;
                const int kg = min(idx / (C::HR * NG), NKG - 1);
                const int rem = idx - (idx / (C::HR * NG)) * (C::HR * NG);
                const int pr = rem / NG, g = rem - pr * NG;
                const int cb = chunk * CH + kg * 8;
                float sc[8], sh[8], tb[8];
                if constexpr (PRO == PRO_GN) {
                    const float4 *q0 = reinterpret_cast<const float4 *>(sp + cb);
                    const float4 *q1 = reinterpret_cast<const float4 *>(sp + cpad + cb);
                    const float4 *q2 = reinterpret_cast<const float4 *>(sp + 2 * cpad + cb);
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const float4 u = q0[h], u1 = q1[h], u2 = q2[h];
                        sc[4 * h + 0] = u.x; sc[4 * h + 1] = u.y; sc[4 * h + 2] = u.z; sc[4 * h + 3] = u.w;
                        sh[4 * h + 0] = u1.x; sh[4 * h + 1] = u1.y; sh[4 * h + 2] = u1.z; sh[4 * h + 3] = u1.w;
                        tb[4 * h + 0] = u2.x; tb[4 * h + 1] = u2.y; tb[4 * h + 2] = u2.z; tb[4 * h + 3] = u2.w;
                    }
                }
                bf16x8 vh[4], vl[4];
                [[maybe_unused]] f16x8 fh[4];                   // f16 + fp8 split: fp16 hi values of the four frames ...
                [[maybe_unused]] int lw[4][2], xw[4][2];        // ... and their fp8 operands q8(xl 2^S), q8(x 2^-D): 8 channels = 8 bytes per frame
                [[maybe_unused]] float vprev[4];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (GTTS_WS_EXP != 3) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float v;
                            if constexpr (AB == 4) {
                                const unsigned u = R.x[it][i][j];
                                v = __builtin_bit_cast(float, u);
                            } else {
                                const unsigned u = R.x[it][i][j >> 1];
                                v = __builtin_bit_cast(float, (j & 1) ? (u & 0xffff0000u) : (u << 16));
                            }
                            const float m = m_cur[it][j];
                            if constexpr (PRO == PRO_MASK) {
                                v = mul_mask0(v, m);                 // (m == 0 also marks frames outside the tensor: NaN-proof, common.h)
                            } else {
                                const float y = fmaf(v, sc[i], sh[i]);
                                v = mul_mask0(fmaf(mish_f(y), m, tb[i]), m);
                            }
                            if constexpr (F8) {
                                const _Float16 h = (_Float16)v;
                                fh[j][i] = h;
                                if (i & 1) {        // channels (i - 1, i) of frame j -> two fp8 bytes of word i >> 2, half (i >> 1) & 1
                                    vmax = f8_range_track(vmax, vprev[j], v);
                                    if (i & 2) f8_cross_pair<true>(vprev[j], v, fh[j][i - 1], h, lw[j][i >> 2], xw[j][i >> 2]);
                                    else { lw[j][i >> 2] = 0; xw[j][i >> 2] = 0; f8_cross_pair<false>(vprev[j], v, fh[j][i - 1], h, lw[j][i >> 2], xw[j][i >> 2]); }
                                } else {
                                    vprev[j] = v;
                                }
                            } else if constexpr (NSPLIT > 1) {
                                __bf16 h, l;
                                split_bf16(v, h, l);
                                vh[j][i] = h;
                                vl[j][i] = l;
                            } else {
                                vh[j][i] = (__bf16)v;
                            }
                        }
                    }
                    // the re-request of channel i may not move ahead of its transform (hipcc otherwise copies the four values
                    // aside and issues all eight loads first): an opaque dependence of the load's offset on the last result
                    int voff = it_goff[it];
                    if constexpr (F8) asm volatile("" : "+v"(voff) : "v"(fh[0][i]), "v"(fh[1][i]), "v"(fh[2][i]), "v"(fh[3][i]));
                    else if constexpr (NSPLIT > 1) asm volatile("" : "+v"(voff) : "v"(vl[0][i]), "v"(vl[1][i]), "v"(vl[2][i]), "v"(vl[3][i]));
                    else asm volatile("" : "+v"(voff) : "v"(vh[0][i]), "v"(vh[1][i]), "v"(vh[2][i]), "v"(vh[3][i]));
                    load_one(R, voff, it, i);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (GTTS_WS_EXP != 3) {
                    u32x4 *drow = dst + kg * NPIX + pr * HC + 4 * g;
                    if constexpr (F8) {
                        // fp8 plane: [g8 = plane * 2 + (kg >> 1)][pixel][16 bytes]; this item owns bytes 8 (kg & 1) .. + 7 of its pixels
                        typedef __attribute__((ext_vector_type(2))) int i32x2;
                        i32x2 *d8 = reinterpret_cast<i32x2 *>(dst + PLANE16) + (((kg >> 1) * NPIX + pr * HC + 4 * g) * 2 + (kg & 1));
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (has && 4 * g + j < HC) {
                                drow[j] = __builtin_bit_cast(u32x4, fh[j]);
                                i32x2 q0, q1;
                                q0[0] = lw[j][0]; q0[1] = lw[j][1];
                                q1[0] = xw[j][0]; q1[1] = xw[j][1];
                                d8[2 * j] = q0;
                                d8[2 * (2 * NPIX + j)] = q1;
                            }
                        }
                    } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (has && 4 * g + j < HC) {
                            drow[j] = *reinterpret_cast<u32x4 *>(&vh[j]);
                            if constexpr (NSPLIT > 1) drow[PLANE16 + j] = *reinterpret_cast<u32x4 *>(&vl[j]);
                        }
                    }
                    }
                }
            }
        };
        auto load_all = [&](Raw &R) {      // (launch prologue only)
            load_prep();
#pragma unroll
            for (int it = 0; it < LITER; ++it)
#pragma unroll
                for (int i = 0; i < 8; ++i) load_one(R, it_goff[it], it, i);
        };
        // GroupNorm partial sums of a finished tile (the consumers' wave sums are in s_red[par]) + fused finalize; first
        // producer wave only.  Slot = pixel tile; value order: octets of the group, then the WN wave rows -- fixed.
        auto finish_tile = [&](const TileId &t, int par) {
            // One partial slot per (32-frame column block, 5-row band) = per consumer wave row: the slot's value is formed by ONE
            // wave's sums, so the workgroup shape (four consumer waves, or one in the small-launch form) cannot change it.
            const int gs = a.cout / a.groups;
            const int gpw = MT / gs > 0 ? MT / gs : 1;
            const float *red = s_red + par * (NCW * MF * 8);
            if (lane < gpw * WN) {
                const int gl = lane % gpw, wc = lane / gpw;
                const int g = (t.cot * MT) / gs + gl;
                if (g < a.groups) {
                    const int noct = (gs < MT ? gs : MT) >> 3, o0 = gl * noct;
                    float s1 = 0.f, s2 = 0.f;
                    for (int o = o0; o < o0 + noct; ++o) {
                        const int f = o >> 2, q = o & 3, wmm = f / MF, mi = f - wmm * MF;
                        const int w = wmm * WN + wc;
                        s1 += red[((w * MF + mi) * 4 + q) * 2 + 0];
                        s2 += red[((w * MF + mi) * 4 + q) * 2 + 1];
                    }
                    const int pslot = t.tx * (a.tiles_y * WN) + t.ty * WN + wc;
                    float *p = a.partials + (((size_t)t.b * a.nparts + pslot) * a.groups + g) * 2;
                    if (a.ticket != nullptr) {
                        // write-through (sc1) stores + vmcnt drain + relaxed agent-scope ticket: see common.h (the agent-scope
                        // release would write back every dirty output line of this XCD's L2)
                        __hip_atomic_store(p, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        __hip_atomic_store(p + 1, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    } else {
                        p[0] = s1;
                        p[1] = s2;
                    }
                }
            }
            if (a.ticket == nullptr) return;
#if GTTS_FENCED_FINALIZE
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");          // the textbook form (see common.h): buffer_wbl2 sc1
#endif
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            unsigned old = 0;
            if (lane == 0) old = __hip_atomic_fetch_add(a.ticket + t.b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            old = __builtin_amdgcn_readfirstlane(old);
            if (old != (unsigned)(tps * ncot) - 1) return;
#if GTTS_FENCED_FINALIZE
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
            // last tile of the sample: fixed-order fp64 reduction of ALL partials of the sample (sc1 loads), 8 lanes per group
            const int b = t.b;
            const int g = lane >> 3, sub = lane & 7;
            double s1 = 0.0, s2 = 0.0;
            if (g < a.groups) {
                const float *pp = a.partials + ((size_t)b * a.nparts * a.groups + g) * 2;
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
        };

        // stage the next item out of R into ring slot ps while re-requesting R (the item two ahead)
        auto step = [&](Raw &R) {
            if (staged >= nitems) return;
            [[maybe_unused]] const unsigned long long t0 = WT_NOW();
            load_prep();
            [[maybe_unused]] const unsigned long long t1 = WT_NOW();
            stage_and_load(R, sc_, sk & 1, s_img + ps * IMG16);
            ps = ps + 1 == RING ? 0 : ps + 1;
            ++staged;
            if (++sc_ == nchunk) {
                sc_ = 0;
                ++sk;
#pragma unroll
                for (int it = 0; it < LITER; ++it)
#pragma unroll
                    for (int j = 0; j < 4; ++j) m_cur[it][j] = m_nxt[it][j];      // (the load side is at most one tile ahead)
            }
#if GTTS_WS_TRACE
            if (tr_on) __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the LDS writes are done (timing only)
#endif
            WT_ADD(2, t1, t0);
            WT_ADD(1, WT_NOW(), t1);
        };
        auto fin = [&](int i) {       // consumer item i - 1 closed a tile: its wave sums were written before this barrier
            if (wave == NCWP && i > 0 && i % nchunk == 0) {
                const int kt = i / nchunk - 1;
                finish_tile(decode(kt), kt & 1);
            }
        };
        // f16 + fp8 split: ONE register set -- an item is 32 channels (two of them cover what the bf16x3 form keeps in flight with
        // two sets), and a second 64-register set spills the transform
        constexpr bool ONE_SET = F8;
        load_all(rawA);
#pragma unroll
        for (int it = 0; it < LITER; ++it)
#pragma unroll
            for (int j = 0; j < 4; ++j) m_cur[it][j] = m_nxt[it][j];
        if constexpr (!ONE_SET) load_all(rawB);
        lds_barrier();                                 // (P) s_par of tile 0 visible to every producer wave
        // D items ahead (host guarantees nchunk >= 2: the first two items share a tile, so s_par[0] is all they need);
        // the register sets alternate A, B, A, ... from here on
        step(rawA);
        if (D > 1) step(rawB);
        if constexpr (ONE_SET) {
            static_assert(!ONE_SET || D == 1, "one register set: one item ahead");
            for (int i = 0; i < nitems; ++i) {
                [[maybe_unused]] const unsigned long long tb0 = WT_NOW();
                lds_barrier();
                [[maybe_unused]] const unsigned long long tb1 = WT_NOW();
                WT_ADD(0, tb1, tb0);
                fin(i);
                WT_ADD(3, WT_NOW(), tb1);
                step(rawA);
            }
        } else
        for (int i = 0; i < nitems; i += 2) {
            [[maybe_unused]] unsigned long long tb0 = WT_NOW();
            lds_barrier();
            [[maybe_unused]] unsigned long long tb1 = WT_NOW();
            WT_ADD(0, tb1, tb0);
            fin(i);
            WT_ADD(3, WT_NOW(), tb1);
            if (D & 1) step(rawB); else step(rawA);
            if (i + 1 < nitems) {
                tb0 = WT_NOW();
                lds_barrier();
                tb1 = WT_NOW();
                WT_ADD(0, tb1, tb0);
                fin(i + 1);
                WT_ADD(3, WT_NOW(), tb1);
                if (D & 1) step(rawA); else step(rawB);
            }
        }
        lds_barrier();                                 // (F)
        if (wave == NCWP && nitems > 0) finish_tile(decode(my_tiles - 1), (my_tiles - 1) & 1);
        if constexpr (F8) f8_range_note(a.sat, vmax);
    }
#if GTTS_WS_TRACE
    if (tr_on && lane == 0 && (blockIdx.x >> 2) < 64) {
        tr_sum[4] = __builtin_amdgcn_s_memtime() - tr_entry;
        for (int q = 0; q < 8; ++q) g_ws_trace[((blockIdx.x >> 2) * 16 + wave) * 8 + q] = tr_sum[q];
    }
#endif
}

// ---------------------------------------------------------------------------------------------------- host side
// Which Block convolutions take this kernel: 3x3, whole 16-channel chunks (a concatenated input splitting on a chunk
// boundary), mask / GroupNorm prologue, statistics epilogue, at least 32 input channels, whole 128-channel cout tiles --
// and the fp32-grade bf16x3 precision.  Left on conv_mfma.hip (measured, profiles/r03_conv_ws_findings.txt):
//   * the single-pass bf16 modes (BASELINE config 3): a third of the MFMA work per staged value, the four producer waves
//     cannot keep up with the consumers, and uniform waves on three sub-batch streams are faster (5.9 vs 3.7 ms per call);
//   * 64 output channels: the same MFMA work needs twice the activation staging and has half the chunks per tile to spread
//     the epilogue over; a 64 x 640 form of this kernel was level with conv_mfma.hip at 80 x 1024 (300 vs 306 us) and
//     slower at 40 x 512 (92 vs 79 us).
// LDS of the f16 + fp8 form: two 32-channel images + parameters (mt: 128, or 64 = the 12-wave form of the 64-channel tile)
bool conv_ws_f8_fits(int cin, int pro, int mt, int cout) {
    return ws_smem_bytes(12 * 34, 2, 2, cin, pro, cout, 2, mt == 128 ? 4 : 2, 4, (GTTS_W64_RING && mt == 64) ? 2 * WS64_WST16 : 0) <= (size_t)160 * 1024;
}
bool conv_ws_eligible(int mode, int c0, int c1, int cout, int pro, int epi, int nsplit, int f16f8) {
    const int cin = c0 + c1;
    if (!GTTS_WS || nsplit != 2) return false;
    if (mode != CONV_C3 || epi != EPI_STATS || (pro != PRO_MASK && pro != PRO_GN)) return false;
    if (f16f8) {
        // GTTS_PREC_F16F8: 32-channel chunks (two per tile at least), weights in the f16 + fp8 format (conv_f16f8_ok), two images;
        // 128-channel tiles, or the 64-channel tile (four consumer + eight producer waves)
        if (cout % 128 != 0 && cout != 64) return false;
        if (!conv_f16f8_ok(mode, c0, c1, cout, pro, epi, 1) || cin < 64) return false;
        return conv_ws_f8_fits(cin, pro, cout == 64 ? 64 : 128, cout);
    }
    if (cout % 128 != 0) return false;
    if (cin % 16 != 0 || cin < 32 || (c1 != 0 && c0 % 16 != 0)) return false;
    // three activation images + the per-channel parameters must fit the CU's LDS
    return ws_smem_bytes(12 * 34, 2, 3, cin, pro, cout, 2, 4) <= (size_t)160 * 1024;
}
// GroupNorm partial slots per sample: one per (32-frame column block, 5-row band), whatever the workgroup shape
int conv_ws_nparts(int cout, int Hout, int Wout) {
    (void)cout;
    return ((Wout + 31) / 32) * ((Hout + 4) / 5);
}
// A launch whose regular tiling (128 channels x 10 rows) gives fewer workgroups than this takes the small form: one consumer
// wave on a 32-channel x 5-row tile + two producer waves per workgroup.  A wave's accumulators see the same chunk / pass /
// tap order, a partial slot is still ONE wave's sums over the same five rows, and a group's channel octets are added in
// the same order, so every output and every partial sum is bit-identical to the regular form: results do not depend on
// the batch size.  (Groups wider than the 32-channel tile -- DiffVC's 512- and 1024-channel levels -- stay on the regular
// form: two tiles would share a slot.)
#ifndef GTTS_WS_SMALL_WGS
#define GTTS_WS_SMALL_WGS 160
#endif
bool conv_ws_small(int cout, int groups, int Hout, int Wout, int B, int f16f8) {
    if (groups <= 0 || cout / groups > 32) return false;
    // f16 + fp8, 64 output channels: ONE form at every batch size (its weights are packed by column stage and it accumulates kx-major:
    // the 32-channel small form walks row stages) -- a launch smaller than the chip simply runs fewer 12-wave workgroups
    if (GTTS_W64_RING && f16f8 && cout == 64) return false;
    const long wgs = (long)B * ((Wout + 31) / 32) * ((Hout + 9) / 10) * (cout % 128 == 0 ? cout / 128 : cout / 64);
    return wgs < GTTS_WS_SMALL_WGS;
}

template <int WM, int WN, int MF, int PRO, int NSPLIT, typename AT>
static hipError_t launch_ws_ring(ConvArgs &a, hipStream_t st) {
    constexpr int NF = 5;
    constexpr int NKG = NSPLIT == 3 ? 4 : 2, RING = NSPLIT == 3 ? 2 : 3, NPL = NSPLIT == 3 ? 2 : NSPLIT;
    using C = WsCfg<WM, WN, MF, NF, NKG>;
    a.nchunk = a.cin / (8 * NKG);
    a.tiles_x = (a.Wout + 31) / 32;
    a.tiles_y = (a.Hout + C::TR - 1) / C::TR;
    a.nparts = a.tiles_x * a.tiles_y * WN;
    a.stat_rows = 0;
    if (a.nparts != conv_ws_nparts(a.cout, a.Hout, a.Wout)) return hipErrorInvalidValue;      // (H % 5 rows of slack would differ)
    const size_t lim = (size_t)1 << 31;
    if ((size_t)std::max(a.c0, a.c1) * a.Hin * a.Win * sizeof(AT) >= lim || (size_t)a.cout * a.Hout * a.Wout * sizeof(AT) >= lim)
        return hipErrorInvalidValue;
    static std::atomic<int> n_cu[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int cus = n_cu[dev].load(std::memory_order_relaxed);
    if (cus == 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        n_cu[dev].store(cus, std::memory_order_relaxed);
    }
    const size_t smem = ws_smem_bytes(C::NPIX, NPL, RING, a.cin, PRO, a.cout, MF, C::NCW, NKG, (GTTS_W64_RING && NSPLIT == 3 && C::NCWP != C::NCW) ? 2 * WS64_WST16 : 0);
    if (smem > (size_t)160 * 1024) return hipErrorInvalidValue;      // (conv_ws_eligible keeps such layers on conv_mfma.hip)
    // persistent workgroups: one per CU for the eight-wave form; the three-wave form fits two per CU (registers: 8 waves)
    const int per_cu = C::NT >= 512 ? 1 : (int)std::min<size_t>(2, (size_t)160 * 1024 / smem);
    const long ntiles = (long)a.B * a.tiles_x * a.tiles_y * (a.cout / C::MT);
    // (Measured and not kept: persistent workgroups on half of the CUs per launch -- grid 128 with two sub-batch streams, so
    // that the other stream's kernels find free CUs: 7.45 vs 7.48 ms per call; the dispatcher fills the same CUs first.)
    const int grid = (int)std::min<long>(ntiles, (long)cus * per_cu);
    auto kern = &conv3x3_ws_kernel<WM, WN, MF, NF, PRO, NSPLIT, AT, RING>;
    static std::atomic<size_t> attr_set[64];
    if (smem > attr_set[dev].load(std::memory_order_relaxed)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        attr_set[dev].store(smem, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NT), smem, st, a);
    return hipGetLastError();
}

template <int PRO>
static hipError_t launch_ws_pro(ConvArgs &a, hipStream_t st) {
#ifdef GTTS_WS_PROBE      // compile-time probe builds (register / ISA inspection): one instantiation only
    if constexpr (PRO == PRO_GN) return launch_ws_ring<GTTS_WS_PROBE == 1 ? 1 : 2, GTTS_WS_PROBE == 1 ? 1 : 2, GTTS_WS_PROBE == 1 ? 1 : 2, PRO_GN, 2, float>(a, st);
    else return hipErrorInvalidValue;
#else
    if (a.act_bf16 || a.nsplit != 2) return hipErrorInvalidValue;
    if (a.f16f8) {
        if (conv_ws_small(a.cout, a.groups, a.Hout, a.Wout, a.B, 1)) return launch_ws_ring<1, 1, 1, PRO, 3, float>(a, st);
        if (a.cout % 128 != 0) return launch_ws_ring<1, 2, 2, PRO, 3, float>(a, st);      // the 64-channel tile
        return launch_ws_ring<2, 2, 2, PRO, 3, float>(a, st);
    }
    if (conv_ws_small(a.cout, a.groups, a.Hout, a.Wout, a.B)) return launch_ws_ring<1, 1, 1, PRO, 2, float>(a, st);
    return launch_ws_ring<2, 2, 2, PRO, 2, float>(a, st);
#endif
}

hipError_t launch_conv_ws(const ConvArgs &a_in, hipStream_t st) {
    ConvArgs a = a_in;
    if (!conv_ws_eligible(CONV_C3, a.c0, a.c1, a.cout, a.pro, a.epi, a.nsplit, a.f16f8)) return hipErrorInvalidValue;
    if (a.cin / (a.f16f8 ? 32 : 16) < 2) return hipErrorInvalidValue;
    return a.pro == PRO_GN ? launch_ws_pro<PRO_GN>(a, st) : launch_ws_pro<PRO_MASK>(a, st);
}

}  // namespace gtts
