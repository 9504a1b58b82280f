// conv_up_ws.hip -- Upsample = ConvTranspose2d(dim, dim, 4, 2, 1) of x * mask (Grad-TTS/model/diffusion.py:19-25,171) in the f16 + fp8
// split of GTTS_PREC_F16F8 (common.h), WAVE-SPECIALISED: the MFMA waves issue no activation load and no output store.
//
// Why (measured on conv_up.hip's uniform-wave f16 + fp8 kernel, us per launch, 128- / 64-channel layer at B = 16, T = 1024): as built
// 111.6 / 152; without its output stores 106.5 / 121; without its activation loads 81 / 98; neither 77 / 87.  A wave's memory
// operations complete in order as far as s_waitcnt can tell (loads and stores share vmcnt), so a wave that waits for its next tap's
// weight fragments -- L2 hits, needed every 512 cycles -- also waits for every HBM activation load and every output store it issued
// before them.  Here, as in conv_ws.hip, other waves issue those:
//   waves 0-3  CONSUMERS, one per SIMD: wave = (output row parity py, output column parity px); a tile is 64 output channels x 2 input
//              rows x 32 input columns (-> 4 x 64 output pixels): accumulators [2 x 32 channels][2 rows] = 64 registers.  Weight
//              fragments straight from the fragment-ordered blob (pack.hip, CONV_UP | 32), FOUR sets deep = three taps (1536 MFMA
//              cycles) ahead: nothing else is in these waves' memory queue.  A finished tile goes to LDS (16 x 16 bytes per lane).
//   waves 4-7  PRODUCERS: halo tile (4 x 34 pixels) of the next 32-channel chunk: 16-byte global loads two chunks ahead of the staging (two register sets),
//              x * mask, fp16 hi + fp8 cross-term operands, LDS image (two images: one barrier per chunk); and the OUTPUT: producer
//              (py, 32-channel half) picks both column parities of its rows out of the consumers' LDS tile and stores them as 8 bytes
//              per lane (64 lanes = full lines), sixteen stores per chunk step.  Every step issues exactly sixteen stores -- steps with
//              nothing to store aim them past the end of the buffer (dropped by the range check) -- so that the wait in front of the
//              next chunk's activation registers is a COUNTED one (vmcnt(16)), never a wait for the stores' acknowledgements.
// MEASURED (round 6, same box, us per launch, 128- / 64-channel layer): 114.5 / 132.3 against 110 / 142.6 for the uniform-wave kernel:
// level, so the product keeps that one (GTTS_UP_WS = 0).  Traced (GTTS_UPW_TRACE, cycles of one workgroup, 64-channel layer, 40 chunk
// steps): consumers 160-170k in their taps for 82k of MFMA issue + 40-50k at barriers; producers 170-180k in staging + stores.  What
// binds both is the CU's ONE vector-memory path: per chunk step the four consumers fetch 128 KB of weight fragments (a fragment feeds
// two MFMAs: the tile is two rows) -- 2k cycles of the L1's 64 bytes per clock -- and the producers' 32 KB of output leave at the
// ~13 bytes per clock a CU gets while every CU stores (2.4k cycles, whether the stores are issued back to back or spread through the
// staging code).  Larger tiles per weight fragment need accumulators that do not exist (4 rows: 128 + 128 weight registers).
// Also measured: workgroups walking CONSECUTIVE tiles (long output runs per CU, GTTS_UPW_BLOCKED) 148 / 160 -- worse.
// Persistent: one workgroup per CU walks tiles pos, pos + grid, ... in XCD-banded order.  Per-accumulator order: chunk, stage (ky),
// tap (kx): k-step 0, k-step 1, fp8 -- as in conv_up.hip's f16 + fp8 kernel, whose results this kernel reproduces bit for bit.
#include "common.h"
#include "kernels.h"
#include <atomic>
#include <type_traits>

namespace gtts {

constexpr int UW_HC = 34, UW_HR = 4, UW_NPIX = UW_HR * UW_HC;      // halo tile: 4 x 34 pixels
constexpr int UW_MAXC = 1024;                                      // output channels the bias vector in LDS holds
constexpr int UW_OOB = 0x7ffffff0;                                 // a byte offset past any buffer: the store is dropped
#ifndef GTTS_UPW_TRACE    // diagnostic builds: one workgroup prints where its waves' cycles went (s_memtime, 100 MHz)
#define GTTS_UPW_TRACE 0
#endif
#if GTTS_UPW_TRACE
#define UWT_NOW() __builtin_amdgcn_s_memtime()
#define UWT_ADD(i, b, a) (tr[i] += (b) - (a))
#else
#define UWT_NOW() 0ull
#define UWT_ADD(i, b, a) ((void)0)
#endif
#ifndef GTTS_UPW_BLOCKED
#define GTTS_UPW_BLOCKED 0
#endif
#ifndef GTTS_UPW_ABL      // timing ablations (results are WRONG): bit 0 no output stores, bit 1 no activation loads
#define GTTS_UPW_ABL 0
#endif

__global__ __launch_bounds__(512, 2) void conv_up4_ws_kernel(const ConvArgs a) {
    __shared__ __attribute__((aligned(16))) u32x4 s_img[2][2][4 * UW_NPIX];      // [buffer][fp16 | fp8 plane][kg / g][pixel] (34 KB)
    __shared__ __attribute__((aligned(16))) u32x4 s_out[4 * 16 * 64];            // finished tile: [consumer wave][4-register group][lane] (64 KB)
    __shared__ float s_bias[UW_MAXC];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kg_l = lane >> 5;

    const int ncot = a.cout / 64;
    const int ntiles = a.B * a.tiles_x * a.tiles_y * ncot, G = gridDim.x;
    const int HW = a.Hin * a.Win, HWo = a.Hout * a.Wout;
    const int nch = a.cin / 32;                                    // (even: launch_conv_up4_ws)
    const int pos = xcd_slot(blockIdx.x, G);
#if GTTS_UPW_BLOCKED      // a workgroup walks CONSECUTIVE tiles (x fastest): its output rows are written as long runs
    const int per = (ntiles + G - 1) / G;
    const int my_tiles = max(0, min(per, ntiles - pos * per));
#else
    const int my_tiles = pos < ntiles ? (ntiles - pos + G - 1) / G : 0;
#endif
    const int nitems = my_tiles * nch;
    struct Tile { int cot, b, y0, x0; };
    auto decode = [&](int k) {                                     // k-th tile of this workgroup (past the end: the last one)
#if GTTS_UPW_BLOCKED
        int t = pos * per + min(k, max(my_tiles - 1, 0));
#else
        int t = pos + k * G;
#endif
        t = t < ntiles ? t : ntiles - 1;
        Tile r;
        r.cot = t % ncot; t /= ncot;
        r.x0 = (t % a.tiles_x) * 32; t /= a.tiles_x;
        r.y0 = (t % a.tiles_y) * 2;
        r.b = t / a.tiles_y;
        return r;
    };
    auto uniform_rsrc = [](const void *p, int bytes) {
        const unsigned long long u = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo), 0,
                                                 __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
    };
    for (int c = tid; c < a.cout; c += 512) s_bias[c] = a.bias[c];      // (visible after the first barrier)
    [[maybe_unused]] unsigned long long tr[4] = {0, 0, 0, 0};
    [[maybe_unused]] const unsigned long long tr_t0 = UWT_NOW();

    if (wave < 4) {
        // =================================================================================== CONSUMERS
        __builtin_amdgcn_s_setprio(3);
        const int py = wave & 1, px = wave >> 1, phase = py * 2 + px;
        const int MTP = a.cout > 64 ? 128 : 64, ncotp = a.cout / MTP, cpp = MTP / 64;
        const int wblk16 = 16 * MTP;                  // 16-byte units of one packed block: [fp16: tap 2][kg 4][MTP] [fp8: tap 2][g 4][MTP]
        const __amdgpu_buffer_rsrc_t rsw = uniform_rsrc(a.w, 4 * nch * 2 * ncotp * wblk16 * 16);
        const int wl_h = (kg_l * MTP + l31) * 16;     // lane's row in a (tap, kg pair) segment of the fp16 plane
        const int wl_8 = (kg_l * 2 * MTP + l31) * 16; // ... of the fp8 plane (g = kg_l * 2 + q)
        struct WSet { f16x8 a[2], b[2]; i32x8 w8[2]; };      // [mi]
        auto wload = [&](WSet &w, int cot, int chunk, int u) {      // tap u of a chunk: stage (ky) u >> 1, tap (kx) u & 1
            const int stage = u >> 1, tap = u & 1;
            const int blk = ((phase * nch + chunk) * 2 + stage) * ncotp + cot / cpp;
            const int so = blk * (wblk16 * 16) + (cot % cpp) * 64 * 16;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi) {
                const u32x4 va = __builtin_amdgcn_raw_buffer_load_b128(rsw, wl_h + mi * 32 * 16, so + ((tap * 4) * MTP) * 16, 0);
                const u32x4 vb = __builtin_amdgcn_raw_buffer_load_b128(rsw, wl_h + mi * 32 * 16, so + ((tap * 4 + 2) * MTP) * 16, 0);
                const u32x4 q0 = __builtin_amdgcn_raw_buffer_load_b128(rsw, wl_8 + mi * 32 * 16, so + (((2 + tap) * 4) * MTP) * 16, 0);
                const u32x4 q1 = __builtin_amdgcn_raw_buffer_load_b128(rsw, wl_8 + mi * 32 * 16, so + (((2 + tap) * 4 + 1) * MTP) * 16, 0);
                w.a[mi] = __builtin_bit_cast(f16x8, va);
                w.b[mi] = __builtin_bit_cast(f16x8, vb);
                w.w8[mi][0] = (int)q0[0]; w.w8[mi][1] = (int)q0[1]; w.w8[mi][2] = (int)q0[2]; w.w8[mi][3] = (int)q0[3];
                w.w8[mi][4] = (int)q1[0]; w.w8[mi][5] = (int)q1[1]; w.w8[mi][6] = (int)q1[2]; w.w8[mi][7] = (int)q1[3];
            }
        };
        f32x16 acc[2][2];      // [mi][row]
        auto zero_acc = [&]() {
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;
        };
        zero_acc();
        WSet ws[4];            // set u: tap u of the chunk being multiplied; refilled three taps ahead
        int k = 0, cc = 0;
        Tile tl = decode(0);
        if (nitems > 0) {
#pragma unroll
            for (int u = 0; u < 3; ++u) wload(ws[u], tl.cot, 0, u);
        }
        for (int i = 0; i < nitems; ++i) {
            [[maybe_unused]] const unsigned long long c0 = UWT_NOW();
            lds_barrier();                                          // image of item i is complete; the producers are done with the tile in s_out
            [[maybe_unused]] const unsigned long long c1 = UWT_NOW();
            UWT_ADD(0, c1, c0);
            const int buf = i & 1;
            const bool lastc = cc + 1 == nch;
            const Tile nx = decode(lastc ? k + 1 : k);
            const int nchk = lastc ? 0 : cc + 1;                    // (past the last item: tile my_tiles - 1 again, never used)
            const u32x4 *xh_p = &s_img[buf][0][kg_l * UW_NPIX];                  // fp16 plane, k-step 0 (k-step 1: + 2 NPIX)
            const u32x4 *x8_p = &s_img[buf][1][kg_l * 2 * UW_NPIX];              // fp8 plane: g = 2 kg_l (second half: + NPIX)
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int st = u >> 1, j = u & 1;
                // three taps ahead: tap 3 of this chunk at tap 0, then taps 0..2 of the next chunk (or of the next tile's first)
                if (u == 0) wload(ws[3], tl.cot, cc, 3);
                else wload(ws[u - 1], nx.cot, nchk, u - 1);
                __builtin_amdgcn_sched_barrier(0);
                const int dyr = py == 0 ? (st == 0 ? 0 : -1) : (st == 0 ? 1 : 0);
                const int dxc = px == 0 ? (j == 0 ? 0 : -1) : (j == 0 ? 1 : 0);
                f16x8 fa[2], fb[2];
                i32x8 b8[2];
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int pi = (ni + 1 + dyr) * UW_HC + 1 + dxc + l31;
                    fa[ni] = __builtin_bit_cast(f16x8, xh_p[pi]);
                    fb[ni] = __builtin_bit_cast(f16x8, xh_p[2 * UW_NPIX + pi]);
                    const u32x4 q0 = x8_p[pi], q1 = x8_p[UW_NPIX + pi];
                    b8[ni][0] = (int)q0[0]; b8[ni][1] = (int)q0[1]; b8[ni][2] = (int)q0[2]; b8[ni][3] = (int)q0[3];
                    b8[ni][4] = (int)q1[0]; b8[ni][5] = (int)q1[1]; b8[ni][6] = (int)q1[2]; b8[ni][7] = (int)q1[3];
                }
                const WSet &wc = ws[u];
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc.a[mi], fa[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wc.b[mi], fb[ni], acc[mi][ni], 0, 0, 0);
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(wc.w8[mi], b8[ni], acc[mi][ni], 0, 0, 0, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            UWT_ADD(1, UWT_NOW(), c1);
            if (!lastc) { ++cc; continue; }
            [[maybe_unused]] const unsigned long long c2 = UWT_NOW();
            // ---- tile done: acc 2^-S + bias -> LDS, group (mi, ni, rg >> 2) = four channels (rg & 3) of one pixel
            const float *bias_l = s_bias + tl.cot * 64 + 4 * kg_l;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        u32x4 v;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            v[e] = __builtin_bit_cast(unsigned, fmaf(acc[mi][ni][4 * q + e], 1.0f / (float)(1 << F8_S), bias_l[mi * 32 + 8 * q + e]));
                        s_out[(wave * 16 + (mi * 2 + ni) * 4 + q) * 64 + lane] = v;
                    }
            zero_acc();
            cc = 0;
            ++k;
            tl = nx;
            UWT_ADD(2, UWT_NOW(), c2);
        }
        lds_barrier();                                              // (F) the last tile is in s_out
    } else {
        // =================================================================================== PRODUCERS
        const int ptid = tid - 256, pw = wave - 4;
        const int opy = pw & 1, omi = pw >> 1;                      // output side: rows of parity opy, channels omi * 32 ..
        // Staging work of a thread per 32-channel chunk (a memory queue holds 63 operations -- vmcnt -- so the loads are wide: with dword
        // loads two chunks in flight + a step's sixteen stores were 64, and every request stalled for an HBM round trip: traced, 54 % of
        // the producers' cycles):
        //   MAIN   (all 256 threads) halo row r, FOUR consecutive frames 1 + 4 g .. of the 32 inner columns (16-byte loads, aligned to the
        //          tile), four channels 4 h .. of 8-channel group kg: 16 values -> four 8-byte fp16 writes + eight 4-byte fp8 writes
        //   EDGE   (threads 0..127) halo row r, the column left (frame 0) or right (frame 33) of the tile, channel pair p of group kg
        const int m_g = ptid & 7, m_r = (ptid >> 3) & 3, m_h = (ptid >> 5) & 1, m_kg = ptid >> 6;
        const int e_p = ptid & 3, e_side = (ptid >> 2) & 1, e_r = (ptid >> 3) & 3, e_kg = (ptid >> 5) & 3;
        const bool e_has = ptid < 128;
        int m_off = 0, e_off = 0;
        // mask factors: the tile at the request cursor keeps INDICES and in-image bits only; the values are loaded with every request, into
        // the item's register set, and first used when it is staged -- a value consumed at the tile switch put a vmcnt(0), i.e. a wait for
        // every activation load and output store in flight, into every tile (traced: half of the producers' cycles)
        int m_mi[4] = {0, 0, 0, 0}, e_mi = 0, m_in = 0, e_in = 0;
        float m_ms[2][4], e_ms[2];
        int m_ins[2], e_ins[2];
        __amdgpu_buffer_rsrc_t rsx;
        const __amdgpu_buffer_rsrc_t rsm = uniform_rsrc(a.mask, a.B * a.T * 4);
        auto set_items = [&](const Tile &tl) {
            rsx = uniform_rsrc(reinterpret_cast<const float *>(a.src0) + (size_t)tl.b * a.cin * HW, a.cin * HW * 4);
            {
                const int gy = tl.y0 - 1 + m_r, gx0 = tl.x0 + 4 * m_g;
                const bool row_in = gy >= 0 && gy < a.Hin;
                // (frames past the row's end read the next row -- or, behind the tensor, nothing: the range check returns 0 -- and the mask factor zeroes them)
                m_off = ((row_in ? gy : 0) * a.Win + gx0 + (m_kg * 8 + 4 * m_h) * HW) * 4;
                m_in = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int gx = gx0 + j;
                    const bool in = row_in && gx < a.Win;
                    m_mi[j] = (tl.b * a.T + ((in ? gx : 0) << a.lvl_in)) * 4;
                    m_in |= in ? 1 << j : 0;
                }
            }
            {
                const int gy = tl.y0 - 1 + e_r, gx = e_side ? tl.x0 + 32 : tl.x0 - 1;
                const bool in = e_has && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
                e_off = in ? (gy * a.Win + gx + (e_kg * 8 + 2 * e_p) * HW) * 4 : 0;
                e_mi = (tl.b * a.T + ((in ? gx : 0) << a.lvl_in)) * 4;
                e_in = in ? 1 : 0;
            }
        };
        unsigned rawm[2][4][4];                 // two items in flight: set = item parity; [channel 4 h + c][frame]
        float rawe[2][2];
        float vmax = 0.f;                       // activation range record (common.h)
        int rk = 0, rc = 0, requested = 0, rq_chunk = 0;      // tile ordinal / chunk of the next item to request; chunk of the last request
        auto request = [&](auto set_c) {        // (past the last item: the previous request again, never staged: straight-line code)
            constexpr int rs_ = decltype(set_c)::value;
            if (requested < nitems) {
                if (rc == 0) set_items(decode(rk));
                rq_chunk = rc;
                ++requested;
                if (++rc == nch) { rc = 0; ++rk; }
            }
            const int soff = rq_chunk * 32 * HW * 4;
            m_ins[rs_] = m_in;
            e_ins[rs_] = e_in;
#pragma unroll
            for (int j = 0; j < 4; ++j) m_ms[rs_][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsm, m_mi[j], 0, 0));
            e_ms[rs_] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsm, e_mi, 0, 0));
#if GTTS_UPW_ABL & 2      // timing ablation: no activation loads
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int j = 0; j < 4; ++j) rawm[rs_][c][j] = (unsigned)soff;
            rawe[rs_][0] = rawe[rs_][1] = 0.f;
#else
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rsx, m_off, soff + c * HW * 4, 0);
                rawm[rs_][c][0] = r[0]; rawm[rs_][c][1] = r[1]; rawm[rs_][c][2] = r[2]; rawm[rs_][c][3] = r[3];
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) rawe[rs_][c] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsx, e_off, soff + c * HW * 4, 0));
#endif
        };
        // staging of the item of parity buf (image buf, register set buf): frame j of the main item; the edge item
        auto stage_main = [&](auto buf_c, auto j_c) {
            constexpr int buf = decltype(buf_c)::value, j = decltype(j_c)::value;
            typedef __attribute__((ext_vector_type(4))) _Float16 f16x4;
            unsigned char *ph = reinterpret_cast<unsigned char *>(&s_img[buf][0][0]);      // fp16 plane: [kg][pixel] x 16 bytes
            unsigned char *p8 = reinterpret_cast<unsigned char *>(&s_img[buf][1][0]);      // fp8 plane: [g = term * 2 + (kg >> 1)][pixel] x 16 bytes
            const float m = ((m_ins[buf] >> j) & 1) ? m_ms[buf][j] : 0.f;
            float v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = mul_mask0(__builtin_bit_cast(float, rawm[buf][c][j]), m);
            f16x4 fh;
#pragma unroll
            for (int c = 0; c < 4; ++c) fh[c] = (_Float16)v[c];
            vmax = f8_range_track(f8_range_track(vmax, v[0], v[1]), v[2], v[3]);
            int lw = 0, xw = 0;
            f8_cross_pair<false>(v[0], v[1], fh[0], fh[1], lw, xw);
            f8_cross_pair<true>(v[2], v[3], fh[2], fh[3], lw, xw);
            const int pix = m_r * UW_HC + 1 + 4 * m_g + j;
            *reinterpret_cast<f16x4 *>(ph + (m_kg * UW_NPIX + pix) * 16 + 8 * m_h) = fh;
            const int o8 = ((m_kg >> 1) * UW_NPIX + pix) * 16 + 8 * (m_kg & 1) + 4 * m_h;
            *reinterpret_cast<int *>(p8 + o8) = lw;
            *reinterpret_cast<int *>(p8 + o8 + 2 * UW_NPIX * 16) = xw;
        };
        auto stage_edge = [&](auto buf_c) {
            constexpr int buf = decltype(buf_c)::value;
            unsigned char *ph = reinterpret_cast<unsigned char *>(&s_img[buf][0][0]);
            unsigned char *p8 = reinterpret_cast<unsigned char *>(&s_img[buf][1][0]);
            if (e_has) {
                typedef __attribute__((ext_vector_type(2))) _Float16 f16x2;
                const float m = e_ins[buf] ? e_ms[buf] : 0.f;
                const float v0 = mul_mask0(rawe[buf][0], m), v1 = mul_mask0(rawe[buf][1], m);
                f16x2 fh;
                fh[0] = (_Float16)v0;
                fh[1] = (_Float16)v1;
                vmax = f8_range_track(vmax, v0, v1);
                int lw = 0, xw = 0;
                f8_cross_pair<false>(v0, v1, fh[0], fh[1], lw, xw);
                const int pix = e_r * UW_HC + (e_side ? 33 : 0);
                *reinterpret_cast<f16x2 *>(ph + (e_kg * UW_NPIX + pix) * 16 + 4 * e_p) = fh;
                const int o8 = ((e_kg >> 1) * UW_NPIX + pix) * 16 + 8 * (e_kg & 1) + 2 * e_p;
                *reinterpret_cast<unsigned short *>(p8 + o8) = (unsigned short)(lw & 0xffff);
                *reinterpret_cast<unsigned short *>(p8 + o8 + 2 * UW_NPIX * 16) = (unsigned short)(xw & 0xffff);
            }
        };
        auto stage = [&](auto buf_c) {
            stage_main(buf_c, std::integral_constant<int, 0>{});
            stage_main(buf_c, std::integral_constant<int, 1>{});
            stage_main(buf_c, std::integral_constant<int, 2>{});
            stage_main(buf_c, std::integral_constant<int, 3>{});
            stage_edge(buf_c);
        };
        // ---- output side: the finished tile, both column parities of rows (ni, opy), channels omi * 32 + 8 q + e (+ 4 kg_l)
        u32x4 park[2][4][2];                    // [ni][q][px]
        Tile pt = decode(0);                    // the tile the parked values belong to
        auto take = [&](const Tile &t) {
            pt = t;
#pragma unroll
            for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int px = 0; px < 2; ++px) park[ni][q][px] = s_out[((opy + 2 * px) * 16 + (omi * 2 + ni) * 4 + q) * 64 + lane];
        };
        // four of the sixteen 8-byte stores of row ni of the parked tile: channels omi * 32 + 8 q + 0..3 (valid == false: dropped)
        auto put_q = [&](auto ni_c, auto q_c, bool valid) {
            constexpr int ni = decltype(ni_c)::value, q = decltype(q_c)::value;
            typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
            const __amdgpu_buffer_rsrc_t rso = uniform_rsrc(reinterpret_cast<float *>(a.out) + (size_t)pt.b * a.cout * HWo, a.cout * HWo * 4);
            const int ix = pt.x0 + l31, iy = pt.y0 + ni;
            const bool ok = valid && iy < a.Hin && ix < a.Win;
            const int voff = ok ? ((2 * iy + opy) * a.Wout + 2 * ix + 4 * kg_l * HWo) * 4 : UW_OOB;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                u32x2 v;
                v[0] = park[ni][q][0][e];
                v[1] = park[ni][q][1][e];
#if GTTS_UPW_ABL & 1      // timing ablation: no output stores
                if (v[0] != 0x12345678u) continue;
#endif
                __builtin_amdgcn_raw_buffer_store_b64(v, rso, voff, (pt.cot * 64 + omi * 32 + 8 * q + e) * HWo * 4, GTTS_OUT_NT);
            }
        };
        auto put = [&](auto ni_c, bool valid) {
            put_q(ni_c, std::integral_constant<int, 0>{}, valid);
            put_q(ni_c, std::integral_constant<int, 1>{}, valid);
            put_q(ni_c, std::integral_constant<int, 2>{}, valid);
            put_q(ni_c, std::integral_constant<int, 3>{}, valid);
        };
        // step g (after barrier g): pick up a finished tile, stage item g + 1, request item g + 3, sixteen stores
        auto step = [&](int g, auto par_c) {
            constexpr int par = decltype(par_c)::value;
            bool valid;
            [[maybe_unused]] const unsigned long long p0 = UWT_NOW();
            if constexpr (par == 0) {
                valid = g > 0 && g % nch == 0;            // item g - 1 closed a tile
                if (valid) take(decode(g / nch - 1));
            } else {
                valid = g > 1 && (g - 1) % nch == 0;
            }
            [[maybe_unused]] const unsigned long long p1 = UWT_NOW();
            // The stores are spread through the staging code, four behind each frame: when every CU stores the chip's write path takes
            // ~13 bytes per clock and CU (r06_mem_probe.txt) -- a wave that issues its sixteen stores back to back sits in them for 2.5k
            // cycles (traced), as long as it needs to stage and request an item; interleaved, the two overlap
            const bool st = g + 1 < nitems;
            constexpr auto bc = std::integral_constant<int, 1 - par>{};
            if (st) stage_main(bc, std::integral_constant<int, 0>{});
            put_q(par_c, std::integral_constant<int, 0>{}, valid);
            __builtin_amdgcn_sched_barrier(0);
            if (st) stage_main(bc, std::integral_constant<int, 1>{});
            put_q(par_c, std::integral_constant<int, 1>{}, valid);
            __builtin_amdgcn_sched_barrier(0);
            if (st) stage_main(bc, std::integral_constant<int, 2>{});
            put_q(par_c, std::integral_constant<int, 2>{}, valid);
            __builtin_amdgcn_sched_barrier(0);
            if (st) stage_main(bc, std::integral_constant<int, 3>{});
            put_q(par_c, std::integral_constant<int, 3>{}, valid);
            __builtin_amdgcn_sched_barrier(0);
            if (st) stage_edge(bc);
#if GTTS_UPW_TRACE
            __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): the LDS writes are done (timing only)
#endif
            [[maybe_unused]] const unsigned long long p2 = UWT_NOW();
            request(bc);                                                             // item g + 3 into the set item g + 1 leaves
            [[maybe_unused]] const unsigned long long p2b = UWT_NOW();
            [[maybe_unused]] const unsigned long long p3 = p2b;
            UWT_ADD(1, p2b, p2); UWT_ADD(2, p2, p1); UWT_ADD(3, p3, p2b);      // [1] request (take is small), [2] stage, [3] put
        };
        if (nitems > 0) {
            request(std::integral_constant<int, 0>{});      // items 0, 1, 2: two stay in flight
            request(std::integral_constant<int, 1>{});
            stage(std::integral_constant<int, 0>{});
            request(std::integral_constant<int, 0>{});
        }
        for (int g = 0; g < nitems; g += 2) {     // (nitems is even)
            [[maybe_unused]] unsigned long long b0 = UWT_NOW();
            lds_barrier();
            UWT_ADD(0, UWT_NOW(), b0);
            step(g, std::integral_constant<int, 0>{});
            b0 = UWT_NOW();
            lds_barrier();
            UWT_ADD(0, UWT_NOW(), b0);
            step(g + 1, std::integral_constant<int, 1>{});
        }
        lds_barrier();                            // (F)
        if (nitems > 0) {
            take(decode(my_tiles - 1));
            put(std::integral_constant<int, 0>{}, true);
            put(std::integral_constant<int, 1>{}, true);
        }
        f8_range_note(a.sat, vmax);
    }
#if GTTS_UPW_TRACE
    if (blockIdx.x == 37 && lane == 0)
        printf("upw cin %d wave %d items %d total %llu : %llu %llu %llu %llu\n", a.cin, wave, nitems, UWT_NOW() - tr_t0, tr[0], tr[1], tr[2], tr[3]);
#endif
}

// layers launch_conv_up4 hands over (conv_up.hip): f16 + fp8 packing, whole 64-channel chunk pairs, a bias vector that fits LDS
bool conv_up4_ws_ok(int cin, int cout) { return GTTS_UP_WS && cin % 64 == 0 && cin >= 64 && cout % 64 == 0 && cout <= UW_MAXC; }

const char *conv_up4_ws_name() { return "gtts::conv_up4_ws_kernel"; }

hipError_t launch_conv_up4_ws(const ConvArgs &a_in, hipStream_t st) {
    ConvArgs a = a_in;
    if (!conv_up4_ws_ok(a.cin, a.cout)) return hipErrorInvalidValue;
    a.tiles_x = (a.Win + 31) / 32;
    a.tiles_y = (a.Hin + 1) / 2;
    const long tiles = (long)a.B * a.tiles_x * a.tiles_y * (a.cout / 64);
    if (tiles <= 0 || tiles > 0x7fffffffL) return hipErrorInvalidValue;
    static std::atomic<int> n_cu[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    int cus = n_cu[dev].load(std::memory_order_relaxed);
    if (cus == 0) {
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        n_cu[dev].store(cus, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL(conv_up4_ws_kernel, dim3((unsigned)std::min<long>(tiles, cus)), dim3(512), 0, st, a);      // one workgroup per CU
    return hipGetLastError();
}

}  // namespace gtts
