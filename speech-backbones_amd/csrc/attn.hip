// attn.hip -- LinearAttention (Grad-TTS/model/diffusion.py:82-100, wrapped by Rezero :39-46 and Residual
// :103-110) restructured for gfx950 so that q/k/v are never materialised in HBM.
//
// Reference:  qkv = to_qkv(x); k = softmax_n(k); ctx[h][d][e] = sum_n k[d,n] v[e,n];
//             out[h][e][n] = sum_d ctx[d][e] q[d,n];  y = to_out(out) + b;  result = y * g + x
// Algebra:    q = Wq x is linear, so   y[:, n] = (Wout . blockdiag_h(ctx_h^T) . Wq) x[:, n] + b = M_b x[:, n] + b
//             with one C x C matrix M_b per sample.  Hence:
//   pass 1  attn_ctx    per (sample, head, pixel slice): k,v = Wkv_h x on the MFMA pipe, flash-style online
//                       softmax over n (running max m, normaliser Z) and ctx += P V^T on the MFMA pipe with
//                       the projection's own accumulator registers as operands (no LDS transpose: the
//                       k-slot <-> pixel mapping of the C/D layout is identical for P and V).
//           attn_merge  log-sum-exp merge of the per-wave partials -> normalised ctx[b][h][32][32]
//           attn_fold   M_b = g * Wout . blockdiag(ctx^T) . Wq (fp32 VALU, tiny), written straight into the
//                       packed bf16 hi/lo layout of the 1x1 MFMA convolution; bias_b = g * b
//   pass 2  conv_mfma CONV_P1 with per-sample weights and EPI_ATTN (+ x): reads x once, writes once.
// Softmax quirks kept: over ALL h*w positions, no mask, no 1/sqrt(d) scaling, q not normalised.
#include <cstring>
#include "common.h"
#include "kernels.h"

// heads per workgroup of the per-head context kernel (C >= 128): see attn_ctx_kernel
// (GTTS_ATTN_HPW: heads per workgroup of the C >= 128 context kernel -- kernels.h, shared with the kernel-name table of plan.hip)

namespace gtts {

struct AttnCtxArgs {
    const void *x;
    const unsigned char *wkv;
    float *partials;
    int C, HW, nstage, tiles, tps, nrec, nsplit, B;
    AttnTail tail;          // attn_ctx64_kernel<..., TAIL = 1>: the fused identity tail (x is written by the launch)
};

__device__ __forceinline__ void pack8_split(const float (&v)[8], u32x4 &hi, u32x4 &lo) {
    bf16x8 vh, vl;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        __bf16 h, l;
        split_bf16(v[i], h, l);
        vh[i] = h;
        vl[i] = l;
    }
    hi = *reinterpret_cast<u32x4 *>(&vh);
    lo = *reinterpret_cast<u32x4 *>(&vl);
}

// FULLC: C is a multiple of 32 (every staged channel exists)
// (Measured and not kept, round 3: head PAIRS per workgroup -- the x tile loaded, split and staged once for two heads, each
// head with its own projection accumulators / softmax state / context, records bit-identical.  128 + 32 accumulator registers
// beside the 32-register x prefetch do not fit 256 VGPRs: 54 spilled registers, 81.6 vs 62.0 us per launch.)
// HPW = heads per workgroup (256 threads per head).  HPW = 2: the x tile is loaded, split into bf16 hi/lo and staged ONCE for
// two heads by 512 threads -- half the staging work per wave (the kernel issues ten VALU instructions per MFMA, more than
// half of them in the staging); every wave does exactly the arithmetic of the one-head form on the same pixels in the same
// order and the four waves of a head are merged as before, so the records are bit-identical.  One 8-wave workgroup per CU =
// the two waves per SIMD of the one-head form.  Measured (B = 16, one stream, alternating repeats on one box): C = 128 at
// 40 x 512: 117.3 -> 103.8 us; C = 256 at 20 x 256: 58.5 -> 57.8 us (one tile per workgroup: prologue, merge and the launch's
// tail bound it, not the staging); sampler outputs bit-identical in all three precisions.
// TAIL = 1 (FULLC, fp32 storage): the ResnetBlock identity tail in front of the attention rides in the staging step, as in
// attn_ctx64_kernel -- a thread loads the block's input and the raw output of its second convolution for its 8-channel groups,
// forms x = xin m + Mish(GN(h)) m (common.h::tail_value, bit-identical to tail_identity_kernel) and stages it; the workgroup of the
// pixel slice's FIRST head group also stores it (the other head groups of the slice recompute the same values from the same -- by
// then L2-resident -- inputs: the Mish runs 4 / HPW times per element, which is what the separate tail pass's write + re-read cost).
template <int NSPLIT, int FULLC, typename AT = float, int HPW = 1, int TAIL = 0>
__global__ __launch_bounds__(256 * HPW, HPW == 1 ? 2 : 1) void attn_ctx_kernel(const AttnCtxArgs a) {
    constexpr int AB = (int)sizeof(AT);
    static_assert(!TAIL || (FULLC && AB == 4), "the fused tail: whole 32-channel stages, fp32 storage");
    constexpr int KCH = ATTN_KCH, NKG = 2 * KCH;
    constexpr int NKGT = NKG / HPW;                     // channel groups a thread loads and stages per stage
    constexpr int WHEAD = 2 * NKG * 64;                 // 16-byte units of one head's weight stage
    static_assert(HPW == 1 || HPW == 2, "one or two heads per workgroup");
    // one buffer: [x hi NKG*256][x lo NKG*256][weights HPW * WHEAD]; the final merge re-uses it as [4 HPW waves][ATTN_REC] floats
    constexpr int SM16 = 2 * NKG * 256 + HPW * WHEAD;
    static_assert((size_t)SM16 * 16 >= (size_t)4 * HPW * ATTN_REC * 4, "merge records must fit the staging buffer");
    __shared__ __attribute__((aligned(16))) u32x4 s_all[SM16];
    u32x4 *s_ah = s_all, *s_al = s_all + NKG * 256;
    u32x4 *s_wall = s_all + 2 * NKG * 256;                // [head in workgroup][split][kg][64 rows: k_h(32) | v_h(32)]

    const int tid = threadIdx.x, lane = tid & 63;
    // hw: this thread's head inside the workgroup (staging AND compute), as an SGPR: it selects channel offsets of buffer loads
    const int t255 = tid & 255, hw = __builtin_amdgcn_readfirstlane(tid >> 8);
    const int wave = (tid >> 6) & 3;                      // pixel quarter of the 256-pixel tile
    const int l31 = lane & 31, kgl = lane >> 5;
    // XCD-banded order with the four heads of a pixel slice adjacent: they read the same x tiles (L2 hits)
    constexpr int HG = 4 / HPW;                           // workgroups per pixel slice
    const int nsl = gridDim.x / (HG * a.B);
    const int wg = xcd_slot(blockIdx.x, gridDim.x);
    const int head = (wg % HG) * HPW + hw, slice = (wg / HG) % nsl, b = (wg / HG) / nsl;
    u32x4 *s_w = s_wall + hw * WHEAD;
    const int tile0 = slice * a.tps;
    const int tile1 = min(tile0 + a.tps, a.tiles);
    const AT *xb = reinterpret_cast<const AT *>(a.x) + (size_t)b * a.C * a.HW;
    const u32x4 *wblk = reinterpret_cast<const u32x4 *>(a.wkv) + (size_t)head * a.nstage * (2 * NKG * 64);

    float raw[NKGT][8];
    u32x4 wregs[KCH];
    // Buffer loads: the lane's pixel is the per-lane byte offset, the channel offset is an SGPR -> no VALU address
    // arithmetic.  Pixels >= HW get a per-lane offset equal to the descriptor size: the bounds check (which covers the
    // per-lane offset) returns 0 for them without touching memory.  Channels >= C (only when C % 32 != 0) are
    // clamped and zeroed by a select, since the scalar offset is not part of the check.
    const unsigned long long xaddr = reinterpret_cast<unsigned long long>(xb);
    const unsigned xlo = __builtin_amdgcn_readfirstlane((unsigned)xaddr);           // (unsigned: the builtin returns int and
    const unsigned xhi = __builtin_amdgcn_readfirstlane((unsigned)(xaddr >> 32));   //  would sign-extend into the high word)
    const int xbytes = a.C * a.HW * AB;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void *>(((unsigned long long)xhi << 32) | xlo), 0, __builtin_amdgcn_readfirstlane(xbytes), 0x00020000);
    [[maybe_unused]] float rawh[NKGT][8], tmask = 0.f;
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rsh = rsx, rsi = rsx;
    [[maybe_unused]] const bool tail_store = (wg % HG) == 0;          // one head group of a pixel slice writes the block's output
    if constexpr (TAIL) {
        auto mk = [&](const void *p) {
            const unsigned long long u = reinterpret_cast<unsigned long long>(reinterpret_cast<const AT *>(p) + (size_t)b * a.C * a.HW);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
            return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo), 0,
                                                     __builtin_amdgcn_readfirstlane(xbytes), 0x00020000);
        };
        rsh = mk(a.tail.h);
        rsi = mk(a.tail.xin);
    }
    // the fused tail of (tile, stage): raw <- the block's output for this thread's channels, stored by the slice's first head group
    auto apply_tail = [&](int tile, int stage) {
        if constexpr (TAIL) {
            const int n = tile * 256 + t255;
            const int voff = n < a.HW ? n * AB : xbytes;           // (stores beyond the sample are dropped by the bounds check)
#pragma unroll
            for (int kg = 0; kg < NKGT; ++kg) {
                const int cb = stage * 16 * KCH + (hw * NKGT + kg) * 8;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float sc = a.tail.esc[(size_t)b * a.C + cb + i], sh = a.tail.esh[(size_t)b * a.C + cb + i];
                    const float v = tail_value(rawh[kg][i], raw[kg][i], sc, sh, tmask);
                    raw[kg][i] = v;
                    if (tail_store) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rsx, voff, (cb + i) * a.HW * 4, GTTS_OUT_NT);
                }
            }
        }
    };
    auto load = [&](int tile, int stage) {
        const int n = tile * 256 + t255;
        const int voff = n < a.HW ? n * AB : xbytes;
        if constexpr (TAIL) {
            if (stage == 0) tmask = n < a.HW ? a.tail.mask[(size_t)b * a.tail.T + ((size_t)(n % a.tail.W) << a.tail.lvl)] : 0.f;
        }
#pragma unroll
        for (int kg = 0; kg < NKGT; ++kg) {
            const int cb = stage * 16 * KCH + (hw * NKGT + kg) * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = FULLC ? cb + i : min(cb + i, a.C - 1);
                float v;
                if constexpr (TAIL) {
                    v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsi, voff, c * a.HW * 4, 0));
                    rawh[kg][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsh, voff, c * a.HW * 4, 0));
                } else if constexpr (AB == 4) v = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsx, voff, c * a.HW * 4, 0));
                else v = __builtin_bit_cast(float, (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsx, voff, c * a.HW * 2, 0) << 16);
                raw[kg][i] = (FULLC || cb + i < a.C) ? v : 0.f;
            }
        }
        const u32x4 *g = wblk + (size_t)stage * (2 * NKG * 64);
#pragma unroll
        for (int j = 0; j < KCH; ++j) wregs[j] = g[t255 + j * 256];
    };

    const float NEG_INF = -__builtin_inff();
    float m_run = NEG_INF, z_run = 0.f;
    f32x16 ctx;
#pragma unroll
    for (int r = 0; r < 16; ++r) ctx[r] = 0.f;
    constexpr bool lo_on = NSPLIT > 1;

    load(tile0, 0);
    for (int tile = tile0; tile < tile1; ++tile) {
        f32x16 acc[2][2];     // [pixel fragment][0: k_h, 1: v_h]; D' layout: col = channel, rows = pixels
#pragma unroll
        for (int pf = 0; pf < 2; ++pf)
#pragma unroll
            for (int cf = 0; cf < 2; ++cf)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[pf][cf][r] = 0.f;

        for (int stage = 0; stage < a.nstage; ++stage) {
            apply_tail(tile, stage);
            lds_barrier();
#pragma unroll
            for (int kg = 0; kg < NKGT; ++kg) {
                u32x4 hi, lo;
                pack8_split(raw[kg], hi, lo);
                s_ah[(hw * NKGT + kg) * 256 + t255] = hi;
                s_al[(hw * NKGT + kg) * 256 + t255] = lo;
            }
#pragma unroll
            for (int j = 0; j < KCH; ++j) s_w[t255 + j * 256] = wregs[j];
            lds_barrier();
            if (stage + 1 < a.nstage) load(tile, stage + 1);
            else if (tile + 1 < tile1) load(tile + 1, 0);
#pragma unroll
            for (int kc = 0; kc < KCH; ++kc) {
                bf16x8 xh[2], xl[2], wh[2], wl[2];
#pragma unroll
                for (int pf = 0; pf < 2; ++pf) {
                    const int xi = (kc * 2 + kgl) * 256 + wave * 64 + pf * 32 + l31;
                    xh[pf] = *reinterpret_cast<const bf16x8 *>(&s_ah[xi]);
                    xl[pf] = *reinterpret_cast<const bf16x8 *>(&s_al[xi]);
                }
#pragma unroll
                for (int cf = 0; cf < 2; ++cf) {
                    const int wi = (kc * 2 + kgl) * 64 + cf * 32 + l31;
                    wh[cf] = *reinterpret_cast<const bf16x8 *>(&s_w[wi]);
                    wl[cf] = *reinterpret_cast<const bf16x8 *>(&s_w[wi + NKG * 64]);
                }
#pragma unroll
                for (int pf = 0; pf < 2; ++pf)
#pragma unroll
                    for (int cf = 0; cf < 2; ++cf) {
                        if (lo_on) {
                            acc[pf][cf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[pf], wh[cf], acc[pf][cf], 0, 0, 0);
                            acc[pf][cf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[pf], wl[cf], acc[pf][cf], 0, 0, 0);
                        }
                        acc[pf][cf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[pf], wh[cf], acc[pf][cf], 0, 0, 0);
                    }
            }
        }

        // ---- online softmax over this wave's 64 pixels: lane (l31, kgl) owns channel d = l31 (k) / e = l31 (v)
        const int nbase = tile * 256 + wave * 64 + 4 * kgl;
        const bool full = tile * 256 + 256 <= a.HW;          // workgroup-uniform: only the last tile of a sample is ragged
        float tmax = NEG_INF;
        if (full) {
#pragma unroll
            for (int pf = 0; pf < 2; ++pf)
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) tmax = fmaxf(tmax, acc[pf][0][rg]);
        } else {
#pragma unroll
            for (int pf = 0; pf < 2; ++pf)
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const int n = nbase + pf * 32 + (rg & 3) + 8 * (rg >> 2);
                    if (n < a.HW) tmax = fmaxf(tmax, acc[pf][0][rg]);
                }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = (m_run == NEG_INF) ? 0.f : __expf(m_run - m_new);
        const float msub = (m_new == NEG_INF) ? 0.f : m_new;
        // p = exp(k - m) as one fma + v_exp_f32 per element: exp2(k * log2(e) - m * log2(e))
        constexpr float LOG2E = 1.44269504088896340736f;
        const float ml2 = msub * LOG2E;
        float psum = 0.f;
        if (full) {
#pragma unroll
            for (int pf = 0; pf < 2; ++pf)
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[pf][0][rg], LOG2E, -ml2));
                    acc[pf][0][rg] = p;
                    psum += p;
                }
        } else {
#pragma unroll
            for (int pf = 0; pf < 2; ++pf)
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const int n = nbase + pf * 32 + (rg & 3) + 8 * (rg >> 2);
                    const float p = (n < a.HW) ? __builtin_amdgcn_exp2f(__builtin_fmaf(acc[pf][0][rg], LOG2E, -ml2)) : 0.f;
                    acc[pf][0][rg] = p;
                    psum += p;
                }
        }
        psum += __shfl_xor(psum, 32, 64);
        z_run = z_run * alpha + psum;
        m_run = m_new;
        // ctx rows are d = (rg&3) + 8*(rg>>2) + 4*kgl: fetch that channel's rescale factor from lane d.  Once the running
        // maxima have settled every alpha is exactly 1 and the 16 cross-lane fetches + multiplies are skipped (wave-uniform)
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0ull) {
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) {
                const int d = (rg & 3) + 8 * (rg >> 2) + 4 * kgl;
                ctx[rg] *= __shfl(alpha, d, 64);
            }
        }
        // ctx[d][e] += sum_n p[d,n] v[e,n]: A = P (i = d), B = V (n = e); the k-slot (lane>>5, j) maps to the
        // same pixel in both operands because both come from the same C/D register layout.
#pragma unroll
        for (int pf = 0; pf < 2; ++pf)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                float pv[8], vv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    pv[i] = acc[pf][0][hh * 8 + i];
                    vv[i] = acc[pf][1][hh * 8 + i];
                }
                u32x4 ph, pl, vh, vl;
                pack8_split(pv, ph, pl);
                pack8_split(vv, vh, vl);
                const bf16x8 Ph = *reinterpret_cast<bf16x8 *>(&ph), Pl = *reinterpret_cast<bf16x8 *>(&pl);
                const bf16x8 Vh = *reinterpret_cast<bf16x8 *>(&vh), Vl = *reinterpret_cast<bf16x8 *>(&vl);
                if (lo_on) {
                    ctx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Pl, Vh, ctx, 0, 0, 0);
                    ctx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ph, Vl, ctx, 0, 0, 0);
                }
                ctx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ph, Vh, ctx, 0, 0, 0);
            }
    }

    // ---- merge the four waves' partials (log-sum-exp, fixed order) -> one record per workgroup:
    // m[32], Z[32], ctx[32][32] (relative to m)
    __syncthreads();                                   // every wave is done with the staging buffer: it becomes the merge buffer
    float (*s_mrg)[ATTN_REC] = reinterpret_cast<float (*)[ATTN_REC]>(s_all) + hw * 4;      // this head's four records
    {
        float *mine = s_mrg[wave];
        if (kgl == 0) {
            mine[l31] = m_run;
            mine[32 + l31] = z_run;
        }
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) {
            const int d = (rg & 3) + 8 * (rg >> 2) + 4 * kgl;
            mine[64 + d * 32 + l31] = ctx[rg];
        }
    }
    __syncthreads();
    float *rec = a.partials + ((((size_t)b * 4 + head) * a.nrec) + (size_t)slice) * ATTN_REC;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int idx = t255 + 256 * k, d = idx >> 5;
        float M = NEG_INF;
#pragma unroll
        for (int w = 0; w < 4; ++w) M = fmaxf(M, s_mrg[w][d]);
        float acc_c = 0.f, acc_z = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            const float mw = s_mrg[w][d];
            const float sc = (mw == NEG_INF) ? 0.f : __expf(mw - M);
            acc_c = fmaf(s_mrg[w][64 + idx], sc, acc_c);
            acc_z = fmaf(s_mrg[w][32 + d], sc, acc_z);
        }
        rec[64 + idx] = acc_c;
        if ((idx & 31) == 0) {
            rec[d] = M;
            rec[32 + d] = acc_z;
        }
    }
}

// ------------------------------------------------------------------------------------------------ head-per-wave variant
// C == 64 (attn_head_per_wave): a workgroup owns a slice of 64-pixel tiles for ALL FOUR heads, wave w = head w.  The x
// tile (all 64 channels) is loaded, split into bf16 hi/lo and staged ONCE for the four heads (the per-head kernel above does
// that four times, in four workgroups), and the k|v projection weights of a head -- 16 KB -- stay resident in that wave's own
// LDS region for the whole workgroup, so nothing but the x image is shared: two workgroup barriers per tile (the per-head
// kernel: four per 256-pixel tile, i.e. sixteen per 4 x 64 pixels).  Per wave and tile: 48 projection MFMAs + the online
// softmax + 12 context MFMAs, exactly the unit of work of the per-head kernel; records go out per wave (= per head).
// TAIL: the ResnetBlock's identity tail rides in the staging step (round-5 review item 5: tail_identity wrote the tensor this kernel
// then read back -- two passes over 335 MB at level 0): the thread that stages (8 channels, pixel n) loads the block's input and the
// raw output of its second convolution instead, forms x = xin m + Mish(h sc + sh) m exactly as tail_identity_kernel does, stores it (the
// apply pass and the skip connection read it later) and stages it.  One read of the tensor less per attention block.
template <int NSPLIT, typename AT, int TAIL = 0>
__global__ __launch_bounds__(256, 2) void attn_ctx64_kernel(const AttnCtxArgs a) {
    constexpr int AB = (int)sizeof(AT);
    constexpr int C = 64, NKGT = C / 8;                         // 8 channel groups of 8
    constexpr bool lo_on = NSPLIT > 1;
    __shared__ __attribute__((aligned(16))) u32x4 s_ah[NKGT * 64];          // [kg][pixel] hi
    __shared__ __attribute__((aligned(16))) u32x4 s_al[NKGT * 64];          // [kg][pixel] lo
    __shared__ __attribute__((aligned(16))) u32x4 s_w[4][1024];             // per wave: [stage 2][split 2][kg 4][64 rows]

    const int tid = threadIdx.x, lane = tid & 63;
    // the wave index as an SGPR: it selects channel offsets of buffer loads (a VGPR there costs a waterfall loop per load)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kgl = lane >> 5;
    const int nsl = gridDim.x / a.B;
    const int wg = xcd_slot(blockIdx.x, gridDim.x);
    const int slice = wg % nsl, b = wg / nsl;
    const int head = wave;
    const int tile0 = slice * a.tps, tile1 = min(tile0 + a.tps, a.tiles);
    const AT *xb = reinterpret_cast<const AT *>(a.x) + (size_t)b * C * a.HW;

    // this head's packed projection weights: one contiguous 16 KB block, copied once
    {
        const u32x4 *wsrc = reinterpret_cast<const u32x4 *>(a.wkv) + (size_t)head * 1024;
        u32x4 wr[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) wr[i] = wsrc[lane + 64 * i];
#pragma unroll
        for (int i = 0; i < 16; ++i) s_w[wave][lane + 64 * i] = wr[i];
    }

    const unsigned long long xaddr = reinterpret_cast<unsigned long long>(xb);
    const unsigned xlo = __builtin_amdgcn_readfirstlane((unsigned)xaddr);
    const unsigned xhi = __builtin_amdgcn_readfirstlane((unsigned)(xaddr >> 32));
    const int xbytes = C * a.HW * AB;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void *>(((unsigned long long)xhi << 32) | xlo), 0, __builtin_amdgcn_readfirstlane(xbytes), 0x00020000);
    // staging items: thread -> (channel group kg = it * 4 + wave, pixel = lane); the channel offset is wave-uniform
    float raw[2][8];
    [[maybe_unused]] float rawh[2][8], tsc[2][8], tsh[2][8], tmask = 0.f;
    [[maybe_unused]] __amdgpu_buffer_rsrc_t rsh = rsx, rsi = rsx;
    if constexpr (TAIL) {
        static_assert(!TAIL || AB == 4, "the fused tail is the fp32-storage form");
        auto mk = [&](const void *p) {
            const unsigned long long u = reinterpret_cast<unsigned long long>(reinterpret_cast<const AT *>(p) + (size_t)b * C * a.HW);
            const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u), hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
            return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo), 0,
                                                     __builtin_amdgcn_readfirstlane(xbytes), 0x00020000);
        };
        rsh = mk(a.tail.h);
        rsi = mk(a.tail.xin);
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = (it * 4 + wave) * 8 + i;
                tsc[it][i] = a.tail.esc[(size_t)b * C + c];
                tsh[it][i] = a.tail.esh[(size_t)b * C + c];
            }
    }
    auto load = [&](int tile) {
        const int n = tile * 64 + lane;
        const int voff = n < a.HW ? n * AB : xbytes;               // beyond the sample: the bounds check returns 0
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int c = (it * 4 + wave) * 8 + i;
                if constexpr (TAIL) {
                    raw[it][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsi, voff, c * a.HW * 4, 0));
                    rawh[it][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsh, voff, c * a.HW * 4, 0));
                } else if constexpr (AB == 4) raw[it][i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsx, voff, c * a.HW * 4, 0));
                else raw[it][i] = __builtin_bit_cast(float, (unsigned)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(rsx, voff, c * a.HW * 2, 0) << 16);
            }
        if constexpr (TAIL) tmask = n < a.HW ? a.tail.mask[(size_t)b * a.tail.T + ((size_t)(n % a.tail.W) << a.tail.lvl)] : 0.f;
    };
    // the fused tail: turn the loaded (xin, h) pair of the tile into the block's output, store it, keep it for staging
    auto apply_tail = [&](int tile) {
        if constexpr (TAIL) {
            const int n = tile * 64 + lane;
            const int voff = n < a.HW ? n * AB : xbytes;           // (stores beyond the sample are dropped by the bounds check)
#pragma unroll
            for (int it = 0; it < 2; ++it)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int c = (it * 4 + wave) * 8 + i;
                    const float v = tail_value(rawh[it][i], raw[it][i], tsc[it][i], tsh[it][i], tmask);
                    raw[it][i] = v;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rsx, voff, c * a.HW * 4, GTTS_OUT_NT);
                }
        }
    };

    const float NEG_INF = -__builtin_inff();
    float m_run = NEG_INF, z_run = 0.f;
    f32x16 ctx;          // (two accumulators -- two dependent chains of 6 context MFMAs instead of one of 12 -- measured: no change)
#pragma unroll
    for (int r = 0; r < 16; ++r) ctx[r] = 0.f;

    load(tile0);
    for (int tile = tile0; tile < tile1; ++tile) {
        apply_tail(tile);
        lds_barrier();                                  // every head is done with the previous tile's image (first pass: s_w is written)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            u32x4 hi, lo;
            pack8_split(raw[it], hi, lo);
            s_ah[(it * 4 + wave) * 64 + lane] = hi;
            if (lo_on) s_al[(it * 4 + wave) * 64 + lane] = lo;
        }
        lds_barrier();
        load(min(tile + 1, tile1 - 1));                 // unconditional prefetch (exact waitcnt; the last one is never used)

        f32x16 acc[2][2];     // [pixel fragment][0: k_h, 1: v_h]; D layout: col = channel, rows = pixels
#pragma unroll
        for (int pf = 0; pf < 2; ++pf)
#pragma unroll
            for (int cf = 0; cf < 2; ++cf)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[pf][cf][r] = 0.f;
#pragma unroll
        for (int s = 0; s < 4; ++s) {                   // 16-channel k-steps
            bf16x8 xh[2], xl[2], wh[2], wl[2];
#pragma unroll
            for (int pf = 0; pf < 2; ++pf) {
                const int xi = (s * 2 + kgl) * 64 + pf * 32 + l31;
                xh[pf] = *reinterpret_cast<const bf16x8 *>(&s_ah[xi]);
                if (lo_on) xl[pf] = *reinterpret_cast<const bf16x8 *>(&s_al[xi]);
            }
#pragma unroll
            for (int cf = 0; cf < 2; ++cf) {
                const int wi = (s >> 1) * 512 + ((s & 1) * 2 + kgl) * 64 + cf * 32 + l31;
                wh[cf] = *reinterpret_cast<const bf16x8 *>(&s_w[wave][wi]);
                if (lo_on) wl[cf] = *reinterpret_cast<const bf16x8 *>(&s_w[wave][wi + 256]);
            }
#pragma unroll
            for (int pf = 0; pf < 2; ++pf)
#pragma unroll
                for (int cf = 0; cf < 2; ++cf) {
                    if (lo_on) {
                        acc[pf][cf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xl[pf], wh[cf], acc[pf][cf], 0, 0, 0);
                        acc[pf][cf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[pf], wl[cf], acc[pf][cf], 0, 0, 0);
                    }
                    acc[pf][cf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xh[pf], wh[cf], acc[pf][cf], 0, 0, 0);
                }
        }

        // ---- online softmax over the tile's 64 pixels: lane (l31, kgl) owns channel d = l31 (k) / e = l31 (v)
        const int nbase = tile * 64 + 4 * kgl;
        const bool full = tile * 64 + 64 <= a.HW;
        float tmax = NEG_INF;
        if (full) {
#pragma unroll
            for (int pf = 0; pf < 2; ++pf)
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) tmax = fmaxf(tmax, acc[pf][0][rg]);
        } else {
#pragma unroll
            for (int pf = 0; pf < 2; ++pf)
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const int n = nbase + pf * 32 + (rg & 3) + 8 * (rg >> 2);
                    if (n < a.HW) tmax = fmaxf(tmax, acc[pf][0][rg]);
                }
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float m_new = fmaxf(m_run, tmax);
        const float alpha = (m_run == NEG_INF) ? 0.f : __expf(m_run - m_new);
        const float msub = (m_new == NEG_INF) ? 0.f : m_new;
        constexpr float LOG2E = 1.44269504088896340736f;
        const float ml2 = msub * LOG2E;
        float psum = 0.f;
        if (full) {
#pragma unroll
            for (int pf = 0; pf < 2; ++pf)
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const float p = __builtin_amdgcn_exp2f(__builtin_fmaf(acc[pf][0][rg], LOG2E, -ml2));
                    acc[pf][0][rg] = p;
                    psum += p;
                }
        } else {
#pragma unroll
            for (int pf = 0; pf < 2; ++pf)
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    const int n = nbase + pf * 32 + (rg & 3) + 8 * (rg >> 2);
                    const float p = (n < a.HW) ? __builtin_amdgcn_exp2f(__builtin_fmaf(acc[pf][0][rg], LOG2E, -ml2)) : 0.f;
                    acc[pf][0][rg] = p;
                    psum += p;
                }
        }
        psum += __shfl_xor(psum, 32, 64);
        z_run = z_run * alpha + psum;
        m_run = m_new;
        if (__builtin_amdgcn_ballot_w64(alpha != 1.f) != 0ull) {
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) {
                const int d = (rg & 3) + 8 * (rg >> 2) + 4 * kgl;
                ctx[rg] *= __shfl(alpha, d, 64);
            }
        }
#pragma unroll
        for (int pf = 0; pf < 2; ++pf)
#pragma unroll
            for (int hh = 0; hh < 2; ++hh) {
                float pv[8], vv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    pv[i] = acc[pf][0][hh * 8 + i];
                    vv[i] = acc[pf][1][hh * 8 + i];
                }
                u32x4 ph, pl, vh, vl;
                pack8_split(pv, ph, pl);
                pack8_split(vv, vh, vl);
                const bf16x8 Ph = *reinterpret_cast<bf16x8 *>(&ph), Pl = *reinterpret_cast<bf16x8 *>(&pl);
                const bf16x8 Vh = *reinterpret_cast<bf16x8 *>(&vh), Vl = *reinterpret_cast<bf16x8 *>(&vl);
                if (lo_on) {
                    ctx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Pl, Vh, ctx, 0, 0, 0);
                    ctx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ph, Vl, ctx, 0, 0, 0);
                }
                ctx = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ph, Vh, ctx, 0, 0, 0);
            }
    }

    // ---- one record per wave (= per head): m[32], Z[32], ctx[32][32] relative to m
    float *rec = a.partials + ((((size_t)b * 4 + head) * a.nrec) + (size_t)slice) * ATTN_REC;
    if (kgl == 0) {
        rec[l31] = m_run;
        rec[32 + l31] = z_run;
    }
#pragma unroll
    for (int rg = 0; rg < 16; ++rg) {
        const int d = (rg & 3) + 8 * (rg >> 2) + 4 * kgl;
        rec[64 + d * 32 + l31] = ctx[rg];
    }
}

hipError_t launch_attn_ctx(const void *x, const unsigned char *wkv, float *partials, int B, int C, int HW, int nsplit,
                           hipStream_t st, int act_bf16, const AttnTail *tail) {
    AttnGeom g = attn_geom(HW, C);
    if ((size_t)C * HW * 4 >= ((size_t)1 << 31)) return hipErrorInvalidValue;     // 32-bit offsets in the buffer descriptor
    if (tail != nullptr && (C % 32 != 0 || act_bf16)) return hipErrorInvalidValue;
    AttnCtxArgs a;
    memset(&a, 0, sizeof(a));
    if (tail != nullptr) a.tail = *tail;
    a.x = x; a.wkv = wkv; a.partials = partials; a.C = C; a.HW = HW;
    a.nstage = (C + 16 * ATTN_KCH - 1) / (16 * ATTN_KCH);
    a.tiles = g.tiles; a.tps = g.tps; a.nrec = g.nrec; a.nsplit = nsplit; a.B = B;
    if (attn_head_per_wave(C)) {
        static_assert(ATTN_KCH == 2, "attn_ctx64_kernel indexes the packed k|v blocks as 32-channel stages");
        const dim3 grid64(g.nslices * B);
        if (act_bf16) {
            if (nsplit > 1) return hipErrorInvalidValue;
            hipLaunchKernelGGL((attn_ctx64_kernel<1, __bf16>), grid64, dim3(256), 0, st, a);
        } else if (tail != nullptr) {
            if (nsplit > 1) hipLaunchKernelGGL((attn_ctx64_kernel<2, float, 1>), grid64, dim3(256), 0, st, a);
            else hipLaunchKernelGGL((attn_ctx64_kernel<1, float, 1>), grid64, dim3(256), 0, st, a);
        } else if (nsplit > 1) hipLaunchKernelGGL((attn_ctx64_kernel<2, float>), grid64, dim3(256), 0, st, a);
        else hipLaunchKernelGGL((attn_ctx64_kernel<1, float>), grid64, dim3(256), 0, st, a);
        return hipGetLastError();
    }
    constexpr int HPW = GTTS_ATTN_HPW;
    const dim3 grid(g.nslices * (4 / HPW) * B), block(256 * HPW);
    if (act_bf16) {
        if (nsplit > 1) return hipErrorInvalidValue;
        if (C % 32 == 0) hipLaunchKernelGGL((attn_ctx_kernel<1, 1, __bf16, HPW>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((attn_ctx_kernel<1, 0, __bf16, HPW>), grid, block, 0, st, a);
        return hipGetLastError();
    }
    if (C % 32 == 0 && tail != nullptr) {
        if (nsplit > 1) hipLaunchKernelGGL((attn_ctx_kernel<2, 1, float, HPW, 1>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((attn_ctx_kernel<1, 1, float, HPW, 1>), grid, block, 0, st, a);
    } else if (C % 32 == 0) {
        if (nsplit > 1) hipLaunchKernelGGL((attn_ctx_kernel<2, 1, float, HPW>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((attn_ctx_kernel<1, 1, float, HPW>), grid, block, 0, st, a);
    } else {
        if (nsplit > 1) hipLaunchKernelGGL((attn_ctx_kernel<2, 0, float, HPW>), grid, block, 0, st, a);
        else hipLaunchKernelGGL((attn_ctx_kernel<1, 0, float, HPW>), grid, block, 0, st, a);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ attn_merge
// grid (4 heads, B), 1024 threads.  ctxn[b][h][d][e] = sum_i ctx_i[d][e] w_i[d] / sum_i Z_i[d] w_i[d],
// w_i[d] = exp(m_i[d] - M[d]), M[d] = max_i m_i[d].  Fixed reduction order => bit-reproducible.
__global__ __launch_bounds__(1024) void attn_merge_kernel(const float *__restrict__ partials, float *__restrict__ ctxn,
                                                           int nrec) {
    extern __shared__ float sm[];
    float *s_w = sm;                        // [nrec][32]
    float *s_red = sm + (size_t)nrec * 32;  // [32][32]
    float *s_M = s_red + 1024;              // [32]
    float *s_Zi = s_M + 32;                 // [32]
    const int head = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    const int d = tid & 31, g = tid >> 5;
    const float *rec = partials + ((size_t)b * 4 + head) * nrec * ATTN_REC;
    const float NEG_INF = -__builtin_inff();
    float mx = NEG_INF;
    for (int i = g; i < nrec; i += 32) mx = fmaxf(mx, rec[(size_t)i * ATTN_REC + d]);
    s_red[g * 32 + d] = mx;
    __syncthreads();
    if (g == 0) {
        float M = NEG_INF;
        for (int k = 0; k < 32; ++k) M = fmaxf(M, s_red[k * 32 + d]);
        s_M[d] = M;
    }
    __syncthreads();
    const float M = s_M[d];
    float zz = 0.f;
    for (int i = g; i < nrec; i += 32) {
        const float mi = rec[(size_t)i * ATTN_REC + d];
        const float w = (mi == NEG_INF) ? 0.f : expf(mi - M);
        s_w[i * 32 + d] = w;
        zz += rec[(size_t)i * ATTN_REC + 32 + d] * w;
    }
    __syncthreads();          // everyone is done reading the maxima in s_red
    s_red[g * 32 + d] = zz;
    __syncthreads();
    if (g == 0) {
        float Z = 0.f;
        for (int k = 0; k < 32; ++k) Z += s_red[k * 32 + d];
        s_Zi[d] = 1.0f / Z;
    }
    __syncthreads();
    const int dd = tid >> 5;                // row of ctx entry `tid` (entry = dd*32 + e)
    float acc = 0.f;
    const float *cp = rec + 64 + tid;
    int i = 0;
    for (; i + 4 <= nrec; i += 4) {
        const float c0 = cp[(size_t)(i + 0) * ATTN_REC], c1 = cp[(size_t)(i + 1) * ATTN_REC];
        const float c2 = cp[(size_t)(i + 2) * ATTN_REC], c3 = cp[(size_t)(i + 3) * ATTN_REC];
        acc = fmaf(c0, s_w[(i + 0) * 32 + dd], acc);
        acc = fmaf(c1, s_w[(i + 1) * 32 + dd], acc);
        acc = fmaf(c2, s_w[(i + 2) * 32 + dd], acc);
        acc = fmaf(c3, s_w[(i + 3) * 32 + dd], acc);
    }
    for (; i < nrec; ++i) acc = fmaf(cp[(size_t)i * ATTN_REC], s_w[i * 32 + dd], acc);
    ctxn[(((size_t)b * 4 + head) * 1024) + tid] = acc * s_Zi[dd];
}

hipError_t launch_attn_merge(const float *partials, float *ctxn, int B, int nrec, hipStream_t st) {
    const size_t smem = ((size_t)nrec * 32 + 1024 + 64) * sizeof(float);
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    if (smem > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&attn_merge_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(attn_merge_kernel, dim3(4, B), dim3(1024), smem, st, partials, ctxn, nrec);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ attn_fold
// M[co][ci] = g * sum_{h,d} U[co][h,d] Wq[h*32+d][ci],  U[co][h,d] = sum_e Wout[co][h*32+e] ctx_h[d][e]
// grid (C/8, B), 256 threads: a workgroup owns 8 output channels of one sample.  The 128-long (h,d) contraction is
// split over 256 / min(C,256) thread groups (short dependent chains, many loads in flight -- the kernel is pure
// latency) and combined through LDS in a fixed order.
constexpr int FOLD_CO = 8;
__global__ __launch_bounds__(256) void attn_fold_kernel(const float *__restrict__ ctxn, const float *__restrict__ wq,
                                                         const float *__restrict__ wout, const float *__restrict__ bout,
                                                         const float *__restrict__ g, unsigned char *__restrict__ wpk,
                                                         size_t wpk_bstride, float *__restrict__ biasb, int C, int MT,
                                                         int nkg) {
    __shared__ float s_U[FOLD_CO][128];
    __shared__ float s_part[4][FOLD_CO][128];     // partial sums of j-groups 1..3 (only used when C <= 128)
    const int co0 = blockIdx.x * FOLD_CO, b = blockIdx.y, tid = threadIdx.x;
    const float *cb = ctxn + (size_t)b * 4096;
    {   // U: 8 x 128 outputs, 4 per thread, 32-long dot products of contiguous rows
        const int j = tid & 127, h = j >> 5, d = j & 31;
        const float4 *cx = reinterpret_cast<const float4 *>(cb + (h * 32 + d) * 32);
        float4 cv[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) cv[e] = cx[e];
#pragma unroll
        for (int rr = 0; rr < FOLD_CO / 2; ++rr) {
            const int r = (tid >> 7) * (FOLD_CO / 2) + rr;
            const float4 *wo = reinterpret_cast<const float4 *>(wout + (size_t)(co0 + r) * 128 + h * 32);
            float acc = 0.f;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float4 w4 = wo[e];
                acc = fmaf(w4.x, cv[e].x, acc); acc = fmaf(w4.y, cv[e].y, acc);
                acc = fmaf(w4.z, cv[e].z, acc); acc = fmaf(w4.w, cv[e].w, acc);
            }
            s_U[r][j] = acc;
        }
    }
    __syncthreads();
    const float gv = g[0];
    const int ncot = (C + MT - 1) / MT;
    __bf16 *wp = reinterpret_cast<__bf16 *>(wpk + (size_t)b * wpk_bstride);
    const int Cc = C < 256 ? C : 256;          // input channels handled per pass
    const int nj = 256 / Cc >= 4 ? 4 : (256 / Cc >= 2 ? 2 : 1);   // j-groups (C = 64: 4, 128: 2, >= 256: 1)
    const int jq = tid / Cc, cl = tid - jq * Cc;
    const int jn = 128 / nj;                   // contraction length per group
    const bool active = jq < nj;               // C not dividing 256 leaves a few threads idle
    for (int ci = cl; ci < C; ci += Cc) {
        float acc[FOLD_CO];
#pragma unroll
        for (int r = 0; r < FOLD_CO; ++r) acc[r] = 0.f;
        const float *qp = wq + (size_t)(active ? jq * jn : 0) * C + ci;
        for (int j0 = 0; j0 < (active ? jn : 0); j0 += 8) {
            float q[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) q[u] = qp[(size_t)(j0 + u) * C];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int r = 0; r < FOLD_CO; ++r) acc[r] = fmaf(s_U[r][jq * jn + j0 + u], q[u], acc[r]);
        }
        if (nj > 1) {
            if (active && jq > 0) {
#pragma unroll
                for (int r = 0; r < FOLD_CO; ++r) s_part[jq][r][cl] = acc[r];
            }
            __syncthreads();                   // uniform: nj > 1 implies a single pass of the ci loop
            if (jq == 0) {
                for (int k = 1; k < nj; ++k)
#pragma unroll
                    for (int r = 0; r < FOLD_CO; ++r) acc[r] += s_part[k][r][cl];
            }
        }
        if (jq == 0) {
            const int chunk = ci / (8 * nkg), kg = (ci >> 3) % nkg, i = ci & 7;
#pragma unroll
            for (int r = 0; r < FOLD_CO; ++r) {
                const int co = co0 + r, cot = co / MT, m = co % MT;
                const size_t blk = (size_t)chunk * ncot + cot;             // CONV_P1: one stage, one tap
                const size_t e_hi = blk * ((size_t)MT * 16 * nkg) + ((size_t)(0 * nkg + kg) * MT + m) * 8 + i;
                const size_t e_lo = blk * ((size_t)MT * 16 * nkg) + ((size_t)(1 * nkg + kg) * MT + m) * 8 + i;
                __bf16 hi, lo;
                split_bf16(acc[r] * gv, hi, lo);
                wp[e_hi] = hi;
                wp[e_lo] = lo;
            }
        }
    }
    if (tid < FOLD_CO) biasb[(size_t)b * C + co0 + tid] = gv * bout[co0 + tid];
}

hipError_t launch_attn_fold(const float *ctxn, const float *wq, const float *wout, const float *bout, const float *g,
                            unsigned char *wpk, size_t wpk_bstride, float *biasb, int B, int C, hipStream_t st) {
    if (C % 16 != 0) return hipErrorInvalidValue;
    ConvGeom geom = conv_geom(CONV_P1, C, C);
    if (C % (16 * geom.kch) != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(attn_fold_kernel, dim3(C / FOLD_CO, B), dim3(256), 0, st, ctxn, wq, wout, bout, g, wpk,
                       wpk_bstride, biasb, C, geom.MT, 2 * geom.kch);
    return hipGetLastError();
}

}  // namespace gtts
