// conv1d.h -- 1-D implicit-GEMM convolution on the bf16 MFMA pipe (split-bf16 hi/lo, fp32 accumulate), shared by the
// HiFi-GAN generator (voc.hip) and the text / mel encoders (enc.hip).
//     D[row m][position q] = sum_{tap, cin} W[m][cin, tap] * pro(X[cin][q + off(tap)])
//   Conv1d(k, dilation d):        rows = cout, off(tap) = (tap - (k-1)/2) * d
//   ConvTranspose1d(k, stride s): rows = (cout, phase r), r = output position mod s; every phase is a 2-tap convolution
//       over the INPUT positions, the union over phases a 3-tap convolution with per-row zero weights.
// Prologue on load: LeakyReLU(slope) (slope 1 = identity, 0 = ReLU), then the [B][L] input mask.  Epilogue: bias, residual,
// running-sum modes of the HiFi-GAN ResBlocks, output mask.  Layout [B][C][L] fp32, L fastest (coalesced along time).
#pragma once
#include <atomic>
#include <type_traits>

#include "common.h"

namespace gtts {

constexpr int C1_MAXTAP = 12;
struct C1Args {
    const float *x;          // [B][cin][Lin]
    float *out;              // [B][cout][Lin * S]
    const float *res;        // residual added to the result (same indexing as out) or nullptr
    const float *accsrc;     // running sum over ResBlocks (accmode 1 / 2) or nullptr
    const unsigned char *w;  // packed [chunk][stage][cot][split][tap][kg][MT][8] bf16
    const float *bias;       // [cout]
    int B, cin, cout, Lin, S;
    int nchunk, nst;
    int toff[C1_MAXTAP];     // input offset of padded tap t (zero-weight pad taps use offset 0)
    int halo_lo, npx;        // -min(toff);  NT + halo_lo + max(toff)
    float slope;             // LeakyReLU slope applied to the input on load (1 = identity, 0 = ReLU)
    int accmode;             // 0 none, 1 v = accsrc + v, 2 v = (accsrc + v) / div
    float div;
    int ls;                  // log2(S) (S is a power of two)
    const float *in_mask;    // [B][Lin] multiplied into the input on load (x * x_mask), or nullptr
    const float *out_mask;   // [B][Lin * S] multiplied into the result, or nullptr
};

// Occupancy by tile: the 32- and 64-row tiles (HiFi-GAN's last two stages, the encoders' small layers) are bandwidth /
// latency bound -- a workgroup's whole life is a handful of dependent memory round trips -- and want many workgroups per CU;
// the 128-row tile is MFMA-bound and keeps the deeper (two chunks ahead) activation prefetch instead.
#ifndef GTTS_C1_WAVES128
#define GTTS_C1_WAVES128 2
#endif
// KCH = 2 (three-tap layers with cin % 32 == 0 on the 128-row tile): a step of the channel loop stages 32 input channels and the
// two packed 16-channel weight blocks that belong to them (the packing is unchanged), so the per-step costs that do not scale with
// the MFMA work -- two barriers per weight stage, the exposed part of the activation round trip -- are paid half as often.  A k = 3
// layer has only 36 MFMAs per wave and 16 channels: these layers were latency-bound (0.6 - 0.7 PFLOP/s executed at 2.2 TB/s).  The
// MFMAs are issued in the order of two consecutive 16-channel steps: results are bit-identical to KCH = 1.
template <int WM, int WN, int MF, int TPS, int AITER, int KCH = 1>
__global__ __launch_bounds__(256, (WM * MF == 1) ? 4 : (WM * MF == 2 ? 3 : GTTS_C1_WAVES128)) void conv1d_mfma_kernel(const C1Args a) {
    constexpr int MT = WM * MF * 32, NT = WN * 64, NKG = 2 * KCH;
    constexpr bool PF2 = MT >= 128;                             // activation prefetch distance 2 (else 1)
    constexpr int WBLK16 = 2 * TPS * 2 * MT;                   // 16-byte units per packed 16-channel weight stage (hi + lo)
    constexpr int WITER = (KCH * WBLK16 + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int NPX = a.npx;
    u32x4 *s_ah = reinterpret_cast<u32x4 *>(smem);             // [NKG][NPX]
    u32x4 *s_al = s_ah + NKG * NPX;
    u32x4 *s_w = s_al + NKG * NPX;                             // [KCH][split][tap][kg 2][MT]
    const int nsteps = a.nchunk / KCH;                          // (the launcher picks KCH = 2 only for an even number of 16-channel chunks)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kgl = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int M = a.cout * a.S;
    const int ncot = (M + MT - 1) / MT;
    const int ntile = (a.Lin + NT - 1) / NT;
    int wg = xcd_slot(blockIdx.x, gridDim.x);
    const int cot = wg % ncot; wg /= ncot;
    const int tile = wg % ntile;
    const int b = wg / ntile;
    const int q0 = tile * NT;
    const float *xb = a.x + (size_t)b * a.cin * a.Lin;

    // staging items: (8-channel group, halo position); geometry is chunk-invariant
    int it_pos[AITER];
    bool it_ok[AITER];
#pragma unroll
    for (int it = 0; it < AITER; ++it) {
        const int idx = tid + it * 256;
        const int p = idx % NPX;
        const int t = q0 - a.halo_lo + p;
        it_ok[it] = idx < NKG * NPX && t >= 0 && t < a.Lin;
        it_pos[it] = it_ok[it] ? t : 0;
    }
    float it_m[AITER];       // input mask at the item's position
#pragma unroll
    for (int it = 0; it < AITER; ++it) it_m[it] = a.in_mask ? a.in_mask[(size_t)b * a.Lin + it_pos[it]] : 1.f;
    // Activations are prefetched TWO chunks ahead into two statically indexed register sets: a 1-D convolution has only
    // `taps` MFMA groups per chunk (a third of the 3x3 kernel's), so one chunk of MFMAs does not cover an HBM round trip.
    float araw2[PF2 ? 2 : 1][AITER][8];
    // Buffer loads (the launcher requires cin % 16 == 0 -- every layer of the vocoder and the encoders -- and a sample below
    // 2 GB): per-lane byte offset of (first channel of the item's 8-group, position), out-of-range items point past the
    // descriptor and read 0, the chunk / channel offset rides in an SGPR -- no 64-bit VALU address arithmetic and no exec-mask
    // branch per load (plain pointer loads under a validity select compile to one s_cbranch_execz per element)
    const int xbytes = a.cin * a.Lin * 4;
    const unsigned long long xaddr = reinterpret_cast<unsigned long long>(xb);
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void *>(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(xaddr >> 32)) << 32) |
                                 (unsigned)__builtin_amdgcn_readfirstlane((unsigned)xaddr)),
        0, __builtin_amdgcn_readfirstlane(xbytes), 0x00020000);
    int it_voff[AITER];
#pragma unroll
    for (int it = 0; it < AITER; ++it) {
        const int idx = tid + it * 256;
        it_voff[it] = it_ok[it] ? (min(idx / NPX, NKG - 1) * 8 * a.Lin + it_pos[it]) * 4 : xbytes;
    }
    auto load_act = [&](int chunk, auto set_c) {
        constexpr int SET = PF2 ? decltype(set_c)::value : 0;
        float (&araw)[AITER][8] = araw2[SET];
#pragma unroll
        for (int it = 0; it < AITER; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned u = __builtin_amdgcn_raw_buffer_load_b32(rsx, it_voff[it], (chunk * 16 * KCH + i) * a.Lin * 4, 0);
                araw[it][i] = __builtin_bit_cast(float, u);
            }
    };
    u32x4 wregs[WITER];
    const u32x4 *wsrc = reinterpret_cast<const u32x4 *>(a.w);
    auto load_w = [&](int chunk, int stage) {
#pragma unroll
        for (int i = 0; i < WITER; ++i) {
            const int u = tid + i * 256;
            const int kc = KCH == 1 ? 0 : min(u / WBLK16, KCH - 1), ub = u - kc * WBLK16;      // packed block of 16-channel chunk chunk * KCH + kc
            const size_t blk = ((size_t)(chunk * KCH + kc) * a.nst + stage) * ncot + cot;
            wregs[i] = wsrc[blk * WBLK16 + (ub < WBLK16 ? ub : 0)];
        }
    };

    f32x16 acc[MF][2];
#pragma unroll
    for (int mi = 0; mi < MF; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    load_w(0, 0);
    load_act(0, std::integral_constant<int, 0>{});
    if (PF2 && nsteps > 1) load_act(1, std::integral_constant<int, 1>{});
    const int m0 = wm * MF * 32;
    auto do_chunk = [&](int chunk, auto set_c) {
        constexpr int SET = PF2 ? decltype(set_c)::value : 0;
        float (&araw)[AITER][8] = araw2[SET];
        lds_barrier();                      // previous chunk's MFMAs are done with the images
#pragma unroll
        for (int it = 0; it < AITER; ++it) {
            const int idx = tid + it * 256;
            bf16x8 vh, vl;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v = araw[it][i];
                v = v > 0.f ? v : v * a.slope;                    // F.leaky_relu(x, slope)
                v *= it_m[it];                                    // x * x_mask
                __bf16 h, l;
                split_bf16(v, h, l);
                vh[i] = h;
                vl[i] = l;
            }
            if (idx < NKG * NPX) {
                s_ah[idx] = *reinterpret_cast<u32x4 *>(&vh);
                s_al[idx] = *reinterpret_cast<u32x4 *>(&vl);
            }
        }
        if (chunk + (PF2 ? 2 : 1) < nsteps) load_act(chunk + (PF2 ? 2 : 1), set_c);     // the set just consumed is free again
        for (int stage = 0; stage < a.nst; ++stage) {
            if (stage > 0) lds_barrier();                         // previous stage's MFMAs are done with s_w
#pragma unroll
            for (int i = 0; i < WITER; ++i) {
                const int u = tid + i * 256;
                if (u < KCH * WBLK16) s_w[u] = wregs[i];
            }
            lds_barrier();
            if (stage + 1 < a.nst) load_w(chunk, stage + 1);
            else if (chunk + 1 < nsteps) load_w(chunk + 1, 0);
#pragma unroll
            for (int kc = 0; kc < KCH; ++kc)
#pragma unroll
            for (int j = 0; j < TPS; ++j) {
                const int off = a.halo_lo + a.toff[stage * TPS + j] + wn * 64 + l31;
                bf16x8 wh[MF], wl[MF], xh[2], xl[2];
#pragma unroll
                for (int mi = 0; mi < MF; ++mi) {
                    const int wi = kc * WBLK16 + (j * 2 + kgl) * MT + m0 + mi * 32 + l31;
                    wh[mi] = *reinterpret_cast<const bf16x8 *>(&s_w[wi]);
                    wl[mi] = *reinterpret_cast<const bf16x8 *>(&s_w[wi + TPS * 2 * MT]);
                }
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int xi = (kc * 2 + kgl) * NPX + off + ni * 32;
                    xh[ni] = *reinterpret_cast<const bf16x8 *>(&s_ah[xi]);
                    xl[ni] = *reinterpret_cast<const bf16x8 *>(&s_al[xi]);
                }
#pragma unroll
                for (int mi = 0; mi < MF; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[mi], xh[ni], acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[mi], xl[ni], acc[mi][ni], 0, 0, 0);
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[mi], xh[ni], acc[mi][ni], 0, 0, 0);
                    }
            }
        }
    };
    for (int chunk = 0; chunk < nsteps; chunk += 2) {
        do_chunk(chunk, std::integral_constant<int, 0>{});
        if (chunk + 1 < nsteps) do_chunk(chunk + 1, std::integral_constant<int, 1>{});
    }

    // ---- epilogue: bias, ResBlock residual, running sum over ResBlocks (reference operation order, fp32).
    // S is a power of two (a.ls = log2 S): row m -> (channel m >> ls, output phase m & (S-1)); 32-bit offsets inside a sample.
    const int Lout = a.Lin << a.ls;
    const size_t ob = (size_t)b * a.cout * Lout;
    float bv[MF][16];
#pragma unroll
    for (int mi = 0; mi < MF; ++mi)
#pragma unroll
        for (int rg = 0; rg < 16; ++rg) {
            const int m = cot * MT + m0 + mi * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * kgl;
            bv[mi][rg] = a.bias[min(m, M - 1) >> a.ls];
        }
    if (a.S == 1 && M % MT == 0) {
        // Plain convolutions on whole row tiles (every ResBlock / encoder layer): residual, running sum and output go through
        // buffer descriptors of this sample's tensors -- per-lane byte offset = (the lane's 4-row sub-block, position), the row
        // offset of each of the 16 x MF values in an SGPR -- so the 64 (x 3 arrays) accesses of a lane need no 64-bit VALU address
        // arithmetic (with plain pointers the residual convolutions of a ResBlock pair took 1.5 - 2x the time of the first ones:
        // profiles/r04_hifigan_layers.txt); positions past the end get an out-of-range offset (loads return 0, stores are dropped).
        const int tbytes = a.cout * Lout * 4;
        auto rsrc = [&](const float *p) {
            const unsigned long long u = reinterpret_cast<unsigned long long>(p);
            return __builtin_amdgcn_make_buffer_rsrc(
                reinterpret_cast<void *>(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(u >> 32)) << 32) |
                                         (unsigned)__builtin_amdgcn_readfirstlane((unsigned)u)),
                0, __builtin_amdgcn_readfirstlane(tbytes), 0x00020000);
        };
        const __amdgpu_buffer_rsrc_t rs_out = rsrc(a.out + ob);
        const __amdgpu_buffer_rsrc_t rs_res = rsrc(a.res ? a.res + ob : a.out + ob);
        const __amdgpu_buffer_rsrc_t rs_acc = rsrc(a.accsrc ? a.accsrc + ob : a.out + ob);
        const int row0 = __builtin_amdgcn_readfirstlane(cot * MT + m0);
        const float *omb = a.out_mask ? a.out_mask + (size_t)b * Lout : nullptr;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int q = q0 + wn * 64 + ni * 32 + l31;
            const int voff = q < a.Lin ? (4 * kgl * Lout + q) * 4 : tbytes;       // (>= num_records: out of range whatever the row offset)
            const float om = (omb && q < a.Lin) ? omb[q] : 1.f;
#pragma unroll
            for (int mi = 0; mi < MF; ++mi) {
                float rv[16], av[16];
                if (a.res) {
#pragma unroll
                    for (int rg = 0; rg < 16; ++rg)
                        rv[rg] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                     rs_res, voff, (row0 + mi * 32 + (rg & 3) + 8 * (rg >> 2)) * Lout * 4, 0));
                }
                if (a.accmode != 0) {
#pragma unroll
                    for (int rg = 0; rg < 16; ++rg)
                        av[rg] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
                                     rs_acc, voff, (row0 + mi * 32 + (rg & 3) + 8 * (rg >> 2)) * Lout * 4, 0));
                }
#pragma unroll
                for (int rg = 0; rg < 16; ++rg) {
                    float v = acc[mi][ni][rg] + bv[mi][rg];
                    if (a.res) v = v + rv[rg];
                    if (a.accmode == 1) v = av[rg] + v;
                    else if (a.accmode == 2) v = __fdiv_rn(av[rg] + v, a.div);
                    if (omb) v *= om;
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rs_out, voff,
                                                          (row0 + mi * 32 + (rg & 3) + 8 * (rg >> 2)) * Lout * 4, 0);
                }
            }
        }
        return;
    }
    if (a.S >= 2 && M % MT == 0 && !a.res && a.accmode == 0 && !a.out_mask) {
        // ConvTranspose1d (every upsampling layer): the four consecutive rows (rg & 3) of a C/D fragment are four consecutive
        // output phases of one channel (S >= 4) or two phases of two adjacent channels (S == 2): one 16-byte / two 8-byte
        // buffer stores per lane instead of four scalar stores with 64-bit address arithmetic each.
        const int tbytes = a.cout * Lout * 4;
        const unsigned long long u = reinterpret_cast<unsigned long long>(a.out + ob);
        const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<void *>(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(u >> 32)) << 32) |
                                     (unsigned)__builtin_amdgcn_readfirstlane((unsigned)u)),
            0, __builtin_amdgcn_readfirstlane(tbytes), 0x00020000);
        typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
#pragma unroll
        for (int ni = 0; ni < 2; ++ni) {
            const int q = q0 + wn * 64 + ni * 32 + l31;
            const bool ok = q < a.Lin;
            const int qs = q << a.ls;
#pragma unroll
            for (int mi = 0; mi < MF; ++mi)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int m4 = cot * MT + m0 + mi * 32 + 8 * g + 4 * kgl;          // first of the lane's four rows
                    float v[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = acc[mi][ni][4 * g + i] + bv[mi][4 * g + i];
                    if (a.ls >= 2) {
                        const int co = m4 >> a.ls, r0 = m4 & (a.S - 1);
                        const int voff = ok ? (co * Lout + qs + r0) * 4 : tbytes;
                        u32x4 pk;
#pragma unroll
                        for (int i = 0; i < 4; ++i) pk[i] = __builtin_bit_cast(unsigned, v[i]);
                        __builtin_amdgcn_raw_buffer_store_b128(pk, rs_out, voff, 0, 0);
                    } else {
                        const int co = m4 >> 1;
                        const int voff = ok ? (co * Lout + qs) * 4 : tbytes;
                        u32x2 p0, p1;
                        p0[0] = __builtin_bit_cast(unsigned, v[0]); p0[1] = __builtin_bit_cast(unsigned, v[1]);
                        p1[0] = __builtin_bit_cast(unsigned, v[2]); p1[1] = __builtin_bit_cast(unsigned, v[3]);
                        __builtin_amdgcn_raw_buffer_store_b64(p0, rs_out, voff, 0, 0);
                        __builtin_amdgcn_raw_buffer_store_b64(p1, rs_out, ok ? voff + Lout * 4 : tbytes, 0, 0);
                    }
                }
        }
        return;
    }
    float *outb = a.out + ob;
    const float *resb = a.res ? a.res + ob : nullptr;
    const float *accb = a.accsrc ? a.accsrc + ob : nullptr;
    const float *omb = a.out_mask ? a.out_mask + (size_t)b * Lout : nullptr;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int q = q0 + wn * 64 + ni * 32 + l31;
        if (q >= a.Lin) continue;
        const int qs = q << a.ls;
#pragma unroll
        for (int mi = 0; mi < MF; ++mi)
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) {
                const int m = cot * MT + m0 + mi * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * kgl;
                if (m >= M) continue;
                const int co = m >> a.ls, r = m & (a.S - 1);
                const int idx = co * Lout + qs + r;
                float v = acc[mi][ni][rg] + bv[mi][rg];
                if (resb) v = v + resb[idx];
                if (a.accmode == 1) v = accb[idx] + v;
                else if (a.accmode == 2) v = __fdiv_rn(accb[idx] + v, a.div);
                if (omb) v *= omb[qs + r];
                outb[idx] = v;
            }
    }
}

template <int WM, int WN, int MF, int TPS, int AITER, int KCH = 1>
static hipError_t launch_c1_cfg(const C1Args &a, hipStream_t st) {
    constexpr int MT = WM * MF * 32, NT = WN * 64;
    const int M = a.cout * a.S;
    const int ncot = (M + MT - 1) / MT, ntile = (a.Lin + NT - 1) / NT;
    const size_t smem = (size_t)2 * 2 * KCH * a.npx * 16 + (size_t)KCH * 2 * TPS * 2 * MT * 16;
    if ((size_t)2 * KCH * a.npx > (size_t)AITER * 256 || a.nchunk % KCH != 0) return hipErrorInvalidValue;
    // the attribute is per device and sticky: raise it once per (instance, device) to the largest image any layer needs
    static std::atomic<unsigned long long> attr_done{0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    constexpr size_t SMEM_MAX = (size_t)2 * 2 * (AITER * 128) * 16 + (size_t)KCH * 2 * TPS * 2 * MT * 16;
    if (smem > SMEM_MAX) return hipErrorInvalidValue;
    if (!((attr_done.load(std::memory_order_relaxed) >> dev) & 1ull)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv1d_mfma_kernel<WM, WN, MF, TPS, AITER, KCH>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)SMEM_MAX);
        if (e != hipSuccess) return e;
        attr_done.fetch_or(1ull << dev, std::memory_order_relaxed);
    }
    hipLaunchKernelGGL((conv1d_mfma_kernel<WM, WN, MF, TPS, AITER, KCH>), dim3((unsigned)(ncot * ntile * a.B)), dim3(256), smem, st, a);
    return hipGetLastError();
}

// tiling of a layer by its row count M = cout * S (must agree with the packer below)
struct C1Geom { int MT, NT, tps; };
static C1Geom c1_geom(int M, int ntap_real) {
    C1Geom g;
    g.MT = M >= 128 ? 128 : (M >= 64 ? 64 : 32);
    g.NT = M >= 128 ? 128 : 256;
    g.tps = ntap_real <= 3 ? 3 : 4;
    return g;
}

template <int TPS>
static hipError_t launch_c1_t(const C1Args &a, hipStream_t st) {
    const int M = a.cout * a.S;
    const C1Geom g = c1_geom(M, TPS == 3 ? 3 : 4);
    const int aiter = (2 * a.npx + 255) / 256;
    if constexpr (TPS == 3) {
        // three-tap layers on the 128-row tile: 32 channels per step when the channel count allows it (see the kernel comment)
        if (g.MT == 128 && a.nchunk % 2 == 0 && a.cin % 32 == 0) {
            const int aiter2 = (4 * a.npx + 255) / 256;
            if (aiter2 <= 3) return launch_c1_cfg<2, 2, 2, 3, 3, 2>(a, st);
        }
    }
    if (g.MT == 128) {
        if (aiter <= 2) return launch_c1_cfg<2, 2, 2, TPS, 2>(a, st);
        if (aiter == 3) return launch_c1_cfg<2, 2, 2, TPS, 3>(a, st);
    } else if (g.MT == 64) {
        if (aiter <= 3) return launch_c1_cfg<1, 4, 2, TPS, 3>(a, st);
    } else {
        if (aiter <= 3) return launch_c1_cfg<1, 4, 1, TPS, 3>(a, st);
    }
    return hipErrorInvalidValue;
}

// ---- weight packer: reference layouts -> [chunk][stage][cot][split][tap][kg][MT][8] bf16 (hi, lo)
//   mode 0: Conv1d weight [cout][cin][K]            rows m = co,        padded tap t < K: weight[co][ci][t]
//   mode 1: ConvTranspose1d weight [cin][cout][Kt]  rows m = co*S + r,  tap t -> input offset d = t - 1:
//           a = r + pad; j = a / S - d; k = a % S + S * j; weight[ci][co][k] if 0 <= j < Kt / S else 0
static __global__ void pack_conv1d_kernel(const float *__restrict__ w, __bf16 *__restrict__ dst, int mode, int cin, int cout, int K,
                                   int S, int pad, int MT, int nst, int tps, int nchunk, int ncot, size_t total) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    size_t rr = t;
    const int i = rr % 8; rr /= 8;
    const int m = rr % MT; rr /= MT;
    const int kg = rr % 2; rr /= 2;
    const int tj = rr % tps; rr /= tps;
    const int cot = rr % ncot; rr /= ncot;
    const int stage = rr % nst; rr /= nst;
    const int chunk = (int)rr;
    const int ci = chunk * 16 + kg * 8 + i;
    const int row = cot * MT + m;
    const int tap = stage * tps + tj;
    float v = 0.f;
    if (ci < cin && row < cout * S) {
        if (mode == 0) {
            if (tap < K) v = w[((size_t)row * cin + ci) * K + tap];
        } else {
            const int co = row / S, r = row % S;
            const int d = tap - 1;
            if (tap < 3) {
                const int a = r + pad;
                const int j = a / S - d;
                if (j >= 0 && j < K / S) v = w[((size_t)ci * cout + co) * K + (a % S) + S * j];
            }
        }
    }
    __bf16 hi, lo;
    split_bf16(v, hi, lo);
    const size_t blk = ((size_t)chunk * nst + stage) * ncot + cot;
    const size_t blk_elems = (size_t)2 * tps * 2 * MT * 8;
    dst[blk * blk_elems + (((size_t)(0 * tps + tj) * 2 + kg) * MT + m) * 8 + i] = hi;
    dst[blk * blk_elems + (((size_t)(1 * tps + tj) * 2 + kg) * MT + m) * 8 + i] = lo;
}

// pack one layer's weights (device pointers) into its blob slot
static inline hipError_t launch_pack_conv1d(const float *w, unsigned char *dst, int mode, int cin, int cout, int K, int S, int pad,
                                            hipStream_t st) {
    const int real = mode == 0 ? K : 3;
    const C1Geom g = c1_geom(cout * S, real);
    const int nst = (real + g.tps - 1) / g.tps;
    const int nchunk = (cin + 15) / 16, ncot = (cout * S + g.MT - 1) / g.MT;
    const size_t total = (size_t)nchunk * nst * ncot * g.tps * 2 * g.MT * 8;
    hipLaunchKernelGGL(pack_conv1d_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, w,
                       reinterpret_cast<__bf16 *>(dst), mode, cin, cout, K, S, pad, g.MT, nst, g.tps, nchunk, ncot, total);
    return hipGetLastError();
}
static inline size_t conv1d_packed_bytes(int mode, int cin, int cout, int K, int S) {
    const int real = mode == 0 ? K : 3;
    const C1Geom g = c1_geom(cout * S, real);
    const size_t nst = (real + g.tps - 1) / g.tps;
    const size_t nchunk = (cin + 15) / 16, ncot = ((size_t)cout * S + g.MT - 1) / g.MT;
    return nchunk * nst * ncot * (size_t)2 * g.tps * 2 * g.MT * 16;
}

// fill the geometry fields of a[] for a Conv1d (mode 0: kernel K, dilation dil) or a ConvTranspose1d (mode 1) and launch
static inline hipError_t launch_conv1d(C1Args a, int mode, int K, int dil, hipStream_t st) {
    const int real = mode == 0 ? K : 3;
    if (real > C1_MAXTAP) return hipErrorInvalidValue;
    const C1Geom g = c1_geom(a.cout * a.S, real);
    a.nchunk = (a.cin + 15) / 16;
    a.nst = (real + g.tps - 1) / g.tps;
    int lo = 0, hi = 0;
    for (int t = 0; t < C1_MAXTAP; ++t) a.toff[t] = 0;
    for (int t = 0; t < real; ++t) {
        a.toff[t] = mode == 0 ? (t - (K - 1) / 2) * dil : t - 1;
        lo = a.toff[t] < lo ? a.toff[t] : lo;
        hi = a.toff[t] > hi ? a.toff[t] : hi;
    }
    a.halo_lo = -lo;
    a.npx = g.NT + hi - lo;
    a.ls = 0;
    while ((1 << a.ls) < a.S) ++a.ls;
    if ((1 << a.ls) != a.S) return hipErrorInvalidValue;
    // (the buffer-descriptor epilogues address a sample's OUTPUT with 32-bit byte offsets: cout * Lout * 4 bytes, Lout = Lin * S)
    if ((size_t)a.cout * a.Lin * a.S * 4 >= ((size_t)1 << 31)) return hipErrorInvalidValue;
    if (a.cin % 16 != 0 || (size_t)a.cin * a.Lin * 4 >= ((size_t)1 << 31)) return hipErrorInvalidValue;   // buffer-load staging
    return g.tps == 3 ? launch_c1_t<3>(a, st) : launch_c1_t<4>(a, st);
}

}  // namespace gtts
