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
// with cin walked in chunks of 16*KCH (K of v_mfma_f32_32x32x16_bf16 is 16; KCH k-steps per chunk).  Per chunk the workgroup stages the halo
// tile of 16 input channels into LDS *through registers*, applying the producer's epilogue on the way
// (GroupNorm affine + Mish + mask + time bias: "apply-on-load", so normalised tensors never touch HBM) and
// splitting fp32 into bf16 hi/lo.  LDS image: [kgroup(2*KCH)][pixel][8 channels] x {hi, lo}: one 16-byte slot per
// (pixel, 8-channel group) -> both the staging ds_write_b128 and the MFMA B-fragment ds_read_b128 are
// conflict-free, and a tap is just a pixel offset.  Weights are pre-packed on the device (pack.hip) in
// exactly the LDS image order, one contiguous block per (chunk, stage, cout tile).
//
// Precision: nsplit == 2 computes hi*hi + hi*lo + lo*hi with fp32 accumulation (error ~2^-17 per product,
// i.e. fp32-grade: SURVEY.md section 0); nsplit == 1 is plain bf16.
//
// Wave tile: (MF x 32) output channels x (2 rows x 32 columns) pixels; a workgroup is WM x WN waves.
#include "common.h"

namespace gtts {

template <int MODE, int WM, int WN, int MF, int KCH>
struct ConvCfg {
    static constexpr int MT = WM * MF * 32;
    static constexpr int TR = WN * 2;
    static constexpr int TC = 32;
    static constexpr int NST = MODE == CONV_P1 ? 1 : (MODE == CONV_UP ? 2 : 3);
    static constexpr int TPS = NST;
    static constexpr int HR = MODE == CONV_P1 ? TR : (MODE == CONV_DN ? 2 * TR + 1 : TR + 2);
    static constexpr int HC = MODE == CONV_P1 ? TC : (MODE == CONV_DN ? 2 * TC + 1 : TC + 2);
    static constexpr int NPIX = HR * HC;
    static constexpr int NKG = 2 * KCH;                     // 8-channel groups per chunk (chunk = 16*KCH channels)
    static constexpr int AITER = (NKG * NPIX + 255) / 256;  // (pixel, kgroup) staging items per thread
    static constexpr int WBLK16 = TPS * MT * 2 * NKG;       // 16-byte units per weight block
    static constexpr int WITER = WBLK16 / 256;
    static_assert(WBLK16 % 256 == 0, "weight block must be a whole number of 256 x 16-byte rows");
};

static inline size_t conv_smem_bytes(int npix, int nkg, int wblk16, int cin, int pro) {
    size_t cpad = (size_t)((cin + 8 * nkg - 1) / (8 * nkg)) * 8 * nkg;
    return (size_t)npix * nkg * 16 * 2 + (size_t)wblk16 * 16 + (pro == PRO_GN ? 3 * cpad * 4 : 0) + 4 * 2 * 4 * 2 * 4;
}

template <int MODE, int WM, int WN, int MF, int KCH>
__global__ __launch_bounds__(256, (MODE == CONV_P1 && !(WM == 1 && KCH == 2)) ? 3 : 2) void conv_mfma_kernel(const ConvArgs a) {
    using C = ConvCfg<MODE, WM, WN, MF, KCH>;
    constexpr int MT = C::MT, TR = C::TR, TC = C::TC, NST = C::NST, TPS = C::TPS, NKG = C::NKG;
    constexpr int HC = C::HC, NPIX = C::NPIX, AITER = C::AITER, WBLK16 = C::WBLK16, WITER = C::WITER;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4 *s_ah = reinterpret_cast<u32x4 *>(smem);      // [NKG][NPIX]  hi
    u32x4 *s_al = s_ah + NKG * NPIX;                    // [NKG][NPIX]  lo
    u32x4 *s_w = s_al + NKG * NPIX;                     // [split][tap][kg][MT]
    const int cpad = a.nchunk * 8 * NKG;
    float *s_par = reinterpret_cast<float *>(s_w + WBLK16);   // [3][cpad]: scale, shift, time bias
    float *s_red = s_par + (a.pro == PRO_GN ? 3 * cpad : 0);  // [4 waves][MF][4 slots][2]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, kg_l = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;
    const int b = blockIdx.z;
    const int tx = blockIdx.x % a.tiles_x, ty = blockIdx.x / a.tiles_x;
    int cot = blockIdx.y, phase = 0;
    if (MODE == CONV_UP) { phase = cot & 3; cot >>= 2; }
    const int ncot = (a.cout + MT - 1) / MT;
    const int ph_y = phase >> 1, ph_x = phase & 1;
    const int y0 = ty * TR, x0 = tx * TC;
    const int iy0 = MODE == CONV_P1 ? y0 : (MODE == CONV_DN ? 2 * y0 - 1 : y0 - 1);
    const int ix0 = MODE == CONV_P1 ? x0 : (MODE == CONV_DN ? 2 * x0 - 1 : x0 - 1);
    const int HWin = a.Hin * a.Win;

    // ---- per-(sample, channel) prologue parameters -> LDS (visible after the first barrier)
    if (a.pro == PRO_GN) {
        for (int i = tid; i < cpad; i += 256) {
            bool ok = i < a.cin;
            s_par[i] = ok ? a.sc[(size_t)b * a.cin + i] : 0.f;
            s_par[cpad + i] = ok ? a.sh[(size_t)b * a.cin + i] : 0.f;
            s_par[2 * cpad + i] = ok ? a.tb[(size_t)b * a.tb_stride + i] : 0.f;
        }
    }

    // ---- staging items: geometry is chunk-invariant
    int it_goff[AITER];     // offset inside a channel plane, -1 = outside the image / no item
    int it_lds[AITER];      // destination slot in s_ah / s_al
    int it_kg8[AITER];      // first channel of the item inside the chunk (kgroup * 8)
    float it_m[AITER];      // mask value at the item's frame (0 outside the image)
#pragma unroll
    for (int it = 0; it < AITER; ++it) {
        int idx = tid + it * 256;
        bool has = idx < NKG * NPIX;
        int kg = idx / NPIX;
        int p = idx - kg * NPIX;
        int pr = p / HC, pc = p - pr * HC;
        int gy = iy0 + pr, gx = ix0 + pc;
        bool in = has && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
        it_goff[it] = in ? gy * a.Win + gx : -1;
        int lc = pc;
        if (MODE == CONV_DN) lc = (pc & 1) ? 33 + (pc >> 1) : (pc >> 1);   // column-parity planes
        it_lds[it] = has ? kg * NPIX + pr * HC + lc : -1;
        it_kg8[it] = kg * 8;
        float m = 0.f;
        if (in) m = (a.pro == PRO_PLAIN) ? 1.f : a.mask[(size_t)b * a.T + ((size_t)gx << a.lvl_in)];
        it_m[it] = m;
    }

    float araw[AITER][8];
    auto load_act = [&](int chunk) {
#pragma unroll
        for (int it = 0; it < AITER; ++it) {
            int cbase = chunk * (8 * NKG) + it_kg8[it];
            const float *src;
            int cloc, ctot;
            if (cbase < a.c0) { src = a.src0; cloc = cbase; ctot = a.c0; }
            else { src = a.src1; cloc = cbase - a.c0; ctot = a.c1; }
            const float *pl = src + ((size_t)b * ctot + cloc) * HWin + (it_goff[it] < 0 ? 0 : it_goff[it]);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                bool ok = it_goff[it] >= 0 && (cbase + i) < a.cin;
                araw[it][i] = ok ? pl[(size_t)i * HWin] : 0.f;
            }
        }
    };

    const unsigned char *wbase = a.w + (size_t)b * a.w_bstride;
    u32x4 wregs[WITER];
    auto load_w = [&](int chunk, int stage) {
        size_t blk = (((size_t)phase * a.nchunk + chunk) * NST + stage) * ncot + cot;
        const u32x4 *g = reinterpret_cast<const u32x4 *>(wbase + blk * (size_t)(WBLK16 * 16));
#pragma unroll
        for (int i = 0; i < WITER; ++i) wregs[i] = g[tid + i * 256];
    };

    f32x16 acc[MF][2];
#pragma unroll
    for (int mi = 0; mi < MF; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

    load_w(0, 0);
    load_act(0);

    const int m0 = wm * MF * 32;
    const bool lo_on = a.nsplit > 1;

    for (int chunk = 0; chunk < a.nchunk; ++chunk) {
        __syncthreads();   // previous chunk's MFMAs are done with s_a* / s_w (and s_par is written)
        // ---- transform + split + stage the activation tile of this chunk
#pragma unroll
        for (int it = 0; it < AITER; ++it) {
            if (it_lds[it] >= 0) {
                const float m = it_m[it];
                const int cb = chunk * (8 * NKG) + it_kg8[it];
                bf16x8 vh, vl;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    float v = araw[it][i];
                    if (a.pro == PRO_MASK) {
                        v *= m;
                    } else if (a.pro == PRO_GN) {
                        float y = v * s_par[cb + i] + s_par[cpad + cb + i];
                        v = (mish_f(y) * m + s_par[2 * cpad + cb + i]) * m;
                    }
                    __bf16 h, l;
                    split_bf16(v, h, l);
                    vh[i] = h;
                    vl[i] = l;
                }
                s_ah[it_lds[it]] = *reinterpret_cast<u32x4 *>(&vh);
                s_al[it_lds[it]] = *reinterpret_cast<u32x4 *>(&vl);
            }
        }
#pragma unroll
        for (int stage = 0; stage < NST; ++stage) {
            if (stage > 0) __syncthreads();   // previous stage's MFMAs are done with s_w
#pragma unroll
            for (int i = 0; i < WITER; ++i) s_w[tid + i * 256] = wregs[i];
            __syncthreads();
            // ---- prefetch behind the MFMAs: next weight block (one stage ahead) and, as early as the staging
            // registers are free again, the next activation chunk (a whole chunk of MFMAs ahead)
            if (stage + 1 < NST) load_w(chunk, stage + 1);
            else if (chunk + 1 < a.nchunk) load_w(chunk + 1, 0);
            if (stage == 0 && chunk + 1 < a.nchunk) load_act(chunk + 1);

#pragma unroll
            for (int j = 0; j < TPS; ++j) {
                int po[2];
#pragma unroll
                for (int ni = 0; ni < 2; ++ni) {
                    const int r = wn * 2 + ni;
                    if (MODE == CONV_C3) po[ni] = (r + stage) * HC + j;
                    else if (MODE == CONV_DN) po[ni] = (2 * r + stage) * HC + (j == 1 ? 33 : (j >> 1));
                    else if (MODE == CONV_UP) {
                        int dy = ph_y == 0 ? (stage == 0 ? 0 : -1) : (stage == 0 ? 1 : 0);
                        int dx = ph_x == 0 ? (j == 0 ? 0 : -1) : (j == 0 ? 1 : 0);
                        po[ni] = (r + 1 + dy) * HC + 1 + dx;
                    } else po[ni] = r * HC;
                }
#pragma unroll
                for (int kc = 0; kc < KCH; ++kc) {
                    bf16x8 wh[MF], wl[MF], xh[2], xl[2];
#pragma unroll
                    for (int mi = 0; mi < MF; ++mi) {
                        int wi = (j * NKG + kc * 2 + kg_l) * MT + m0 + mi * 32 + l31;
                        wh[mi] = *reinterpret_cast<const bf16x8 *>(&s_w[wi]);
                        wl[mi] = *reinterpret_cast<const bf16x8 *>(&s_w[wi + TPS * NKG * MT]);
                    }
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni) {
                        int xi = (kc * 2 + kg_l) * NPIX + po[ni] + l31;
                        xh[ni] = *reinterpret_cast<const bf16x8 *>(&s_ah[xi]);
                        xl[ni] = *reinterpret_cast<const bf16x8 *>(&s_al[xi]);
                    }
#pragma unroll
                    for (int mi = 0; mi < MF; ++mi)
#pragma unroll
                        for (int ni = 0; ni < 2; ++ni) {
                            if (lo_on) {
                                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[mi], xh[ni], acc[mi][ni], 0, 0, 0);
                                acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[mi], xl[ni], acc[mi][ni], 0, 0, 0);
                            }
                            acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[mi], xh[ni], acc[mi][ni], 0, 0, 0);
                        }
                }
            }
        }
    }

    // ---------------------------------------------------------------- epilogue
    const int HWout = a.Hout * a.Wout;
    const int gs = a.cout / a.groups;                   // channels per GroupNorm group (EPI_STATS)
    const float *bias = a.bias + (size_t)b * a.bias_bstride;
    float st1[MF][4], st2[MF][4];
#pragma unroll
    for (int mi = 0; mi < MF; ++mi)
#pragma unroll
        for (int q = 0; q < 4; ++q) { st1[mi][q] = 0.f; st2[mi][q] = 0.f; }

#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int r = wn * 2 + ni;
        int oy = y0 + r, ox = x0 + l31;
        if (MODE == CONV_UP) { oy = 2 * oy + ph_y; ox = 2 * ox + ph_x; }
        const bool pix_ok = oy < a.Hout && ox < a.Wout;
        float m_out = 0.f;
        if (a.epi == EPI_TAIL && pix_ok) m_out = a.mask[(size_t)b * a.T + ((size_t)ox << a.lvl_out)];
#pragma unroll
        for (int mi = 0; mi < MF; ++mi) {
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) {
                const int co = cot * MT + m0 + mi * 32 + (rg & 3) + 8 * (rg >> 2) + 4 * kg_l;
                const bool ok = pix_ok && co < a.cout;
                float v = acc[mi][ni][rg];
                if (ok) {
                    v += bias[co];
                    const size_t o = ((size_t)b * a.cout + co) * HWout + (size_t)oy * a.Wout + ox;
                    if (a.epi == EPI_TAIL) {
                        float y = a.eh[o] * a.esc[(size_t)b * a.cout + co] + a.esh[(size_t)b * a.cout + co];
                        v += mish_f(y) * m_out;
                    } else if (a.epi == EPI_ATTN) {
                        v += a.eres[o];
                    }
                    a.out[o] = v;
                    if (a.epi == EPI_STATS) {
                        // slot rg>>2 = the 8-channel octet of this 32-channel fragment (static index:
                        // runtime-indexed register arrays would go to scratch); octets -> groups below
                        st1[mi][rg >> 2] += v;
                        st2[mi][rg >> 2] += v * v;
                    }
                }
            }
        }
    }

    if (a.epi == EPI_STATS) {
        // wave reduce -> LDS -> fixed-order combine: one partial per (workgroup, group), deterministic
        __syncthreads();   // s_red aliases nothing live, but keep ordering simple
#pragma unroll
        for (int mi = 0; mi < MF; ++mi)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float s1 = wave_sum(st1[mi][q]);
                float s2 = wave_sum(st2[mi][q]);
                if (lane == 0) {
                    s_red[((wave * MF + mi) * 4 + q) * 2 + 0] = s1;
                    s_red[((wave * MF + mi) * 4 + q) * 2 + 1] = s2;
                }
            }
        __syncthreads();
        const int gpw = MT / gs > 0 ? MT / gs : 1;     // groups covered by this workgroup
        if (tid < gpw) {
            const int g = (cot * MT) / gs + tid;       // global group index
            if (g < a.groups) {
                float s1 = 0.f, s2 = 0.f;
                for (int w = 0; w < 4; ++w) {
                    const int wmm = w / WN;
                    for (int mi = 0; mi < MF; ++mi) {
                        const int cbase = cot * MT + (wmm * MF + mi) * 32;      // first channel of fragment
                        for (int q = 0; q < 4; ++q) {
                            const int gq = (cbase + q * 8) / gs;
                            if (gq == g) {
                                s1 += s_red[((w * MF + mi) * 4 + q) * 2 + 0];
                                s2 += s_red[((w * MF + mi) * 4 + q) * 2 + 1];
                            }
                        }
                    }
                }
                float *p = a.partials + (((size_t)b * a.nparts + blockIdx.x) * a.groups + g) * 2;
                p[0] = s1;
                p[1] = s2;
            }
        }
    }
}

template <int MODE, int WM, int WN, int MF, int KCH>
static hipError_t launch_cfg(const ConvArgs &a_in, hipStream_t st) {
    using C = ConvCfg<MODE, WM, WN, MF, KCH>;
    ConvArgs a = a_in;
    a.nchunk = (a.cin + 16 * KCH - 1) / (16 * KCH);
    const int th = (MODE == CONV_UP) ? a.Hin : a.Hout;   // tile space: input resolution for UP
    const int tw = (MODE == CONV_UP) ? a.Win : a.Wout;
    a.tiles_x = (tw + C::TC - 1) / C::TC;
    a.tiles_y = (th + C::TR - 1) / C::TR;
    const int ncot = (a.cout + C::MT - 1) / C::MT;
    dim3 grid(a.tiles_x * a.tiles_y, ncot * (MODE == CONV_UP ? 4 : 1), a.B);
    size_t smem = conv_smem_bytes(C::NPIX, C::NKG, C::WBLK16, a.cin, a.pro);
    static size_t attr_set = 0;
    if (smem > attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&conv_mfma_kernel<MODE, WM, WN, MF, KCH>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
        if (e != hipSuccess) return e;
        attr_set = smem;
    }
    hipLaunchKernelGGL((conv_mfma_kernel<MODE, WM, WN, MF, KCH>), grid, dim3(256), smem, st, a);
    return hipGetLastError();
}

// number of GroupNorm partial slots per sample that EPI_STATS writes for this layer
int conv_nparts(int mode, int cout, int Hout, int Wout) {
    ConvGeom g = conv_geom(mode, 64, cout);
    return ((Wout + 31) / 32) * ((Hout + g.TR - 1) / g.TR);
}

// template instances; must agree with conv_geom() in common.h
hipError_t launch_conv(int mode, const ConvArgs &a, hipStream_t st) {
    const bool wide = a.cout > 64;
    const bool k1 = a.cin <= 16;
    switch (mode) {
        case CONV_C3:
            if (wide) return k1 ? launch_cfg<CONV_C3, 2, 2, 2, 1>(a, st) : launch_cfg<CONV_C3, 2, 2, 2, 2>(a, st);
            return k1 ? launch_cfg<CONV_C3, 1, 4, 2, 1>(a, st) : launch_cfg<CONV_C3, 1, 4, 2, 2>(a, st);
        case CONV_DN:
            return wide ? launch_cfg<CONV_DN, 2, 2, 2, 1>(a, st) : launch_cfg<CONV_DN, 2, 2, 1, 1>(a, st);
        case CONV_UP:
            if (wide) return k1 ? launch_cfg<CONV_UP, 2, 2, 2, 1>(a, st) : launch_cfg<CONV_UP, 2, 2, 2, 2>(a, st);
            return k1 ? launch_cfg<CONV_UP, 1, 4, 2, 1>(a, st) : launch_cfg<CONV_UP, 1, 4, 2, 2>(a, st);
        case CONV_P1:
            if (wide) return k1 ? launch_cfg<CONV_P1, 2, 2, 2, 1>(a, st) : launch_cfg<CONV_P1, 2, 2, 2, 2>(a, st);
            return k1 ? launch_cfg<CONV_P1, 1, 4, 2, 1>(a, st) : launch_cfg<CONV_P1, 1, 4, 2, 2>(a, st);
    }
    return hipErrorInvalidValue;
}

}  // namespace gtts
