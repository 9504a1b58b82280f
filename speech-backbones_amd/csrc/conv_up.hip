// conv_up.hip -- Upsample = ConvTranspose2d(dim, dim, 4, 2, 1) of x * mask (Grad-TTS/model/diffusion.py:19-25,171) with all
// four output phases computed from ONE staged input tile.
//
// conv_mfma.hip's CONV_UP mode runs the four phases (output row / column parity) as four workgroups: each stages the same
// (TR + 2) x 34 halo tile -- global loads, mask, bf16 hi / lo split, LDS writes -- for 2 x 2 taps of MFMA work, a quarter of a
// 3x3 convolution's, and the kernel is bound by that staging (171 / 126 us for the two Upsample layers of the Grad-TTS
// network at B = 16; the output alone is 84 / 42 us of HBM time).  Here a workgroup (4 waves, 64 output channels x 4 input
// rows x 32 input columns -> 8 x 64 output pixels) stages the tile once per 16-channel chunk and
//   wave w: output row parity py = w & 1, input row pair w >> 1, BOTH column parities:
//           accumulators [2 x 32 channels][2 rows][2 column phases] = 8 (128 registers); a lane then owns two ADJACENT output
//           columns (2 ox, 2 ox + 1) of a row and stores them as one 8-byte access: 64 lanes = full 128-byte lines
//           (the per-phase form writes every other column).
// Output row oy = 2 iy - 1 + ky: even rows meet kernel rows ky = 1, 3 at input rows r, r - 1; odd rows ky = 0, 2 at r + 1, r
// (the same for columns) -- the (stage, tap) order and the packed weight blob are those of CONV_UP (pack.hip), so results
// are bit-identical to the per-phase form.  Weight fragments come straight from the fragment-ordered blob (16-byte buffer
// loads, one tap ahead); the next chunk's activation loads are in flight across the MFMAs; one barrier per chunk (two LDS
// images).  fp32 storage, bf16x3 only (the single-pass bf16 modes keep the per-phase kernel).
// Measured (B = 16, both layers): 130-135 us avg against 148 for the per-phase form.  At bf16x3 the 64-channel layer executes
// 129 GFLOP (52 us at peak, ~100 us at the 0.5 of peak this chip sustains under MFMA load), so the remaining gap is MFMA
// time and the epilogue's 1024 lines per tile, not HBM.
#include "common.h"
#include "kernels.h"
#include <atomic>

namespace gtts {

constexpr int UP_TR = 4, UP_HR = UP_TR + 2, UP_HC = 34, UP_NPIX = UP_HR * UP_HC;      // halo tile: 6 x 34 pixels
constexpr int UP_ITEMS = 2 * UP_NPIX;                                                 // (8-channel group, pixel) staging items
constexpr int UP_LITER = (UP_ITEMS + 255) / 256;

__global__ __launch_bounds__(256, 2) void conv_up4_kernel(const ConvArgs a) {
    __shared__ __attribute__((aligned(16))) u32x4 s_img[2][2][2 * UP_NPIX];            // [buffer][hi | lo][kg][pixel]
    __shared__ float s_bias[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, kg_l = lane >> 5;
    const int py = wave & 1, rp = wave >> 1;

    const int ncot = a.cout / 64;
    const int tiles = a.tiles_x * a.tiles_y;
    int t = xcd_slot(blockIdx.x, gridDim.x);
    const int cot = t % ncot; t /= ncot;
    const int tx = t % a.tiles_x; t /= a.tiles_x;
    const int ty = t % a.tiles_y;
    const int b = t / a.tiles_y;
    (void)tiles;
    const int y0 = ty * UP_TR, x0 = tx * 32;
    const int HW = a.Hin * a.Win, HWo = a.Hout * a.Wout;
    const int nchunk = a.cin / 16;

    // ---- descriptors
    auto uniform_rsrc = [](const void *p, int bytes) {
        const unsigned long long u = reinterpret_cast<unsigned long long>(p);
        const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)u);
        const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)(u >> 32));
        return __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void *>(((unsigned long long)hi << 32) | lo), 0,
                                                 __builtin_amdgcn_readfirstlane(bytes), 0x00020000);
    };
    const __amdgpu_buffer_rsrc_t rsx = uniform_rsrc(reinterpret_cast<const float *>(a.src0) + (size_t)b * a.cin * HW, a.cin * HW * 4);
    const int MTP = a.cout > 64 ? 128 : 64, ncotp = a.cout / MTP, cpp = MTP / 64;
    const int wblk16 = 8 * MTP;                                  // 16-byte units of one packed block: [split 2][tap 2][kg 2][MTP]
    const int wtotal = 4 * nchunk * 2 * ncotp * wblk16 * 16;
    const __amdgpu_buffer_rsrc_t rsw = uniform_rsrc(a.w, wtotal);
    const int w_lane = (kg_l * MTP + (cot % cpp) * 64 + l31) * 16;

    // ---- staging items of this thread: (kg, halo row, halo column) -> 8 channels of one pixel
    int it_off[UP_LITER];        // byte offset of the pixel inside a channel plane, or -1
    float it_m[UP_LITER];
    int it_dst[UP_LITER];
#pragma unroll
    for (int it = 0; it < UP_LITER; ++it) {
        const int idx = tid + it * 256;
        const bool has = idx < UP_ITEMS;
        const int kg = idx >= UP_NPIX ? 1 : 0, pix = idx - kg * UP_NPIX;
        const int r = pix / UP_HC, c = pix - r * UP_HC;
        const int gy = y0 - 1 + r, gx = x0 - 1 + c;
        const bool in = has && gy >= 0 && gy < a.Hin && gx >= 0 && gx < a.Win;
        it_off[it] = in ? (gy * a.Win + gx + kg * 8 * HW) * 4 : -1;
        it_m[it] = in ? a.mask[(size_t)b * a.T + ((size_t)gx << a.lvl_in)] : 0.f;
        it_dst[it] = has ? kg * UP_NPIX + pix : -1;
    }
    float raw[UP_LITER][8];
    auto load_chunk = [&](int chunk) {
        const int soff = chunk * 16 * HW * 4;
#pragma unroll
        for (int it = 0; it < UP_LITER; ++it)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                raw[it][i] = it_off[it] >= 0 ? __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsx, it_off[it], soff + i * HW * 4, 0)) : 0.f;
    };
    auto stage_chunk = [&](int buf) {
#pragma unroll
        for (int it = 0; it < UP_LITER; ++it) {
            if (it_dst[it] < 0) continue;
            bf16x8 vh, vl;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __bf16 h, l;
                split_bf16(raw[it][i] * it_m[it], h, l);
                vh[i] = h;
                vl[i] = l;
            }
            s_img[buf][0][it_dst[it]] = __builtin_bit_cast(u32x4, vh);
            s_img[buf][1][it_dst[it]] = __builtin_bit_cast(u32x4, vl);
        }
    };

    // ---- weights of (chunk, stage, tap) for both column phases: [px][mi][hi | lo]
    auto wload = [&](bf16x8 (&w)[2][2][2], int chunk, int stage, int tap) {
#pragma unroll
        for (int px = 0; px < 2; ++px) {
            const int phase = py * 2 + px;
            const int blk = ((phase * nchunk + chunk) * 2 + stage) * ncotp + cot / cpp;
#pragma unroll
            for (int sp = 0; sp < 2; ++sp)
#pragma unroll
                for (int mi = 0; mi < 2; ++mi) {
                    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsw, w_lane + mi * 32 * 16,
                                                                          (blk * wblk16 + (sp * 2 + tap) * 2 * MTP) * 16, 0);
                    w[px][mi][sp] = __builtin_bit_cast(bf16x8, v);
                }
        }
    };

    f32x16 acc[2][2][2];      // [mi][row][px]
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
            for (int px = 0; px < 2; ++px)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mi][ni][px][r] = 0.f;
    if (tid < 64) s_bias[tid] = a.bias[cot * 64 + tid];

    bf16x8 wc[2][2][2], wn[2][2][2];
    load_chunk(0);
    wload(wc, 0, 0, 0);
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const int buf = chunk & 1;
        stage_chunk(buf);
        if (chunk + 1 < nchunk) load_chunk(chunk + 1);
        lds_barrier();
        const u32x4 *xh_p = &s_img[buf][0][kg_l * UP_NPIX], *xl_p = &s_img[buf][1][kg_l * UP_NPIX];
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            const int dyr = py == 0 ? (st == 0 ? 0 : -1) : (st == 0 ? 1 : 0);
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                // next tap's weights (the last tap of a chunk asks for the next chunk's first; past the end: its own, unused)
                const bool last = st == 1 && j == 1;
                const int nchk = last ? (chunk + 1 < nchunk ? chunk + 1 : chunk) : chunk;
                wload(wn, nchk, last ? 0 : (j == 1 ? st + 1 : st), last ? 0 : (j == 1 ? 0 : 1));
                __builtin_amdgcn_sched_barrier(0);      // (the requests stay in front of this tap's MFMAs: hipcc otherwise sinks
                                                        // them below the last reader of the registers they overwrite)
                bf16x8 xh[2][2], xl[2][2];      // [row][px]
#pragma unroll
                for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                    for (int px = 0; px < 2; ++px) {
                        const int dxc = px == 0 ? (j == 0 ? 0 : -1) : (j == 0 ? 1 : 0);
                        const int pi = (rp * 2 + ni + 1 + dyr) * UP_HC + 1 + dxc + l31;
                        xh[ni][px] = __builtin_bit_cast(bf16x8, xh_p[pi]);
                        xl[ni][px] = __builtin_bit_cast(bf16x8, xl_p[pi]);
                    }
#pragma unroll
                for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
                        for (int px = 0; px < 2; ++px) {
                            acc[mi][ni][px] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[px][mi][1], xh[ni][px], acc[mi][ni][px], 0, 0, 0);
                            acc[mi][ni][px] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[px][mi][0], xl[ni][px], acc[mi][ni][px], 0, 0, 0);
                            acc[mi][ni][px] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wc[px][mi][0], xh[ni][px], acc[mi][ni][px], 0, 0, 0);
                        }
#pragma unroll
                for (int px = 0; px < 2; ++px)
#pragma unroll
                    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                        for (int sp = 0; sp < 2; ++sp) wc[px][mi][sp] = wn[px][mi][sp];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- epilogue: + bias; a lane stores the column pair (2 ox, 2 ox + 1) of a channel row as 8 bytes
    const __amdgpu_buffer_rsrc_t rso = uniform_rsrc(reinterpret_cast<float *>(a.out) + (size_t)b * a.cout * HWo, a.cout * HWo * 4);
    const int ix = x0 + l31;
    typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;
#pragma unroll
    for (int ni = 0; ni < 2; ++ni) {
        const int iy = y0 + rp * 2 + ni;
        if (iy >= a.Hin || ix >= a.Win) continue;
        const int oy = 2 * iy + py;
        const int voff = (oy * a.Wout + 2 * ix + 4 * kg_l * HWo) * 4;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int rg = 0; rg < 16; ++rg) {
                const int ch = mi * 32 + (rg & 3) + 8 * (rg >> 2);          // (+ 4 kg_l: in voff)
                const float bv = s_bias[ch + 4 * kg_l];
                u32x2 v;
                v[0] = __builtin_bit_cast(unsigned, acc[mi][ni][0][rg] + bv);
                v[1] = __builtin_bit_cast(unsigned, acc[mi][ni][1][rg] + bv);
                __builtin_amdgcn_raw_buffer_store_b64(v, rso, voff, (cot * 64 + ch) * HWo * 4, GTTS_OUT_NT);
            }
    }
}

// The layers this kernel takes: Upsample of the training / sampling path in fp32 storage and bf16x3, whole 16-channel chunks,
// whole 64-channel output tiles, shared (not per-sample) weights.
bool conv_up4_eligible(const ConvArgs &a) {
    return a.nsplit == 2 && !a.act_bf16 && a.pro == PRO_MASK && a.epi == EPI_PLAIN && a.c1 == 0 && a.cin % 16 == 0 && a.cin >= 16 &&
           a.cout % 64 == 0 && (a.cout <= 64 || a.cout % 128 == 0) && a.w_bstride == 0 && a.bias_bstride == 0 &&
           a.Hout == 2 * a.Hin && a.Wout == 2 * a.Win && (size_t)a.cin * a.Hin * a.Win * 4 < ((size_t)1 << 31) &&
           (size_t)a.cout * a.Hout * a.Wout * 4 < ((size_t)1 << 31);
}

hipError_t launch_conv_up4(const ConvArgs &a_in, hipStream_t st) {
    ConvArgs a = a_in;
    if (!conv_up4_eligible(a)) return hipErrorInvalidValue;
    a.tiles_x = (a.Win + 31) / 32;
    a.tiles_y = (a.Hin + UP_TR - 1) / UP_TR;
    const long grid = (long)a.B * a.tiles_x * a.tiles_y * (a.cout / 64);
    if (grid <= 0 || grid > 0x7fffffffL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(conv_up4_kernel, dim3((unsigned)grid), dim3(256), 0, st, a);
    return hipGetLastError();
}

}  // namespace gtts
