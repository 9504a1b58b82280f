// pack.hip -- one-time re-layout of the estimator parameters into the packed blob (device side).
//
// (CONV_C3 | 32: the f16 + fp8 format of GTTS_PREC_F16F8, 32-channel chunks -- see pack_conv_element and common.h.)
// Convolution weights become bf16 (hi, lo) pairs in MFMA-fragment order, one contiguous block per
// (phase, 16-channel chunk, stage, cout tile); inside a block the order is the LDS image of conv_mfma.hip:
//     [split: hi|lo][tap][kgroup 0..2*kch-1][cout-in-tile MT][8 channels]      (chunk = 16*kch input channels)
// Source layouts are the reference's state_dict layouts:
//   Conv2d          [cout][cin][kh][kw]        (diffusion.py:33,52,70,87-88)
//   ConvTranspose2d [cin][cout][4][4]          (diffusion.py:24)
#include "common.h"
#include "kernels.h"

namespace gtts {

// one (hi, lo) element pair t of one convolution's packed blob
// status (nullable; the f16 + fp8 format only): {count, bit pattern of max |w|, 1 + largest offending tag} of weights whose fp16 hi part
// w 2^S would overflow -- they are packed SATURATED (finite) and gtts_pack_weights reports GTTS_E_RANGE (plan.hip)
__device__ __forceinline__ void pack_conv_element(const float *__restrict__ w, __bf16 *__restrict__ dst, int mode_in, int cin, int cout,
                                                  int MT, int nst, int tps, int nchunk, int ncot, int nkg, size_t t,
                                                  unsigned *__restrict__ status = nullptr, unsigned tag = 0) {
    const int mode = mode_in & 31;
    const bool f16f8 = (mode_in & 32) != 0;      // CONV_C3 | 32: the f16 + fp8 format of GTTS_PREC_F16F8
    // decode t -> (phase, chunk, stage, cot, tap, kg, m, i)
    size_t r = t;
    const int i = r % 8; r /= 8;
    const int m = r % MT; r /= MT;
    const int kg = r % nkg; r /= nkg;
    const int tap = r % tps; r /= tps;
    const int cot = r % ncot; r /= ncot;
    const int stage = r % nst; r /= nst;
    const int chunk = r % nchunk; r /= nchunk;
    const int phase = (int)r;
    const int ci = chunk * 8 * nkg + kg * 8 + i;
    const int co = cot * MT + m;
    float v = 0.f;
    if (ci < cin && co < cout) {
        if (mode == CONV_C3 + 16) {
            // data-gradient convolution of a 3x3 conv: W'[co'][ci'][ky][kx] = W[ci'][co'][2-ky][2-kx], W stored [cout_orig = cin][cin_orig = cout]
            v = w[(((size_t)ci * cout + co) * 3 + (2 - stage)) * 3 + (2 - tap)];
        } else if (mode == CONV_C7) {
            v = w[(((size_t)co * cin + ci) * 7 + stage) * 7 + tap];          // ky = stage, kx = tap
        } else if (GTTS_W64_RING && mode == CONV_C3 && f16f8 && MT == 64) {
            v = w[(((size_t)co * cin + ci) * 3 + tap) * 3 + stage];          // the 64-channel f16 + fp8 tile walks COLUMN stages: kx = stage, ky = tap (conv_ws.hip, W64)
        } else if (mode == CONV_C3 || mode == CONV_DN) {
            v = w[(((size_t)co * cin + ci) * 3 + stage) * 3 + tap];          // ky = stage, kx = tap
        } else if (mode == CONV_P1) {
            v = w[(size_t)co * cin + ci];
        } else if (mode == CONV_P1 + 16) {
            v = w[(size_t)ci * cout + co];          // data gradient of a 1x1 conv: the forward weight [cin'][cout'] transposed
        } else {   // CONV_UP: output phase (py, px); stage/tap pick the two contributing kernel rows/cols
            const int py = phase >> 1, px = phase & 1;
            const int ky = py == 0 ? (stage == 0 ? 1 : 3) : (stage == 0 ? 0 : 2);
            const int kx = px == 0 ? (tap == 0 ? 1 : 3) : (tap == 0 ? 0 : 2);
            if (mode == CONV_UP + 16)      // Downsample's data gradient: its 3x3 forward weight [ci][co][3][3] read as a zero-padded 4x4 kernel
                v = (ky < 3 && kx < 3) ? w[(((size_t)ci * cout + co) * 3 + ky) * 3 + kx] : 0.f;
            else
                v = w[(((size_t)ci * cout + co) * 4 + ky) * 4 + kx];
        }
    }
    const size_t blk = (((size_t)phase * nchunk + chunk) * nst + stage) * ncot + cot;
    const size_t blk_elems = (size_t)tps * MT * 16 * nkg;                    // bf16 elements per block
    if (f16f8) {
        // GTTS_PREC_F16F8 (common.h): split 0 = fp16(w 2^S) in the bf16 hi plane's place; split 1 = [tap][g 0..3][cout][16 bytes] fp8:
        // g = plane * 2 + half, half = 16-channel half of the 32-channel chunk; plane 0 = q8(w), plane 1 = q8(wl 2^(S+D)),
        // wl = w - hi 2^-S.  (nkg == 4: kg * 8 + i is the channel inside the chunk.)
        // |w| 2^S must fit fp16 (|w| < 63.97): beyond it the hi part is saturated -- never inf -- and the event recorded
        constexpr float WMAX = 65504.0f / (float)(1 << F8_S);
        if (!(fabsf(v) <= WMAX)) {      // (NaN counts as out of range as well)
            if (status != nullptr) {
                atomicAdd(status + 0, 1u);
                atomicMax(status + 1, __builtin_bit_cast(unsigned, fabsf(v)) & 0x7fffffffu);
                atomicMax(status + 2, tag + 1);
            }
            v = v != v ? 0.f : (v > 0.f ? WMAX : -WMAX);
        }
        const _Float16 h = (_Float16)(v * (float)(1 << F8_S));
        const float wl = v - (float)h * (1.0f / (float)(1 << F8_S));
        reinterpret_cast<_Float16 *>(dst)[blk * blk_elems + (((size_t)(0 * tps + tap) * nkg + kg) * MT + m) * 8 + i] = h;
        const int cc = kg * 8 + i, half = cc >> 4, j = cc & 15;
        unsigned char *d8 = reinterpret_cast<unsigned char *>(dst) + (blk * blk_elems + (size_t)tps * nkg * MT * 8) * 2;
        const int q0 = __builtin_amdgcn_cvt_pk_fp8_f32(f8_sat(v), 0.f, 0, false);
        const int q1 = __builtin_amdgcn_cvt_pk_fp8_f32(f8_sat(wl * (float)(1 << (F8_S + F8_D))), 0.f, 0, false);
        d8[(((size_t)tap * nkg + half) * MT + m) * 16 + j] = (unsigned char)(q0 & 0xff);
        d8[(((size_t)tap * nkg + 2 + half) * MT + m) * 16 + j] = (unsigned char)(q1 & 0xff);
        return;
    }
    __bf16 hi, lo;
    split_bf16(v, hi, lo);
    const size_t e_hi = blk * blk_elems + (((size_t)(0 * tps + tap) * nkg + kg) * MT + m) * 8 + i;
    const size_t e_lo = blk * blk_elems + (((size_t)(1 * tps + tap) * nkg + kg) * MT + m) * 8 + i;
    dst[e_hi] = hi;
    dst[e_lo] = lo;
}

__global__ void pack_conv_kernel(const float *__restrict__ w, __bf16 *__restrict__ dst, int mode, int cin, int cout,
                                 int MT, int nst, int tps, int nchunk, int ncot, int nkg, size_t total, unsigned *__restrict__ status, unsigned tag) {
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;   // one thread per (hi, lo) element pair
    if (t >= total) return;
    pack_conv_element(w, dst, mode, cin, cout, MT, nst, tps, nchunk, ncot, nkg, t, status, tag);
}

// All convolution weights of a training step in ONE launch (blockIdx.y = weight): the packs of a step live for that step only
// (the weights change with every optimizer step), and ~90 separate 5-us pack launches were 0.4 ms of a 12 ms step.
__global__ void pack_conv_batch_kernel(const PackDesc *__restrict__ descs) {
    const PackDesc d = descs[blockIdx.y];
    for (size_t t = (size_t)blockIdx.x * 256 + threadIdx.x; t < d.total; t += (size_t)gridDim.x * 256)
        pack_conv_element(d.w, reinterpret_cast<__bf16 *>(d.dst), d.mode, d.cin, d.cout, d.MT, d.nst, d.tps, d.nchunk, d.ncot, d.nkg, t);
}

static void pack_geometry(int mode, int cin, int cout, PackDesc &d) {
    ConvGeom g = conv_geom(mode & 15, cin, cout, (mode & 32) ? 1 : 0);
    d.mode = mode; d.cin = cin; d.cout = cout;
    d.nkg = 2 * g.kch;
    d.nchunk = (cin + 8 * d.nkg - 1) / (8 * d.nkg);
    d.ncot = (cout + g.MT - 1) / g.MT;
    d.MT = g.MT; d.nst = g.nst; d.tps = g.tps;
    const int phases = (mode & 15) == CONV_UP ? 4 : 1;
    d.total = (size_t)phases * d.nchunk * g.nst * d.ncot * g.tps * d.nkg * g.MT * 8;
}

void pack_describe(int mode, const float *w, void *dst, int cin, int cout, PackDesc *out) {
    pack_geometry(mode, cin, cout, *out);
    out->w = w;
    out->dst = dst;
}

hipError_t launch_pack_batch(const PackDesc *descs_dev, int n, int grid_x, hipStream_t st) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(pack_conv_batch_kernel, dim3((unsigned)grid_x, (unsigned)n), dim3(256), 0, st, descs_dev);
    return hipGetLastError();
}

// mode CONV_C3 + 16 / CONV_P1 + 16: the transposed (and, 3x3, flipped) packing of a conv for its data gradient (cin, cout are
// those of the gradient convolution, i.e. swapped with respect to the forward weight tensor)
hipError_t launch_pack_conv(int mode, const float *w, unsigned char *dst, int cin, int cout, hipStream_t st, unsigned *status, unsigned tag) {
    PackDesc d;
    pack_geometry(mode, cin, cout, d);
    hipLaunchKernelGGL(pack_conv_kernel, dim3((unsigned)((d.total + 255) / 256)), dim3(256), 0, st, w,
                       reinterpret_cast<__bf16 *>(dst), mode, cin, cout, d.MT, d.nst, d.tps, d.nchunk, d.ncot, d.nkg, d.total, status, tag);
    return hipGetLastError();
}

// to_qkv.weight [384][C] (channel order (qkv, heads, c), diffusion.py:93-94) -> per-head k|v projection blocks
//   [head 4][stage][split][kg 2*KCH][row 64: k_h d=0..31 | v_h e=0..31][8 channels]
__global__ void pack_attn_kv_kernel(const float *__restrict__ wqkv, __bf16 *__restrict__ dst, int C, int nstage,
                                    size_t total) {
    constexpr int NKG = 2 * ATTN_KCH;
    const size_t t = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= total) return;
    size_t r = t;
    const int i = r % 8; r /= 8;
    const int row = r % 64; r /= 64;
    const int kg = r % NKG; r /= NKG;
    const int stage = r % nstage; r /= nstage;
    const int head = (int)r;
    const int ci = stage * 16 * ATTN_KCH + kg * 8 + i;
    const int src_row = row < 32 ? 128 + head * 32 + row : 256 + head * 32 + (row - 32);
    const float v = ci < C ? wqkv[(size_t)src_row * C + ci] : 0.f;
    __bf16 hi, lo;
    split_bf16(v, hi, lo);
    const size_t blk = (size_t)head * nstage + stage;
    const size_t blk_elems = (size_t)2 * NKG * 64 * 8;
    dst[blk * blk_elems + (((size_t)0 * NKG + kg) * 64 + row) * 8 + i] = hi;
    dst[blk * blk_elems + (((size_t)1 * NKG + kg) * 64 + row) * 8 + i] = lo;
}

hipError_t launch_pack_attn_kv(const float *wqkv, unsigned char *dst, int C, hipStream_t st) {
    const int nstage = (C + 16 * ATTN_KCH - 1) / (16 * ATTN_KCH);
    const size_t total = (size_t)4 * nstage * (2 * ATTN_KCH) * 64 * 8;
    hipLaunchKernelGGL(pack_attn_kv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, wqkv,
                       reinterpret_cast<__bf16 *>(dst), C, nstage, total);
    return hipGetLastError();
}

__global__ void copy_f32_kernel(const float *__restrict__ src, float *__restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) dst[i] = src[i];
}

hipError_t launch_copy_f32(const float *src, float *dst, size_t n, hipStream_t st) {
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(copy_f32_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, n);
    return hipGetLastError();
}

}  // namespace gtts
