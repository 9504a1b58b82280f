// f8_probe.hip -- standalone gfx950 probe (no torch): semantics of v_mfma_f32_32x32x64_f8f6f4 (fp8 e4m3 operands, with and
// without E8M0 block scales), of the fp8 conversions, and the sustained rate of the candidate operand splits of a 32-channel
// contraction step under the chip's power cap:
//     bf16x3  : 2 k-steps x 3 v_mfma_f32_32x32x16_bf16            (what conv_mfma.hip issues today)
//     f16f8   : 2 k-steps x 1 v_mfma_f32_32x32x16_f16 + 1 v_mfma_f32_32x32x64_f8f6f4  (hi*hi in fp16, both cross terms in one fp8 MFMA)
// Build:  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe/f8_probe.hip -o tools/probe/f8_probe
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(2))) short s16x2;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

// ------------------------------------------------------------------------------------------------ semantics
// A [32][64] fp8 row-major, B [64][32] (B[k][n]).  Assumed operand layout: lane l holds row/col l & 31 and the K block l >> 5
// (32 consecutive K values = 32 bytes = 8 registers, byte j of the 32 <-> k = 32 (l >> 5) + j).  C layout as for the bf16 32x32 shapes.
__global__ void k_sem(const unsigned char *A, const unsigned char *B, float *C, int mode, const int *sa, const int *sb) {
    const int lane = threadIdx.x, r = lane & 31, kb = lane >> 5;
    i32x8 a, b;
    for (int q = 0; q < 8; ++q) {
        unsigned av = 0, bv = 0;
        for (int t = 0; t < 4; ++t) {
            const int k = kb * 32 + q * 4 + t;
            av |= (unsigned)A[r * 64 + k] << (8 * t);
            bv |= (unsigned)B[k * 32 + r] << (8 * t);
        }
        a[q] = (int)av;
        b[q] = (int)bv;
    }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    if (mode == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
    else c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, sa[lane], 0, sb[lane]);
    for (int i = 0; i < 16; ++i) C[((i & 3) + 8 * (i >> 2) + 4 * kb) * 32 + r] = c[i];
}

__global__ void k_cvt(const float *x, int n, unsigned *out_plain, unsigned *out_scaled, float scale) {
    const int i = threadIdx.x;
    if (2 * i + 1 >= n) return;
    out_plain[i] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], 0, false) & 0xffffu;
    s16x2 old = {0, 0};
    const s16x2 r = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, x[2 * i], x[2 * i + 1], scale, false);
    out_scaled[i] = (unsigned)(unsigned short)r[0];
}

static float e4m3_decode(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f;
    if (e == 0) f = ldexpf((float)m, -9);
    else if (e == 15 && m == 7) f = NAN;
    else f = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -f : f;
}

// ------------------------------------------------------------------------------------------------ throughput
// One "unit" = the MFMA work of one 32-channel step of one (MF x NF) wave tile, operands resident in registers.
// MODE 0 bf16x3, 1 f16f8 (unscaled fp8), 2 f16 only (the two hi*hi k-steps), 3 fp8 only, 4 f16f8 with the scaled fp8 instruction
template <int MODE, int NA, int NB>
__global__ __launch_bounds__(256, 2) void k_rate(const i32x4 *src, size_t n16, float *out, int iters) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    // planes of 16-byte fragments in src: [0] A hi k0, [1] A hi k1, [2] A lo k0, [3] A lo k1 (bf16: lo; f8: the two halves of the fp8 operand), same for B at +4
    auto frag = [&](int plane, int idx) {
        const size_t off = ((size_t)plane * (n16 / 8) + ((wave * 8 + idx) * 64 + lane) % (n16 / 8));
        return src[off];
    };
    i32x4 ah[2][NA], al[2][NA], bh[2][NB], bl[2][NB];
    for (int k = 0; k < 2; ++k) {
        for (int i = 0; i < NA; ++i) { ah[k][i] = frag(k, i); al[k][i] = frag(2 + k, i); }
        for (int j = 0; j < NB; ++j) { bh[k][j] = frag(4 + k, j); bl[k][j] = frag(6 + k, j); }
    }
    f32x16 acc[NA][NB];
    for (int i = 0; i < NA; ++i)
        for (int j = 0; j < NB; ++j)
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int sca = lane < 32 ? 127 : 123, scb = lane < 32 ? 117 : 127;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                if (MODE == 0) {
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al[k][i]), __builtin_bit_cast(bf16x8, bh[k][j]), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[k][i]), __builtin_bit_cast(bf16x8, bl[k][j]), acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah[k][i]), __builtin_bit_cast(bf16x8, bh[k][j]), acc[i][j], 0, 0, 0);
                    }
                } else {
                    if (MODE == 1 || MODE == 2 || MODE == 4) {
#pragma unroll
                        for (int k = 0; k < 2; ++k)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[k][i]), __builtin_bit_cast(f16x8, bh[k][j]), acc[i][j], 0, 0, 0);
                    }
                    if (MODE == 1 || MODE == 3 || MODE == 4) {
                        i32x8 a8, b8;
                        for (int q = 0; q < 4; ++q) { a8[q] = al[0][i][q]; a8[4 + q] = al[1][i][q]; b8[q] = bl[0][j][q]; b8[4 + q] = bl[1][j][q]; }
                        if (MODE == 4) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[i][j], 0, 0, 0, sca, 0, scb);
                        else acc[i][j] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[i][j], 0, 0, 0, 0, 0, 0);
                    }
                }
            }
    }
    float s = 0.f;
    for (int i = 0; i < NA; ++i)
        for (int j = 0; j < NB; ++j)
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

static unsigned short f2bf(float f) {
    unsigned u; memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1);
    return (unsigned short)(u >> 16);
}
static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned short f2h(float f) { _Float16 h = (_Float16)f; unsigned short u; memcpy(&u, &h, 2); return u; }
static float h2f(unsigned short u) { _Float16 h; memcpy(&h, &u, 2); return (float)h; }
static unsigned char f2e4m3(float f) {        // RNE, saturating (host-side reference encoder)
    if (f != f) return 0x7f;
    const int s = f < 0; float a = fabsf(f);
    if (a > 448.f) a = 448.f;
    unsigned char best = 0; float bd = 1e30f;
    for (int v = 0; v < 127; ++v) { const float d = fabsf(e4m3_decode((unsigned char)v) - a); if (d < bd || (d == bd && !(v & 1))) { bd = d; best = (unsigned char)v; } }
    return (unsigned char)(best | (s << 7));
}
static float gauss() { float u = (rand() + 1.0f) / (RAND_MAX + 2.0f), v = (rand() + 1.0f) / (RAND_MAX + 2.0f); return sqrtf(-2.f * logf(u)) * cosf(6.2831853f * v); }

template <int MODE>
static double run_rate(const i32x4 *d_src, size_t n16, float *d_out, int wgs, double seconds, const char *name, double macs_per_unit) {
    constexpr int NA = 2, NB = 4;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k_rate<MODE, NA, NB>), dim3(wgs), dim3(256), 0, 0, d_src, n16, d_out, iters);
    CK(hipDeviceSynchronize());
    // run for `seconds` (the power cap needs a while to bite), report the rate of the second half
    std::vector<float> ms;
    double total = 0;
    while (total < seconds * 1e3) {
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < 4; ++r) hipLaunchKernelGGL((k_rate<MODE, NA, NB>), dim3(wgs), dim3(256), 0, 0, d_src, n16, d_out, iters);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1));
        ms.push_back(t / 4); total += t;
    }
    std::vector<float> tail(ms.begin() + ms.size() / 2, ms.end());
    std::sort(tail.begin(), tail.end());
    const double med = tail[tail.size() / 2];
    const double units = (double)wgs * 4 * iters * NA * NB;
    printf("rate %-22s %8.3f ms/launch  %8.1f G units/s  = %7.1f T channel-MAC-FLOP/s (x2)  [first %.3f ms, n=%zu]\n", name, med, units / med * 1e-6,
           2.0 * units * macs_per_unit / med * 1e-9, ms[0], ms.size());
    return units / med;
}

int main(int argc, char **argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 1.5;
    srand(1234);
    // ---------------- semantics
    std::vector<unsigned char> A(32 * 64), B(64 * 32);
    for (auto &v : A) { do v = (unsigned char)(rand() & 0xff); while ((v & 0x7f) == 0x7f || ((v >> 3) & 15) > 10); }
    for (auto &v : B) { do v = (unsigned char)(rand() & 0xff); while ((v & 0x7f) == 0x7f || ((v >> 3) & 15) > 10); }
    unsigned char *dA, *dB; float *dC; int *dsa, *dsb;
    CK(hipMalloc(&dA, A.size())); CK(hipMalloc(&dB, B.size())); CK(hipMalloc(&dC, 32 * 32 * 4)); CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256));
    CK(hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice));
    std::vector<float> C(32 * 32);
    auto expect = [&](int m, int n, const int *sa, const int *sb) {
        double s = 0;
        for (int kb = 0; kb < 2; ++kb) {
            double p = 0;
            for (int j = 0; j < 32; ++j) p += (double)e4m3_decode(A[m * 64 + kb * 32 + j]) * (double)e4m3_decode(B[(kb * 32 + j) * 32 + n]);
            if (sa) p *= ldexp(1.0, (sa[m + 32 * kb] & 0xff) - 127) * ldexp(1.0, (sb[n + 32 * kb] & 0xff) - 127);
            s += p;
        }
        return s;
    };
    for (int test = 0; test < 4; ++test) {
        int sa[64], sb[64];
        for (int l = 0; l < 64; ++l) {
            if (test == 1) { sa[l] = l < 32 ? 127 : 127 - 16; sb[l] = l < 32 ? 127 - 10 : 127 - 4; }        // per-K-block scales (the form the conv would use)
            else if (test == 2) { sa[l] = 127 + (l & 3); sb[l] = 127; }                                       // row-dependent A scale
            else { sa[l] = 127; sb[l] = 127 - (l & 7); }                                                      // column-dependent B scale
        }
        CK(hipMemcpy(dsa, sa, 256, hipMemcpyHostToDevice)); CK(hipMemcpy(dsb, sb, 256, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, dA, dB, dC, test == 0 ? 0 : 1, dsa, dsb);
        CK(hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost));
        double maxe = 0, maxr = 0;
        for (int m = 0; m < 32; ++m)
            for (int n = 0; n < 32; ++n) {
                const double ex = expect(m, n, test == 0 ? nullptr : sa, test == 0 ? nullptr : sb);
                maxe = fmax(maxe, fabs(ex - C[m * 32 + n])); maxr = fmax(maxr, fabs(ex));
            }
        printf("sem test %d (%s): max|err| %.3e of max|ref| %.3e -> %s\n", test,
               test == 0 ? "unscaled, lane = (row, K block)" : (test == 1 ? "per-K-block scales" : (test == 2 ? "row-dependent A scale" : "col-dependent B scale")),
               maxe, maxr, maxe <= 1e-5 * maxr ? "MATCH" : "MISMATCH");
        if (maxe > 1e-5 * maxr) {
            printf("  C[0][0..3] = %g %g %g %g ; expected %g %g %g %g\n", C[0], C[1], C[2], C[3], expect(0, 0, test ? sa : nullptr, test ? sb : nullptr),
                   expect(0, 1, test ? sa : nullptr, test ? sb : nullptr), expect(0, 2, test ? sa : nullptr, test ? sb : nullptr), expect(0, 3, test ? sa : nullptr, test ? sb : nullptr));
        }
    }
    // ---------------- conversions
    {
        const float xs[] = {0.f, 1.f, -1.f, 0.3f, 447.f, 448.f, 449.f, 464.f, 500.f, 1000.f, -1e6f, 1e-3f, 0.0019f, 0.001953125f, 0.0009765625f, 0.0146f, 0.0156f, 3.3f, -7.7f, INFINITY, NAN, 240.f, 17.f, 18.f};
        const int n = sizeof(xs) / sizeof(xs[0]);
        float *dx; unsigned *dp, *ds;
        CK(hipMalloc(&dx, n * 4)); CK(hipMalloc(&dp, n * 4)); CK(hipMalloc(&ds, n * 4));
        CK(hipMemcpy(dx, xs, n * 4, hipMemcpyHostToDevice));
        for (float scale : {1.0f, 4.0f, 0.25f}) {
            hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, dx, n, dp, ds, scale);
            std::vector<unsigned> p(n / 2), s(n / 2);
            CK(hipMemcpy(p.data(), dp, n / 2 * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(s.data(), ds, n / 2 * 4, hipMemcpyDeviceToHost));
            for (int i = 0; i < n / 2; ++i)
                for (int h = 0; h < 2; ++h) {
                    const unsigned char vp = (p[i] >> (8 * h)) & 0xff, vs = (s[i] >> (8 * h)) & 0xff;
                    if (scale == 1.0f)
                        printf("cvt x = %-12g plain 0x%02x = %-10g host-RNE-sat 0x%02x | scalef32(scale=1) 0x%02x = %g\n", xs[2 * i + h], vp, e4m3_decode(vp), f2e4m3(xs[2 * i + h]), vs, e4m3_decode(vs));
                    else
                        printf("cvt x = %-12g scalef32(scale=%g) 0x%02x = %g\n", xs[2 * i + h], scale, vs, e4m3_decode(vs));
                }
        }
    }
    // ---------------- throughput on realistic operand bits
    {
        const size_t per = 1 << 16;                 // 16-byte fragments per plane
        const size_t n16 = per * 8;
        int dev = 0; hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, dev));
        const int wgs = pr.multiProcessorCount * 2;
        float *d_out; CK(hipMalloc(&d_out, (size_t)wgs * 256 * 4));
        i32x4 *d_src; CK(hipMalloc(&d_src, n16 * 16));
        std::vector<unsigned char> h(n16 * 16);
        auto fill = [&](int fmt) {      // fmt 0: bf16 hi/lo planes, 1: f16 hi + fp8 (lo-scaled | hi) planes, 2: zeros
            for (int side = 0; side < 2; ++side)            // 0: A = weights U(+-0.04), 1: B = activations N(0, 1)
                for (int k = 0; k < 2; ++k)
                    for (size_t f = 0; f < per; ++f) {
                        unsigned char *hi = &h[(((size_t)side * 4 + k) * per + f) * 16], *lo = &h[(((size_t)side * 4 + 2 + k) * per + f) * 16];
                        if (fmt == 2) { memset(hi, 0, 16); memset(lo, 0, 16); continue; }
                        float v[16];
                        for (int i = 0; i < 16; ++i) v[i] = side == 0 ? (rand() / (float)RAND_MAX * 2 - 1) * 0.04f : gauss();
                        if (fmt == 0) {
                            for (int i = 0; i < 8; ++i) {
                                const unsigned short b = f2bf(v[i]), l = f2bf(v[i] - bf2f(b));
                                memcpy(hi + 2 * i, &b, 2); memcpy(lo + 2 * i, &l, 2);
                            }
                        } else {
                            for (int i = 0; i < 8; ++i) { const unsigned short b = f2h(v[i]); memcpy(hi + 2 * i, &b, 2); }
                            // fp8 plane k = 0: the operand's first 16 bytes (A: q8(w), B: q8(xl 2^10)), k = 1: second (A: q8(wl 2^14), B: q8(x / 16))
                            for (int i = 0; i < 16; ++i) {
                                const float r = v[i] - h2f(f2h(v[i]));
                                float t = side == 0 ? (k == 0 ? v[i] : r * 16384.f) : (k == 0 ? r * 1024.f : v[i] / 16.f);
                                lo[i] = f2e4m3(t);
                            }
                        }
                    }
            CK(hipMemcpy(d_src, h.data(), h.size(), hipMemcpyHostToDevice));
        };
        printf("throughput: %d workgroups x 256 threads (2 per CU), NA x NB = 2 x 4 accumulators per wave, %.1f s per mode\n", wgs, secs);
        const double mac32 = 32.0 * 32 * 32;      // channel-MACs of one unit (32 channels, 32 x 32 outputs)
        fill(0);
        const double r0 = run_rate<0>(d_src, n16, d_out, wgs, secs, "bf16x3 (6 bf16)", mac32);
        fill(1);
        const double r1 = run_rate<1>(d_src, n16, d_out, wgs, secs, "f16f8 (2 f16 + 1 fp8)", mac32);
        run_rate<2>(d_src, n16, d_out, wgs, secs, "f16 only (2 f16)", mac32);
        run_rate<3>(d_src, n16, d_out, wgs, secs, "fp8 only (1 fp8 K64)", mac32);
        const double r4 = run_rate<4>(d_src, n16, d_out, wgs, secs, "f16f8 scaled fp8", mac32);
        fill(0);
        const double r0b = run_rate<0>(d_src, n16, d_out, wgs, secs, "bf16x3 again", mac32);
        fill(2);
        run_rate<0>(d_src, n16, d_out, wgs, secs, "bf16x3 zeros", mac32);
        run_rate<1>(d_src, n16, d_out, wgs, secs, "f16f8 zeros", mac32);
        printf("RATIO f16f8 / bf16x3 = %.3f (scaled fp8 form: %.3f)\n", r1 / (0.5 * (r0 + r0b)), r4 / (0.5 * (r0 + r0b)));
    }
    return 0;
}
