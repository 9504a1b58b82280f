// ubench.hip -- measured ceilings of THIS chip for bench.py's roofline block ("where possible, a
// measured hipMemcpy / stream-triad ceiling").  Two micro-benchmarks, both on caller-owned buffers and the caller's stream:
//   gtts_ubench_mfma  back-to-back v_mfma_f32_32x32x16_bf16 on every SIMD of every CU, operands taken from LIVE data
//                     (the chip clocks to its power budget: zero-filled operands run ~20 % faster than random ones, so the
//                     quoted 2.5 PFLOP/s is not what a bf16x3 convolution on real activations can reach);
//   gtts_ubench_hbm   grid-stride 16-byte copy / triad / read-only sweeps over buffers far larger than the 256 MB Infinity Cache.
// They are measurement only: nothing on the sampling path calls them.
#include "kernels.h"
#include "../../include/gradtts_abi.h"

namespace gtts {

// Each wave keeps NA A-fragments and NB B-fragments in registers (NA * NB independent accumulators: no MFMA ever waits for
// its own accumulator) and issues `iters` sweeps over them.  The fragments come from `src` (8 bf16 per lane and fragment).
template <int NA, int NB>
__global__ __launch_bounds__(256, 2) void ubench_mfma_kernel(const unsigned short *src, size_t src_elems, float *out, int iters) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    bf16x8 a[NA], b[NB];
#pragma unroll
    for (int i = 0; i < NA + NB; ++i) {
        const size_t off = ((wave * (NA + NB) + i) * 64 + lane) * 8 % (src_elems - 8);
        const u32x4 v = *reinterpret_cast<const u32x4 *>(src + (off & ~(size_t)7));
        if (i < NA) a[i] = __builtin_bit_cast(bf16x8, v);
        else b[i - NA] = __builtin_bit_cast(bf16x8, v);
    }
    f32x16 acc[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

typedef __attribute__((ext_vector_type(4))) float f32x4;
// The same stream with v_mfma_f32_16x16x32_bf16 (4 accumulator registers per block, K = 32): does the other bf16 shape sustain a
// different rate under the power budget?  NA x NB blocks of 16 x 16.
typedef __attribute__((ext_vector_type(4))) float f32x4v;
template <int NA, int NB>
__global__ __launch_bounds__(256, 2) void ubench_mfma16_kernel(const unsigned short *src, size_t src_elems, float *out, int iters) {
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    bf16x8 a[NA], b[NB];
#pragma unroll
    for (int i = 0; i < NA + NB; ++i) {
        const size_t off = ((wave * (NA + NB) + i) * 64 + lane) * 8 % (src_elems - 8);
        const u32x4 v = *reinterpret_cast<const u32x4 *>(src + (off & ~(size_t)7));
        if (i < NA) a[i] = __builtin_bit_cast(bf16x8, v);
        else b[i - NA] = __builtin_bit_cast(bf16x8, v);
    }
    f32x4v acc[NA][NB];
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NA; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NA; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = s;
}

// mode 0: c = a (copy, 8 B per element moved); 1: c = a + 1.5 b (triad, 12 B); 2: read-only sum of a (4 B); NT: nontemporal
// loads / stores.  Four 16-byte loads per stream in flight per lane (a lane's items are a whole grid apart: every wave access
// is 1 KB contiguous).
template <int MODE, int NT>
__global__ __launch_bounds__(256) void ubench_hbm_kernel(const f32x4 *a, const f32x4 *b, f32x4 *c, size_t n4, float *sink) {
    constexpr int U = 4;
    const size_t stride = (size_t)gridDim.x * 256;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    auto ld = [](const f32x4 *p) { return NT ? __builtin_nontemporal_load(p) : *p; };
    auto st = [](f32x4 v, f32x4 *p) { if (NT) __builtin_nontemporal_store(v, p); else *p = v; };
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (U - 1) * stride < n4; i += U * stride) {
        f32x4 x[U], y[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = ld(a + i + u * stride);
        if (MODE == 1) {
#pragma unroll
            for (int u = 0; u < U; ++u) y[u] = ld(b + i + u * stride);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (MODE == 0) st(x[u], c + i + u * stride);
            else if (MODE == 1) st(x[u] + 1.5f * y[u], c + i + u * stride);
            else s += x[u];
        }
    }
    for (; i < n4; i += stride) {
        const f32x4 x = a[i];
        if (MODE == 0) c[i] = x;
        else if (MODE == 1) c[i] = x + 1.5f * b[i];
        else s += x;
    }
    if (MODE == 2 && s.x + s.y + s.z + s.w == 123.456f) sink[0] = s.x;     // keeps the loads alive, (almost) never taken
}

}  // namespace gtts

extern "C" {

size_t gtts_ubench_mfma_out_floats(int workgroups) { return workgroups > 0 ? (size_t)workgroups * 256 : 0; }

int gtts_ubench_mfma(const void *src, size_t src_bytes, float *out, int workgroups, int iters, double *flops, gtts_stream_t stream) {
    if (!src || !out) return gtts::set_error(GTTS_E_NULL, "gtts_ubench_mfma: null buffer");
    if (src_bytes < 4096 || workgroups <= 0 || iters == 0) return gtts::set_error(GTTS_E_SHAPE, "gtts_ubench_mfma: bad sizes");
    if (iters > 0) {
        hipLaunchKernelGGL((gtts::ubench_mfma_kernel<2, 4>), dim3(workgroups), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const unsigned short *>(src), src_bytes / 2, out, iters);
        if (flops) *flops = (double)workgroups * 4.0 * iters * 8.0 * (2.0 * 32 * 32 * 16);
    } else {        // iters < 0: the 16 x 16 x 32 shape, 4 x 4 blocks per wave, |iters| sweeps
        hipLaunchKernelGGL((gtts::ubench_mfma16_kernel<4, 4>), dim3(workgroups), dim3(256), 0, (hipStream_t)stream,
                           reinterpret_cast<const unsigned short *>(src), src_bytes / 2, out, -iters);
        if (flops) *flops = (double)workgroups * 4.0 * (-iters) * 16.0 * (2.0 * 16 * 16 * 32);
    }
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? GTTS_OK : gtts::set_error(GTTS_E_HIP, hipGetErrorString(e));
}

int gtts_ubench_hbm(const float *a, const float *b, float *c, size_t n, int mode, int workgroups, double *bytes, gtts_stream_t stream) {
    const int m = mode & 3, nt = mode >> 2;
    if (!a || (m == 1 && !b) || !c) return gtts::set_error(GTTS_E_NULL, "gtts_ubench_hbm: null buffer");
    if (n < 4 || n % 4 != 0 || workgroups <= 0 || mode < 0 || m > 2 || nt > 1) return gtts::set_error(GTTS_E_SHAPE, "gtts_ubench_hbm: bad sizes");
    const gtts::f32x4 *a4 = reinterpret_cast<const gtts::f32x4 *>(a), *b4 = reinterpret_cast<const gtts::f32x4 *>(b);
    gtts::f32x4 *c4 = reinterpret_cast<gtts::f32x4 *>(c);
    hipStream_t st = (hipStream_t)stream;
#define GTTS_UB(M, N) hipLaunchKernelGGL((gtts::ubench_hbm_kernel<M, N>), dim3(workgroups), dim3(256), 0, st, a4, b4, c4, n / 4, c)
    if (m == 0) { if (nt) GTTS_UB(0, 1); else GTTS_UB(0, 0); }
    else if (m == 1) { if (nt) GTTS_UB(1, 1); else GTTS_UB(1, 0); }
    else { if (nt) GTTS_UB(2, 1); else GTTS_UB(2, 0); }
#undef GTTS_UB
    if (bytes) *bytes = (double)n * (m == 0 ? 8.0 : (m == 1 ? 12.0 : 4.0));
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? GTTS_OK : gtts::set_error(GTTS_E_HIP, hipGetErrorString(e));
}

}  // extern "C"
