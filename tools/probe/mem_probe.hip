// mem_probe.hip -- two questions the 64-channel tile of conv_ws.hip turns on (round 6), answered on the chip.
//   hipcc --offload-arch=gfx950 -O3 -o mem_probe mem_probe.hip && ./mem_probe
// S: what bounds a tile epilogue -- per-CU store ISSUE (dword vs dwordx4 forms) or the chip's write bandwidth when every CU
//    stores at once?  Each workgroup (4 waves) writes 160 KB per round in one of four access forms, on 256 / 128 / 64 / 32 CUs.
// L: does an L2-hit load wait behind HBM-miss loads of OTHER waves of the same CU (in-order vector L1)?  Waves 0-3 time
//    dependent 16-byte loads from a 144 KB (L2-resident) table while waves 4-7 stream a 2 GB buffer with D loads in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t mk_rsrc(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000);
}

// ---- S: stores.  form 0: dword, 2 rows x 128 B per instruction (the accumulator layout of the conv epilogue)
//                  form 1: dwordx4, 8 rows x 128 B per instruction (a lane owns 4 consecutive frames of one channel row)
//                  form 2: dwordx4, 1 KB contiguous per instruction;  form 3: dword, 256 B contiguous per instruction
template <int FORM>
__global__ __launch_bounds__(256) void store_kernel(float *out, int rounds, int row_stride_f, unsigned long long *cyc) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int r = 0; r < rounds; ++r) {
        // 160 KB per workgroup and round = 40 KB per wave = 10 accumulators of 32 channels x 32 frames.
        // strided forms: a [1024 rows][row_stride_f] tensor; workgroup b owns rows 128 (b % 8) + 32 wave .. + 31 and, per round, 320 columns
        float *tile = out + (size_t)(128 * (blockIdx.x & 7) + 32 * wave) * row_stride_f + (size_t)((blockIdx.x >> 3) * rounds + r) * 320;
        float *rb = out + ((size_t)blockIdx.x * rounds + r) * 40960 + (size_t)wave * 10240;      // contiguous forms
#pragma unroll
        for (int a = 0; a < 10; ++a) {
            if constexpr (FORM == 0) {
                float *p = tile + a * 32 + (lane & 31) + (size_t)(4 * (lane >> 5)) * row_stride_f;
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int i = 0; i < 4; ++i) p[(size_t)(8 * q + i) * row_stride_f] = (float)(a + q + i);
            } else if constexpr (FORM == 1) {
                float *p = tile + a * 32 + 4 * (lane & 7) + (size_t)(lane >> 3) * row_stride_f;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 v = make_float4((float)a, (float)q, 1.f, 2.f);
                    *reinterpret_cast<float4 *>(p + (size_t)(8 * q) * row_stride_f) = v;
                }
            } else if constexpr (FORM == 2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float4 v = make_float4((float)a, (float)q, 1.f, 2.f);
                    *reinterpret_cast<float4 *>(rb + a * 1024 + q * 256 + lane * 4) = v;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q) rb[a * 1024 + q * 64 + lane] = (float)(a + q);
            }
        }
    }
    __builtin_amdgcn_s_waitcnt(0);      // vmcnt(0): the stores are acknowledged
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// ---- L: latency of L2-hit loads beside a streaming wave set
__global__ __launch_bounds__(512) void lat_kernel(const u32x4 *table, int table_n16, const u32x4 *big, size_t big_n16, int depth, int iters,
                                                  int stream_on, unsigned long long *lat_sum, unsigned *sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave < 4) {
        // dependent chain: the next index comes out of the loaded value (table holds small pseudo-random indices)
        unsigned idx = (blockIdx.x * 64 + lane + wave * 977) % (unsigned)table_n16;
        unsigned long long sum = 0;
        unsigned acc = 0;
        for (int i = 0; i < iters; ++i) {
            const unsigned long long t0 = __builtin_amdgcn_s_memtime();
            const u32x4 v = table[idx];
            asm volatile("s_waitcnt vmcnt(0)" ::"v"(v) : "memory");
            const unsigned long long t1 = __builtin_amdgcn_s_memtime();
            sum += t1 - t0;
            acc += v[1];
            idx = (v[0] + lane) % (unsigned)table_n16;
        }
        if (lane == 0) lat_sum[blockIdx.x * 4 + wave] = sum;
        if (acc == 0x12345678u) sink[0] = acc;
    } else if (stream_on) {
        // streamers: `depth` 16-byte loads in flight per lane, sweeping a buffer far larger than the caches
        const size_t per_wg = big_n16 / gridDim.x;
        const u32x4 *p = big + (size_t)blockIdx.x * per_wg + (size_t)(wave - 4) * 64 + lane;
        unsigned acc = 0;
        const int steps = iters * 6;       // long enough to cover the timed waves
        size_t off = 0;
        for (int s = 0; s < steps; ++s) {
            u32x4 v[8];
#pragma unroll
            for (int d = 0; d < 8; ++d)
                if (d < depth) { v[d] = __builtin_nontemporal_load(p + off); off += 256; if (off + 256 >= per_wg) off = 0; }
#pragma unroll
            for (int d = 0; d < 8; ++d)
                if (d < depth) acc += v[d][0];
        }
        if (acc == 0x12345678u) sink[1] = acc;
    }
}

int main() {
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("device %s, %d CUs, clock %d kHz\n", prop.name, cus, prop.clockRate);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));

    // ------------------------------------------------------------------ S
    {
        const int rounds = 8;
        const int row_stride_f = 80 * 1024;                       // one channel row of a level-0 tensor
        float *out;
        const size_t total = (size_t)1024 * row_stride_f;          // [1024 rows][80 x 1024]: a level-0 tensor of B = 16 x 64 channels
        CK(hipMalloc(&out, total * 4));
        unsigned long long *cyc;
        CK(hipMalloc(&cyc, 4096 * 8));
        printf("\nS: 160 KB per workgroup and round, %d rounds; bytes per clock and CU from s_memtime (mean over workgroups), GB/s from events\n", rounds);
        printf("%-44s %8s %12s %12s %10s\n", "form", "CUs", "cycles/round", "B/clk/CU", "GB/s");
        const char *names[4] = {"dword, 2 rows x 128 B (accumulator layout)", "dwordx4, 8 rows x 128 B", "dwordx4, 1 KB contiguous", "dword, 256 B contiguous"};
        for (int form = 0; form < 4; ++form) {
            for (int g : {cus, cus / 2, cus / 4, cus / 8}) {
                float best_ms = 1e9f;
                double cyc_mean = 0;
                for (int rep = 0; rep < 3; ++rep) {
                    CK(hipEventRecord(e0));
                    switch (form) {
                        case 0: store_kernel<0><<<g, 256>>>(out, rounds, row_stride_f, cyc); break;
                        case 1: store_kernel<1><<<g, 256>>>(out, rounds, row_stride_f, cyc); break;
                        case 2: store_kernel<2><<<g, 256>>>(out, rounds, row_stride_f, cyc); break;
                        default: store_kernel<3><<<g, 256>>>(out, rounds, row_stride_f, cyc); break;
                    }
                    CK(hipEventRecord(e1));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (ms < best_ms) {
                        best_ms = ms;
                        std::vector<unsigned long long> h(g);
                        CK(hipMemcpy(h.data(), cyc, g * 8, hipMemcpyDeviceToHost));
                        cyc_mean = 0;
                        for (auto c : h) cyc_mean += (double)c;
                        cyc_mean /= g;
                    }
                }
                const double bytes = (double)g * rounds * 160 * 1024;
                printf("%-44s %8d %12.0f %12.2f %10.0f\n", names[form], g, cyc_mean / rounds, 160.0 * 1024 * rounds / cyc_mean, bytes / best_ms * 1e-6);
            }
        }
        CK(hipFree(out));
        CK(hipFree(cyc));
    }

    // ------------------------------------------------------------------ L
    {
        const int table_n16 = 144 * 1024 / 16;
        std::vector<unsigned> h((size_t)table_n16 * 4);
        unsigned s = 12345;
        for (int i = 0; i < table_n16; ++i) {
            s = s * 1664525u + 1013904223u;
            h[i * 4 + 0] = (s >> 8) % table_n16;
            h[i * 4 + 1] = s;
            h[i * 4 + 2] = h[i * 4 + 3] = 0;
        }
        u32x4 *table;
        CK(hipMalloc(&table, (size_t)table_n16 * 16));
        CK(hipMemcpy(table, h.data(), (size_t)table_n16 * 16, hipMemcpyHostToDevice));
        const size_t big_n16 = ((size_t)2 << 30) / 16;
        u32x4 *big;
        CK(hipMalloc(&big, big_n16 * 16));
        CK(hipMemset(big, 1, big_n16 * 16));
        unsigned long long *lat;
        CK(hipMalloc(&lat, (size_t)cus * 4 * 8));
        unsigned *sink;
        CK(hipMalloc(&sink, 16));
        const int iters = 2000;
        printf("\nL: mean latency (cycles) of a dependent 16-byte load from a 144 KB table (L2 / L1 resident), 4 timed waves per CU, beside 4 streaming waves\n");
        printf("%-40s %12s %12s\n", "streamers", "latency", "kernel ms");
        for (int cfg = 0; cfg < 6; ++cfg) {
            const int stream_on = cfg > 0;
            const int depth = cfg == 0 ? 0 : (cfg == 1 ? 1 : cfg == 2 ? 2 : cfg == 3 ? 4 : cfg == 4 ? 8 : 8);
            const int g = cus;
            float ms = 0;
            for (int rep = 0; rep < 2; ++rep) {
                CK(hipEventRecord(e0));
                lat_kernel<<<g, 512>>>(table, table_n16, big, big_n16, depth, cfg == 5 ? iters / 2 : iters, stream_on, lat, sink);
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            std::vector<unsigned long long> hl((size_t)g * 4);
            CK(hipMemcpy(hl.data(), lat, (size_t)g * 4 * 8, hipMemcpyDeviceToHost));
            double m = 0;
            for (auto c : hl) m += (double)c;
            m /= (double)hl.size() * (cfg == 5 ? iters / 2 : iters);
            char name[64];
            if (!stream_on) snprintf(name, sizeof name, "none");
            else snprintf(name, sizeof name, "%d x 16 B in flight per lane", depth);
            printf("%-40s %12.0f %12.3f\n", name, m, ms);
        }
        CK(hipFree(table));
        CK(hipFree(big));
        CK(hipFree(lat));
        CK(hipFree(sink));
    }
    return 0;
}
