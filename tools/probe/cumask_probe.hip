// Diagnostic: which (XCC, SE, CU) a CU-masked stream's workgroups land on (hipExtStreamCreateWithCUMask bit layout on MI355X).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <set>
#include <map>
__global__ void where_kernel(unsigned *out) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // spin a little so that all workgroups are resident at once
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    while (__builtin_amdgcn_s_memtime() - t0 < 200000ull) {}
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
}
int main(int argc, char **argv) {
    int nbits = argc > 1 ? atoi(argv[1]) : 64;       // number of low mask bits set
    int stride = argc > 2 ? atoi(argv[2]) : 1;       // set every stride-th bit
    std::vector<uint32_t> mask(8, 0);
    for (int i = 0, k = 0; k < nbits && i < 256; i += stride, ++k) mask[i / 32] |= 1u << (i % 32);
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, 8, mask.data());
    printf("create: %s  mask %08x %08x %08x %08x %08x %08x %08x %08x\n", hipGetErrorString(e), mask[0], mask[1], mask[2], mask[3], mask[4], mask[5], mask[6], mask[7]);
    const int nwg = 2048;
    unsigned *d; hipMalloc(&d, nwg * 8);
    hipLaunchKernelGGL(where_kernel, dim3(nwg), dim3(256), 0, st, d);
    hipStreamSynchronize(st);
    std::vector<unsigned> h(nwg * 2);
    hipMemcpy(h.data(), d, nwg * 8, hipMemcpyDeviceToHost);
    std::map<unsigned, std::set<unsigned>> per_xcc;
    for (int i = 0; i < nwg; ++i) {
        unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
        unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 0x7;
        per_xcc[xcc].insert((se << 8) | (sh << 4) | cu);
    }
    size_t tot = 0;
    for (auto &kv : per_xcc) { printf("xcc %u: %zu CUs:", kv.first, kv.second.size()); for (auto c : kv.second) printf(" %x", c); printf("\n"); tot += kv.second.size(); }
    printf("total distinct CUs used: %zu\n", tot);
    return 0;
}
