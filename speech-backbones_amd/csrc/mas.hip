// mas.hip -- Monotonic Alignment Search on the GPU (gfx950), replacing the reference's only native code:
//   Grad-TTS/model/monotonic_align/core.pyx:9-35 (maximum_path_each), :40-45 (batch loop),
//   wrapper __init__.py:8-23 (value * mask, t_x/t_y from the mask).
//
// One workgroup per batch item.  The DP runs column by column (y); inside a column every row x of the band
// [max(0, t_x+y-t_y), min(t_x, y+1)) is independent, so rows are spread over the lanes and the previous column
// lives in LDS (double-buffered).  Each cell does exactly the reference's arithmetic -- one fp32 max and one fp32
// add -- so DP values, and therefore the path, are bit-identical to the CPU result.  The backtrack only ever
// compares value[index][y-1] < value[index-1][y-1]; that predicate is recorded per cell during the forward
// sweep (1 byte in caller-provided scratch), so the DP matrix itself is never written back and `value` stays
// read-only.
#include "common.h"
#include "kernels.h"

namespace gtts {

__global__ __launch_bounds__(256) void mas_kernel(const float *__restrict__ value, const float *__restrict__ mask,
                                                  const int *__restrict__ t_xs, const int *__restrict__ t_ys,
                                                  int *__restrict__ path, unsigned char *__restrict__ flags, int tx,
                                                  int ty) {
    extern __shared__ float sm[];          // [2][tx]: DP values of the previous / current column
    const int b = blockIdx.x, tid = threadIdx.x;
    const int t_x = t_xs[b], t_y = t_ys[b];
    if (t_x <= 0 || t_y <= 0 || t_x > tx || t_y > ty) return;   // reference behaviour undefined: all-zero path
    const float NEG = -1e9f;
    const size_t base = (size_t)b * tx * ty;
    const float *val = value + base;
    const float *msk = mask ? mask + base : nullptr;
    unsigned char *flg = flags + base;
    float *prev = sm, *cur = sm + tx;

    for (int y = 0; y < t_y; ++y) {
        const int lo = max(0, t_x + y - t_y), hi = min(t_x, y + 1);
        for (int x = lo + tid; x < hi; x += 256) {
            const float pc = (y > 0 && x != y) ? prev[x] : NEG;                 // value[x][y-1]   (core.pyx:18-21)
            float pp;
            if (x == 0) pp = (y == 0) ? 0.f : NEG;                              // core.pyx:22-26
            else pp = prev[x - 1];                                              // value[x-1][y-1] (core.pyx:28)
            const size_t o = (size_t)x * ty + y;
            float v = val[o];
            if (msk) v = __fmul_rn(v, msk[o]);                                  // value * mask (__init__.py:13)
            const float mx = (pc > pp) ? pc : pp;
            cur[x] = __fadd_rn(mx, v);                                          // core.pyx:30
            flg[o] = (x > 0 && x != y && y > 0 && prev[x] < prev[x - 1]) ? 1 : 0;   // backtrack predicate (:34)
        }
        __syncthreads();
        float *t = prev; prev = cur; cur = t;
    }
    __threadfence_block();
    __syncthreads();
    if (tid == 0) {
        int index = t_x - 1;
        for (int y = t_y - 1; y >= 0; --y) {                                    // core.pyx:32-35
            path[base + (size_t)index * ty + y] = 1;
            if (index != 0 && (index == y || (y > 0 && flg[(size_t)index * ty + y]))) index -= 1;
        }
    }
}

hipError_t launch_mas(const float *value, const float *mask, const int *t_x, const int *t_y, int *path,
                      unsigned char *scratch, int b, int tx, int ty, hipStream_t st) {
    hipError_t e = hipMemsetAsync(path, 0, (size_t)b * tx * ty * sizeof(int), st);
    if (e != hipSuccess) return e;
    const size_t smem = (size_t)2 * tx * sizeof(float);
    if (smem > 160 * 1024) return hipErrorInvalidValue;
    if (smem > 48 * 1024) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(&mas_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(mas_kernel, dim3(b), dim3(256), smem, st, value, mask, t_x, t_y, path, scratch, tx, ty);
    return hipGetLastError();
}

}  // namespace gtts
