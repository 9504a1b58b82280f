/* TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 *
 * Plain-C restatement of the reference's only native component, Monotonic Alignment Search:
 *   Grad-TTS/model/monotonic_align/core.pyx:9-35  (maximum_path_each)
 *   Grad-TTS/model/monotonic_align/core.pyx:40-45 (maximum_path_c, loop over the batch)
 * `values` is modified in place exactly as the reference does (the Python wrapper hands it a copy).
 * Pinned against the compiled reference (oracle/_ref, built by oracle/build_ref.py) in
 * tests/test_mas_oracle.py and against tests/golden/mas_*.npz.
 *
 * Defined-behaviour note: the reference reads value[index][-1] when y == 0 and index != 0 (only
 * possible when t_x > t_y); the result of that read never reaches `path`, so it is skipped here.
 * t_x <= 0 or t_y <= 0 (undefined behaviour in the reference: path[-1][y]) yields an all-zero path.
 */
#include <stddef.h>

static void mas_each(int *path, float *value, int t_x, int t_y, int tx_stride, int ty_stride,
                     float max_neg_val)
{
    (void)tx_stride;
    if (t_x <= 0 || t_y <= 0) return;
    /* forward DP over the band x in [max(0, t_x + y - t_y), min(t_x, y + 1))   core.pyx:17-30 */
    for (int y = 0; y < t_y; ++y) {
        int lo = t_x + y - t_y; if (lo < 0) lo = 0;
        int hi = (t_x < y + 1) ? t_x : y + 1;
        for (int x = lo; x < hi; ++x) {
            float v_cur = (x == y) ? max_neg_val : value[(size_t)x * ty_stride + (y - 1)];
            float v_prev;
            if (x == 0) v_prev = (y == 0) ? 0.f : max_neg_val;
            else        v_prev = value[(size_t)(x - 1) * ty_stride + (y - 1)];
            float m = (v_cur > v_prev) ? v_cur : v_prev;   /* Cython max(a,b): a if a > b else b */
            value[(size_t)x * ty_stride + y] = m + value[(size_t)x * ty_stride + y];
        }
    }
    /* backtrack   core.pyx:32-35 (strict '<') */
    int index = t_x - 1;
    for (int y = t_y - 1; y >= 0; --y) {
        path[(size_t)index * ty_stride + y] = 1;
        if (index != 0 && (index == y ||
                           (y > 0 && value[(size_t)index * ty_stride + (y - 1)] <
                                     value[(size_t)(index - 1) * ty_stride + (y - 1)])))
            index -= 1;
    }
}

/* paths, values: [b][tx][ty] contiguous; t_xs, t_ys: [b].  paths must be zero-filled by the caller. */
void mas_oracle_maximum_path(int *paths, float *values, const int *t_xs, const int *t_ys,
                             int b, int tx, int ty, float max_neg_val)
{
    for (int i = 0; i < b; ++i)
        mas_each(paths + (size_t)i * tx * ty, values + (size_t)i * tx * ty, t_xs[i], t_ys[i], tx, ty,
                 max_neg_val);
}
