/* Oracle: compiled twin of oracle/dtw_ref.py.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).
 * Restates reference align.py:5-14 (cumulative cost) and align.py:21-26 (backtrace, first-wins tie
 * order up, left, diag).  float32 arithmetic, exactly one add per cell; built with
 * -O2 -fno-fast-math -ffp-contract=off so the compiler cannot reassociate or fuse. */
#include <math.h>
#include <stddef.h>

int oracle_dtw_align_f32(const float* costs, int n, int m, long stride_i, long stride_j,
                         float* dtw /* n*m scratch, row-major */, int* results /* n */)
{
    if (n <= 0 || m <= 0) return 1;
    for (int j = 0; j < m; ++j) dtw[j] = (j == 0) ? 0.0f : INFINITY;
    for (int i = 1; i < n; ++i) {
        float* row = dtw + (size_t)i * m;
        const float* prev = row - m;
        const float* c = costs + (size_t)i * stride_i;
        row[0] = INFINITY;
        for (int j = 1; j < m; ++j) {
            float a = prev[j], b = row[j - 1], d = prev[j - 1];
            float best = a <= b ? a : b;
            best = best <= d ? best : d;
            row[j] = c[(size_t)j * stride_j] + best;
        }
    }
    for (int i = 0; i < n; ++i) results[i] = 0;
    int i = n - 1, j = m - 1;
    while (i > 0 && j > 0) {
        results[i] = j;
        float up = dtw[(size_t)(i - 1) * m + j], left = dtw[(size_t)i * m + j - 1],
              diag = dtw[(size_t)(i - 1) * m + j - 1];
        if (up <= left && up <= diag) { --i; }
        else if (left <= diag) { --j; }
        else { --i; --j; }
    }
    return 0;
}
