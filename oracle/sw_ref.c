/* CPU restatement of the reference's pairwise aligner.  TEST INFRASTRUCTURE ONLY: nothing
 * under ctc_amd/ may include, link or call this file.
 *
 * Follows apps/fastr/SWPairwiseAlignment.cpp:
 *   calculateMatrix   :128-270   score matrix with linear-in-length gap cost (open + (k-1) extend),
 *                                best vertical gap kept per column, best horizontal gap per row;
 *                                row 0 and column 0 stay 0, scores are NOT clamped at 0
 *   calculateCigar    :272-399   start cell = best of last column (>=, top to bottom) then last row
 *                                (equal-or-better AND closer to the diagonal), back-track through
 *                                the recorded gap lengths, soft clips at both ends (DO_SOFTCLIP, :59)
 *   wd / wk           :423-431
 * a = sequence 1 (the reference window), b = sequence 2 (the read).  Arithmetic is double, as
 * there, so any weights give identical results.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* Returns the number of CIGAR operations (also when it exceeds max_ops: then only the first
 * max_ops are written), -1 when out of memory.  Operations come in alignment order. */
int ctc_ref_sw_align(const unsigned char *a, int n, const unsigned char *b, int m, double w_match,
                     double w_mismatch, double w_open, double w_extend, int *alignment_offset,
                     char *op_type, int *op_len, int max_ops)
{
    const int W = m + 1;
    const size_t cells = (size_t)(n + 1) * (size_t)W;
    double *sw = (double *)calloc(cells, sizeof(double));
    int *bt = (int *)calloc(cells, sizeof(int));
    double *best_gap_v = (double *)malloc(sizeof(double) * (size_t)(m + 2));
    double *best_gap_h = (double *)malloc(sizeof(double) * (size_t)(n + 2));
    int *gap_size_v = (int *)calloc((size_t)(m + 2), sizeof(int));
    int *gap_size_h = (int *)calloc((size_t)(n + 2), sizeof(int));
    if (!sw || !bt || !best_gap_v || !best_gap_h || !gap_size_v || !gap_size_h) {
        free(sw); free(bt); free(best_gap_v); free(best_gap_h); free(gap_size_v); free(gap_size_h);
        return -1;
    }
    for (int j = 0; j < m + 2; ++j) best_gap_v[j] = -1.0e40;
    for (int i = 0; i < n + 2; ++i) best_gap_h[i] = -1.0e40;
    const double cutoff = -1e100;

    for (int i = 1; i <= n; ++i) {
        const unsigned char a_base = a[i - 1];
        for (int j = 1; j <= m; ++j) {
            const double step_diag = sw[(size_t)(i - 1) * W + (j - 1)] + (a_base == b[j - 1] ? w_match : w_mismatch);
            double prev_gap = sw[(size_t)(i - 1) * W + j] + w_open;
            best_gap_v[j] += w_extend;
            if (prev_gap > best_gap_v[j]) { best_gap_v[j] = prev_gap; gap_size_v[j] = 1; }
            else gap_size_v[j]++;
            const double step_down = best_gap_v[j];
            const int kd = gap_size_v[j];
            prev_gap = sw[(size_t)i * W + (j - 1)] + w_open;
            best_gap_h[i] += w_extend;
            if (prev_gap > best_gap_h[i]) { best_gap_h[i] = prev_gap; gap_size_h[i] = 1; }
            else gap_size_h[i]++;
            const double step_right = best_gap_h[i];
            const int ki = gap_size_h[i];
            double v;
            int t;
            if (step_down > step_right) {
                if (step_down > step_diag) { v = step_down; t = kd; }
                else { v = step_diag; t = 0; }
            } else {
                if (step_right > step_diag) { v = step_right; t = -ki; }
                else { v = step_diag; t = 0; }
            }
            sw[(size_t)i * W + j] = v > cutoff ? v : cutoff;
            bt[(size_t)i * W + j] = t;
        }
    }

    int p1 = 0, p2 = 0, segment_length = 0;
    double maxscore = 0.0;
    for (int i = 1; i <= n; ++i) {
        const double s = sw[(size_t)i * W + m];
        if (s >= maxscore) { p1 = i; p2 = m; maxscore = s; }
    }
    for (int j = 1; j <= m; ++j) {
        const double s = sw[(size_t)n * W + j];
        if ((s > maxscore || s == maxscore) && abs(n - j) < abs(p1 - p2)) {
            p1 = n; p2 = j; maxscore = s;
            segment_length = m - j;
        }
    }

    /* collected back to front, like the reference, then reversed */
    int cap = n + m + 4, cnt = 0;
    char *ty = (char *)malloc((size_t)cap);
    int *ln = (int *)malloc(sizeof(int) * (size_t)cap);
    if (!ty || !ln) { free(ty); free(ln); free(sw); free(bt); free(best_gap_v); free(best_gap_h); free(gap_size_v); free(gap_size_h); return -1; }
    if (segment_length > 0) { ty[cnt] = 'S'; ln[cnt++] = segment_length; segment_length = 0; }
    int state = 0; /* 0 M, 1 I, 2 D */
    static const char names[3] = {'M', 'I', 'D'};
    do {
        const int btr = bt[(size_t)p1 * W + p2];
        int new_state, step = 1;
        if (btr > 0) { new_state = 2; step = btr; }
        else if (btr < 0) { new_state = 1; step = -btr; }
        else new_state = 0;
        if (new_state == 0) { p1--; p2--; }
        else if (new_state == 1) p2 -= step;
        else p1 -= step;
        if (new_state == state) segment_length += step;
        else {
            ty[cnt] = names[state]; ln[cnt++] = segment_length;
            segment_length = step;
            state = new_state;
        }
    } while (p1 > 0 && p2 > 0);
    ty[cnt] = names[state]; ln[cnt++] = segment_length;
    if (p2 > 0) { ty[cnt] = 'S'; ln[cnt++] = p2; }
    *alignment_offset = p1;
    for (int k = 0; k < cnt && k < max_ops; ++k) {
        op_type[k] = ty[cnt - 1 - k];
        op_len[k] = ln[cnt - 1 - k];
    }
    free(ty); free(ln);
    free(sw); free(bt); free(best_gap_v); free(best_gap_h); free(gap_size_v); free(gap_size_h);
    return cnt;
}

/* Batch form with the layout of the device entry point: pair p aligns read p against reference
 * window ref + ref_off[p] of length ref_len[p]. */
long ctc_ref_sw_align_batch(const unsigned char *ref, const long long *ref_off, const int *ref_len,
                            const unsigned char *reads, int read_stride, const int *read_len, int n_pairs,
                            double w_match, double w_mismatch, double w_open, double w_extend, int *offsets,
                            int *n_ops, char *op_type, int *op_len, int max_ops, int threads)
{
    long failures = 0;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : failures)
#endif
    for (int p = 0; p < n_pairs; ++p) {
        const int c = ctc_ref_sw_align(ref + ref_off[p], ref_len[p], reads + (size_t)p * read_stride, read_len[p],
                                       w_match, w_mismatch, w_open, w_extend, &offsets[p],
                                       op_type + (size_t)p * max_ops, op_len + (size_t)p * max_ops, max_ops);
        n_ops[p] = c;
        if (c < 0) failures++;
    }
    (void)threads;
    return failures;
}
