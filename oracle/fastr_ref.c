/*
 * CPU oracle for FASTR's read-vs-consensus scoring.  TEST INFRASTRUCTURE ONLY: loaded by
 * tests/, __graft_entry__.smoke() and benchmark cpu_baseline legs, never by the product.
 *
 * Restates, in plain C:
 *   mismatchQualitySumIgnoreCigar   apps/fastr/IndelRealigner.cpp:264-295
 *   findBestOffset                  apps/fastr/IndelRealigner.cpp:502-542
 *   isRegularBase                   apps/fastr/HelperFunctions.cpp:371-387
 * over the flattened batch layout that clean_gpu_prep builds
 * (apps/fastr/IndelRealigner.cpp:671-803): reads of all regions back to back in fixed-size
 * slots, consensus strings of all regions back to back, results ordered
 * [region][consensus][read].  The slot size is a parameter here (the reference hard-codes 100).
 *
 * Pinned against the reference itself: oracle/_ref/libfastr_ref.so (the reference's own
 * functions compiled by oracle/build_ref.sh) on seeded inputs, and the vectors that library
 * produced, committed in tests/golden/fastr_golden.json.
 */
#include <limits.h>
#include <stdint.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define FASTR_MAX_QUAL 99 /* apps/fastr/IndelRealigner.cpp:170 */

static int regular_base(unsigned char c)
{
    switch (c) {
    case 'A': case 'a': case 'C': case 'c': case 'G': case 'g': case 'T': case 't': return 1;
    default: return 0;
    }
}

static long mismatch_quality_sum(const uint8_t *read, const uint8_t *quals, int read_len, const uint8_t *ref,
                                 long ref_len, long ref_index, long quit_above)
{
    int sum = 0;
    for (int r = 0; r < read_len; ++r, ++ref_index) {
        if (ref_index >= ref_len) {
            sum += FASTR_MAX_QUAL;
            if (sum > quit_above) return sum;
        } else {
            unsigned char rc = ref[ref_index], qc = read[r];
            if (!regular_base(qc) || !regular_base(rc)) continue;
            if (qc != rc) {
                sum += (int)(signed char)quals[r];
                if (sum > quit_above) return sum;
            }
        }
    }
    return sum;
}

void ctc_ref_fastr_find_best_offset(const uint8_t *ref, int ref_len, const uint8_t *read, const uint8_t *quals,
                                    int read_len, int original_alignment, int *best_index, int *best_score)
{
    long best = mismatch_quality_sum(read, quals, read_len, ref, ref_len, original_alignment, LONG_MAX);
    int index = original_alignment;
    if (best != 0) {
        for (int i = 0; i < original_alignment && best != 0; ++i) {
            long s = mismatch_quality_sum(read, quals, read_len, ref, ref_len, i, best);
            if (s < best) { best = s; index = i; }
        }
        const int max_start = ref_len - read_len;
        for (int i = original_alignment + 1; i <= max_start && best != 0; ++i) {
            long s = mismatch_quality_sum(read, quals, read_len, ref, ref_len, i, best);
            if (s < best) { best = s; index = i; }
        }
    }
    *best_index = index;
    *best_score = (int)best;
}

/* whole batch; returns the number of (read, consensus) pairs scored */
long ctc_ref_fastr_best_offsets(const uint8_t *ref, const uint8_t *read_seq, const uint8_t *quals, int read_stride,
                                const int32_t *read_len, const int32_t *original_alignment, int n_regions,
                                const int32_t *n_consensus, const int32_t *alt_reads, const int32_t *ref_lens,
                                int32_t *best_score, int32_t *best_index, int n_threads)
{
    long ref_off = 0, read_off = 0, score_off = 0, cons = 0;
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
    for (int t = 0; t < n_regions; ++t) {
        for (int k = 0; k < n_consensus[t]; ++k) {
            const int len = ref_lens[cons];
#pragma omp parallel for schedule(dynamic, 8) num_threads(n_threads)
            for (int b = 0; b < alt_reads[t]; ++b) {
                const long g = read_off + b;
                int idx, sc;
                ctc_ref_fastr_find_best_offset(ref + ref_off, len, read_seq + g * read_stride, quals + g * read_stride,
                                               read_len[g], original_alignment[g], &idx, &sc);
                best_index[score_off + b] = idx;
                best_score[score_off + b] = sc;
            }
            ref_off += len;
            cons += 1;
            score_off += alt_reads[t];
        }
        read_off += alt_reads[t];
    }
    return score_off;
}
