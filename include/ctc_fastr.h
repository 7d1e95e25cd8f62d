/*
 * ctc_fastr.h - C ABI of the MI355X read-vs-consensus scoring of FASTR (apps/fastr of
 * zjing14/CTC), part of libctc_tc.so.
 *
 * The entry point takes the place of the reference's device path
 *   clean_gpu_prep   apps/fastr/IndelRealigner.cpp:671-803   (flatten a batch, upload)
 *   clean_gpu        apps/fastr/IndelRealigner.cpp:805-829   (launch findBestOffset1)
 *   findBestOffset1  apps/fastr/IndelRealigner_kernel.cl:160-297
 * and returns, for every (region, consensus, read), what the reference's CPU function
 * findBestOffset (apps/fastr/IndelRealigner.cpp:502-542) returns: the offset of the read on
 * the consensus with the smallest quality-weighted mismatch sum, ties broken in the order the
 * CPU loop visits offsets (the original alignment first, then 0 .. original-1, then
 * original+1 .. refLen-readLen).  Unlike the reference's OpenCL kernel it scans every offset
 * (the OpenCL kernel stops at 2 x local-size offsets per side) and accepts any read length
 * (the OpenCL path stores reads in fixed 100-byte slots), so it agrees with the CPU path.
 *
 * Batch layout (that of clean_gpu_prep): reads of all regions back to back, read g in bytes
 * [g*read_stride, g*read_stride + read_len[g]) of read_seq and quals (quals are phred values,
 * not ASCII); consensus strings of all regions back to back in ref, lengths in ref_lens
 * (sum of n_consensus entries); results ordered [region][consensus][read], i.e.
 * sum_t n_consensus[t] * alt_reads[t] entries.
 *
 * d_* pointers are DEVICE memory, h_* pointers are HOST memory (the per-region counts are a
 * few hundred bytes; the library turns them into a work table and uploads it).  The call
 * enqueues on `stream` and returns; h_* arrays may be reused as soon as it returns.
 * original_alignment must lie in [0, ref_len] for every consensus of the read's region.
 * Returns 0 or a negative ctc_tc_status (see ctc_tc.h); message in ctc_tc_last_error().
 */
#ifndef CTC_FASTR_H
#define CTC_FASTR_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int ctc_fastr_best_offsets(const uint8_t *d_ref, const uint8_t *d_read_seq, const uint8_t *d_quals, int read_stride,
                           const int32_t *d_read_len, const int32_t *d_original_alignment, int n_regions,
                           const int32_t *h_n_consensus, const int32_t *h_alt_reads, const int32_t *h_ref_lens,
                           int32_t *d_best_score, int32_t *d_best_index, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CTC_FASTR_H */
