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
 * original_alignment must lie in [0, ref_len] for every consensus of the read's region and
 * read_len in [0, read_stride]; values outside are clamped on the device (negative to 0,
 * read_len to read_stride), so bad input yields a meaningless score rather than a stray access.
 * Returns 0 or a negative ctc_tc_status (see ctc_tc.h); message in ctc_tc_last_error().
 */
#ifndef CTC_FASTR_H
#define CTC_FASTR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

int ctc_fastr_best_offsets(const uint8_t *d_ref, const uint8_t *d_read_seq, const uint8_t *d_quals, int read_stride,
                           const int32_t *d_read_len, const int32_t *d_original_alignment, int n_regions,
                           const int32_t *h_n_consensus, const int32_t *h_alt_reads, const int32_t *h_ref_lens,
                           int32_t *d_best_score, int32_t *d_best_index, void *stream);

/*
 * Batched pairwise alignment of reads against reference windows: what the reference computes
 * one read at a time, serially on the host, with
 *   SWPairwiseAlignment(reference, referenceLen, read, readLen, match, mismatch, open, extend)
 *       apps/fastr/IndelRealigner.cpp:366-371 (createAndAddAlternateConsensus, --use-sw)
 *   calculateMatrix / calculateCigar      apps/fastr/SWPairwiseAlignment.cpp:128-270, 272-399
 * For pair p (read p against the window ref + ref_off[p] of ref_len[p] bytes) it returns
 * getAlignmentStart2wrt1() in offset[p] and getCigar() as n_ops[p] operations in
 * op_type[p*max_ops ..] ('M', 'I', 'D', 'S') and op_len[p*max_ops ..], in alignment order and
 * identical to the reference's (double arithmetic, same tie rules, soft clips, including the
 * zero-length 'M' it can emit when the back-track starts inside a gap).  n_ops[p] > max_ops:
 * the operations did not fit and none were written; n_ops[p] == -1: ref_len[p] or read_len[p]
 * lies outside 0..max_ref_len / 0..max_read_len.
 *
 * All d_* pointers are DEVICE memory.  max_ref_len <= 32767.  The workspace holds the
 * back-track entries: ctc_fastr_sw_workspace_bytes(n, max_ref_len, max_read_len) bytes let n
 * pairs run in one launch; with less the call issues several launches back to back on `stream`
 * (at least one pair's worth is required).  Enqueues and returns.
 */
size_t ctc_fastr_sw_workspace_bytes(int n_pairs, int max_ref_len, int max_read_len);
int ctc_fastr_sw_align(const uint8_t *d_ref, const int64_t *d_ref_off, const int32_t *d_ref_len,
                       const uint8_t *d_reads, int read_stride, const int32_t *d_read_len, int n_pairs,
                       double w_match, double w_mismatch, double w_open, double w_extend, int max_ref_len,
                       int max_read_len, int32_t *d_offset, int32_t *d_n_ops, uint8_t *d_op_type, int32_t *d_op_len,
                       int max_ops, void *d_workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* CTC_FASTR_H */
