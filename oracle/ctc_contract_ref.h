/*
 * CPU oracle for the tensor-contraction path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load or call this library.  The product (libctc_tc.so) never links it.
 *
 * PARITY UNPINNED: the reference tree (zjing14/CTC) contains no tensor
 * contraction code, no golden vectors and no known-answer tests for this
 * path, so nothing here follows a reference file.  The semantics below are
 * this build's own definition; they were cross-checked once against
 * numpy.einsum (oracle/make_golden.py) and the resulting vectors are
 * committed under tests/golden/.
 *
 * Semantics
 *   spec        "aeim,ebmj->abij": one letter per mode, operands A,B -> C.
 *   class       mode in A,B,C = batch; A,B only = contracted;
 *               A,C only = free-M; B,C only = free-N; anything else invalid.
 *   layout      dense, explicit element strides; default packed with the
 *               LAST mode of each operand fastest (row-major).
 *   value       C[...] = alpha * sum_contracted A[...] * B[...] + beta * C[...]
 *               accumulated in the working precision's wider type
 *               (f64 for f64; f64 for f32/bf16 inputs, rounded once at the end).
 *   generator   element k (packed linear order) = u(splitmix64(seed + k))
 *               mapped to uniform [-1, 1).
 */
#ifndef CTC_CONTRACT_REF_H
#define CTC_CONTRACT_REF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTC_REF_MAX_MODES 16

enum { CTC_REF_F64 = 0, CTC_REF_F32 = 1, CTC_REF_BF16 = 2 };
enum {
    CTC_REF_CLASS_BATCH = 0,
    CTC_REF_CLASS_CONTRACTED = 1,
    CTC_REF_CLASS_FREE_M = 2,
    CTC_REF_CLASS_FREE_N = 3
};

typedef struct {
    int n_modes;                         /* distinct letters, in order of first appearance A,B,C */
    char names[CTC_REF_MAX_MODES];
    int cls[CTC_REF_MAX_MODES];          /* CTC_REF_CLASS_* per distinct mode */
    int na, nb, nc;                      /* operand ranks */
    int a_modes[CTC_REF_MAX_MODES];      /* index into names[] for each position of A */
    int b_modes[CTC_REF_MAX_MODES];
    int c_modes[CTC_REF_MAX_MODES];
    /* canonical GEMM view: batch / M / N in order of appearance in C,
     * contracted in order of appearance in A */
    int n_batch, n_m, n_n, n_k;
    int batch_modes[CTC_REF_MAX_MODES];
    int m_modes[CTC_REF_MAX_MODES];
    int n_modes_list[CTC_REF_MAX_MODES];
    int k_modes[CTC_REF_MAX_MODES];
    /* canonical transposes: position p of the permuted operand takes
     * position perm[p] of the original.  A -> [batch, M, K],
     * B -> [batch, K, N], C(gemm) [batch, M, N] -> C. */
    int perm_a[CTC_REF_MAX_MODES];
    int perm_b[CTC_REF_MAX_MODES];
    int perm_c[CTC_REF_MAX_MODES];
} ctc_ref_classification;

uint64_t ctc_ref_splitmix64(uint64_t x);
void ctc_ref_fill_f64(double *out, int64_t n, uint64_t seed);
void ctc_ref_fill_f32(float *out, int64_t n, uint64_t seed);
void ctc_ref_fill_bf16(uint16_t *out, int64_t n, uint64_t seed);
uint16_t ctc_ref_f32_to_bf16(float f);
float ctc_ref_bf16_to_f32(uint16_t h);

/* returns 0 on success, negative on a malformed spec (message in err) */
int ctc_ref_classify(const char *spec, ctc_ref_classification *out, char *err, size_t errlen);

/* extents are given per distinct mode, aligned with out->names order, through
 * the parallel arrays mode_names/extents (any order).  strides may be NULL
 * (packed, last mode fastest) or per-position element strides of each operand. */
int ctc_ref_contract(const char *spec, int n_ext, const char *mode_names, const int64_t *extents,
                     const int64_t *strides_a, const int64_t *strides_b, const int64_t *strides_c,
                     int dtype, double alpha, const void *A, const void *B, double beta, void *C,
                     int n_threads, char *err, size_t errlen);

/* out[..] with mode order perm: position p of out takes position perm[p] of in.
 * Both packed, last mode fastest.  elem_size in bytes (2, 4 or 8). */
int ctc_ref_permute(const void *in, void *out, int rank, const int64_t *extents_in,
                    const int *perm, int elem_size, int n_threads);

int ctc_ref_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
