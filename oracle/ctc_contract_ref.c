/*
 * CPU oracle for the tensor-contraction path.  TEST INFRASTRUCTURE ONLY
 * (see ctc_contract_ref.h).  PARITY UNPINNED: the reference tree holds no
 * contraction code or vectors; this is the build's own definition of the
 * semantics, cross-checked against numpy.einsum by oracle/make_golden.py.
 *
 * Deliberately simple: one odometer over the output modes, one over the
 * contracted modes, sequential summation in row-major order of the
 * contracted modes as they appear in A.  OpenMP over output rows only.
 */
#include "ctc_contract_ref.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

uint64_t ctc_ref_splitmix64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

void ctc_ref_fill_f64(double *out, int64_t n, uint64_t seed)
{
    for (int64_t k = 0; k < n; ++k) {
        uint64_t x = ctc_ref_splitmix64(seed + (uint64_t)k);
        out[k] = 2.0 * ((double)(x >> 11) * (1.0 / 9007199254740992.0)) - 1.0;
    }
}

void ctc_ref_fill_f32(float *out, int64_t n, uint64_t seed)
{
    for (int64_t k = 0; k < n; ++k) {
        uint64_t x = ctc_ref_splitmix64(seed + (uint64_t)k);
        out[k] = 2.0f * ((float)(x >> 40) * (1.0f / 16777216.0f)) - 1.0f;
    }
}

uint16_t ctc_ref_f32_to_bf16(float f)
{
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (uint16_t)((u >> 16) | 0x0040u); /* NaN stays NaN */
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

float ctc_ref_bf16_to_f32(uint16_t h)
{
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

void ctc_ref_fill_bf16(uint16_t *out, int64_t n, uint64_t seed)
{
    for (int64_t k = 0; k < n; ++k) {
        uint64_t x = ctc_ref_splitmix64(seed + (uint64_t)k);
        float f = 2.0f * ((float)(x >> 40) * (1.0f / 16777216.0f)) - 1.0f;
        out[k] = ctc_ref_f32_to_bf16(f);
    }
}

static void set_err(char *err, size_t errlen, const char *msg)
{
    if (err && errlen) snprintf(err, errlen, "%s", msg);
}

static int find_mode(const ctc_ref_classification *c, char ch)
{
    for (int i = 0; i < c->n_modes; ++i)
        if (c->names[i] == ch) return i;
    return -1;
}

static int parse_operand(const char **p, char stop1, char stop2, ctc_ref_classification *c,
                         int *modes, int *rank, int may_add)
{
    int n = 0;
    while (**p && **p != stop1 && **p != stop2) {
        char ch = **p;
        if (ch == ' ') { ++*p; continue; }
        if (!((ch >= 'a' && ch <= 'z') || (ch >= 'A' && ch <= 'Z'))) return -1;
        int idx = find_mode(c, ch);
        if (idx < 0) {
            if (!may_add || c->n_modes >= CTC_REF_MAX_MODES) return -2;
            idx = c->n_modes++;
            c->names[idx] = ch;
        }
        for (int j = 0; j < n; ++j)
            if (modes[j] == idx) return -3; /* repeated letter in one operand */
        if (n >= CTC_REF_MAX_MODES) return -2;
        modes[n++] = idx;
        ++*p;
    }
    *rank = n;
    return 0;
}

static int position_of(const int *modes, int rank, int m)
{
    for (int i = 0; i < rank; ++i)
        if (modes[i] == m) return i;
    return -1;
}

int ctc_ref_classify(const char *spec, ctc_ref_classification *c, char *err, size_t errlen)
{
    memset(c, 0, sizeof(*c));
    const char *p = spec;
    if (parse_operand(&p, ',', 0, c, c->a_modes, &c->na, 1) || *p != ',') {
        set_err(err, errlen, "malformed spec: operand A");
        return -1;
    }
    ++p;
    if (parse_operand(&p, '-', '=', c, c->b_modes, &c->nb, 1) || (*p != '-' && *p != '=')) {
        set_err(err, errlen, "malformed spec: operand B");
        return -1;
    }
    ++p;
    if (*p == '>') ++p;
    if (parse_operand(&p, 0, 0, c, c->c_modes, &c->nc, 0)) {
        set_err(err, errlen, "malformed spec: output has an unknown or repeated mode");
        return -1;
    }
    for (int m = 0; m < c->n_modes; ++m) {
        int ia = position_of(c->a_modes, c->na, m) >= 0;
        int ib = position_of(c->b_modes, c->nb, m) >= 0;
        int ic = position_of(c->c_modes, c->nc, m) >= 0;
        if (ia && ib && ic) c->cls[m] = CTC_REF_CLASS_BATCH;
        else if (ia && ib) c->cls[m] = CTC_REF_CLASS_CONTRACTED;
        else if (ia && ic) c->cls[m] = CTC_REF_CLASS_FREE_M;
        else if (ib && ic) c->cls[m] = CTC_REF_CLASS_FREE_N;
        else {
            set_err(err, errlen, "mode appears in one operand only");
            return -2;
        }
    }
    /* canonical orders */
    for (int i = 0; i < c->nc; ++i) {
        int m = c->c_modes[i];
        if (c->cls[m] == CTC_REF_CLASS_BATCH) c->batch_modes[c->n_batch++] = m;
        else if (c->cls[m] == CTC_REF_CLASS_FREE_M) c->m_modes[c->n_m++] = m;
        else c->n_modes_list[c->n_n++] = m;
    }
    for (int i = 0; i < c->na; ++i) {
        int m = c->a_modes[i];
        if (c->cls[m] == CTC_REF_CLASS_CONTRACTED) c->k_modes[c->n_k++] = m;
    }
    /* canonical transposes */
    int q = 0;
    for (int i = 0; i < c->n_batch; ++i) c->perm_a[q++] = position_of(c->a_modes, c->na, c->batch_modes[i]);
    for (int i = 0; i < c->n_m; ++i) c->perm_a[q++] = position_of(c->a_modes, c->na, c->m_modes[i]);
    for (int i = 0; i < c->n_k; ++i) c->perm_a[q++] = position_of(c->a_modes, c->na, c->k_modes[i]);
    q = 0;
    for (int i = 0; i < c->n_batch; ++i) c->perm_b[q++] = position_of(c->b_modes, c->nb, c->batch_modes[i]);
    for (int i = 0; i < c->n_k; ++i) c->perm_b[q++] = position_of(c->b_modes, c->nb, c->k_modes[i]);
    for (int i = 0; i < c->n_n; ++i) c->perm_b[q++] = position_of(c->b_modes, c->nb, c->n_modes_list[i]);
    /* gemm-ordered C is [batch.., M.., N..]; perm_c maps it back to C's own order */
    {
        int gemm_order[CTC_REF_MAX_MODES];
        q = 0;
        for (int i = 0; i < c->n_batch; ++i) gemm_order[q++] = c->batch_modes[i];
        for (int i = 0; i < c->n_m; ++i) gemm_order[q++] = c->m_modes[i];
        for (int i = 0; i < c->n_n; ++i) gemm_order[q++] = c->n_modes_list[i];
        for (int i = 0; i < c->nc; ++i) c->perm_c[i] = position_of(gemm_order, q, c->c_modes[i]);
    }
    return 0;
}

static double load_elem(const void *p, int64_t off, int dtype)
{
    switch (dtype) {
    case CTC_REF_F64: return ((const double *)p)[off];
    case CTC_REF_F32: return (double)((const float *)p)[off];
    default: return (double)ctc_ref_bf16_to_f32(((const uint16_t *)p)[off]);
    }
}

static void store_elem(void *p, int64_t off, int dtype, double v)
{
    switch (dtype) {
    case CTC_REF_F64: ((double *)p)[off] = v; break;
    case CTC_REF_F32: ((float *)p)[off] = (float)v; break;
    default: ((uint16_t *)p)[off] = ctc_ref_f32_to_bf16((float)v); break;
    }
}

static void packed_strides(int rank, const int *modes, const int64_t *ext, int64_t *st)
{
    int64_t s = 1;
    for (int i = rank - 1; i >= 0; --i) {
        st[i] = s;
        s *= ext[modes[i]];
    }
}

int ctc_ref_contract(const char *spec, int n_ext, const char *mode_names, const int64_t *extents,
                     const int64_t *strides_a, const int64_t *strides_b, const int64_t *strides_c,
                     int dtype, double alpha, const void *A, const void *B, double beta, void *C,
                     int n_threads, char *err, size_t errlen)
{
    ctc_ref_classification c;
    int rc = ctc_ref_classify(spec, &c, err, errlen);
    if (rc) return rc;

    int64_t ext[CTC_REF_MAX_MODES];
    for (int m = 0; m < c.n_modes; ++m) {
        ext[m] = -1;
        for (int j = 0; j < n_ext; ++j)
            if (mode_names[j] == c.names[m]) ext[m] = extents[j];
        if (ext[m] < 0) {
            set_err(err, errlen, "missing extent for a mode");
            return -3;
        }
    }
    int64_t sa[CTC_REF_MAX_MODES], sb[CTC_REF_MAX_MODES], sc[CTC_REF_MAX_MODES];
    if (strides_a) memcpy(sa, strides_a, sizeof(int64_t) * c.na); else packed_strides(c.na, c.a_modes, ext, sa);
    if (strides_b) memcpy(sb, strides_b, sizeof(int64_t) * c.nb); else packed_strides(c.nb, c.b_modes, ext, sb);
    if (strides_c) memcpy(sc, strides_c, sizeof(int64_t) * c.nc); else packed_strides(c.nc, c.c_modes, ext, sc);

    /* per distinct mode: stride in each operand (0 when absent) */
    int64_t ma[CTC_REF_MAX_MODES] = {0}, mb[CTC_REF_MAX_MODES] = {0}, mc[CTC_REF_MAX_MODES] = {0};
    for (int i = 0; i < c.na; ++i) ma[c.a_modes[i]] = sa[i];
    for (int i = 0; i < c.nb; ++i) mb[c.b_modes[i]] = sb[i];
    for (int i = 0; i < c.nc; ++i) mc[c.c_modes[i]] = sc[i];

    /* output odometer: all C modes but the last; inner run: the last C mode */
    int n_outer = c.nc > 0 ? c.nc - 1 : 0;
    int inner_mode = c.nc > 0 ? c.c_modes[c.nc - 1] : -1;
    int64_t inner_ext = inner_mode >= 0 ? ext[inner_mode] : 1;
    int64_t n_rows = 1;
    for (int i = 0; i < n_outer; ++i) n_rows *= ext[c.c_modes[i]];
    int64_t n_red = 1;
    for (int i = 0; i < c.n_k; ++i) n_red *= ext[c.k_modes[i]];
    if (n_rows == 0 || inner_ext == 0) return 0;

    int64_t ia_in = inner_mode >= 0 ? ma[inner_mode] : 0;
    int64_t ib_in = inner_mode >= 0 ? mb[inner_mode] : 0;
    int64_t ic_in = inner_mode >= 0 ? mc[inner_mode] : 0;

#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
    int failed = 0;
#pragma omp parallel num_threads(n_threads)
    {
        double *acc = (double *)malloc(sizeof(double) * (size_t)inner_ext);
        if (!acc) {
#pragma omp atomic write
            failed = 1;
        }
#pragma omp for schedule(static)
        for (int64_t row = 0; row < n_rows; ++row) {
            if (!acc) continue;
            int64_t offa = 0, offb = 0, offc = 0, r = row;
            for (int i = n_outer - 1; i >= 0; --i) {
                int m = c.c_modes[i];
                int64_t idx = r % ext[m];
                r /= ext[m];
                offa += idx * ma[m];
                offb += idx * mb[m];
                offc += idx * mc[m];
            }
            for (int64_t x = 0; x < inner_ext; ++x) acc[x] = 0.0;
            int64_t kidx[CTC_REF_MAX_MODES] = {0};
            for (int64_t t = 0; t < n_red; ++t) {
                int64_t ka = offa, kb = offb;
                for (int i = 0; i < c.n_k; ++i) {
                    ka += kidx[i] * ma[c.k_modes[i]];
                    kb += kidx[i] * mb[c.k_modes[i]];
                }
                if (dtype == CTC_REF_F64) {
                    const double *pa = (const double *)A + ka;
                    const double *pb = (const double *)B + kb;
                    if (ia_in == 0) {
                        double av = pa[0];
                        for (int64_t x = 0; x < inner_ext; ++x) acc[x] += av * pb[x * ib_in];
                    } else if (ib_in == 0) {
                        double bv = pb[0];
                        for (int64_t x = 0; x < inner_ext; ++x) acc[x] += pa[x * ia_in] * bv;
                    } else {
                        for (int64_t x = 0; x < inner_ext; ++x) acc[x] += pa[x * ia_in] * pb[x * ib_in];
                    }
                } else {
                    for (int64_t x = 0; x < inner_ext; ++x)
                        acc[x] += load_elem(A, ka + x * ia_in, dtype) * load_elem(B, kb + x * ib_in, dtype);
                }
                for (int i = c.n_k - 1; i >= 0; --i) {
                    if (++kidx[i] < ext[c.k_modes[i]]) break;
                    kidx[i] = 0;
                }
            }
            for (int64_t x = 0; x < inner_ext; ++x) {
                int64_t o = offc + x * ic_in;
                double v = alpha * acc[x];
                if (beta != 0.0) v += beta * load_elem(C, o, dtype);
                store_elem(C, o, dtype, v);
            }
        }
        free(acc);
    }
    if (failed) {
        set_err(err, errlen, "out of memory");
        return -4;
    }
    return 0;
}

int ctc_ref_permute(const void *in, void *out, int rank, const int64_t *extents_in,
                    const int *perm, int elem_size, int n_threads)
{
    if (rank <= 0 || rank > CTC_REF_MAX_MODES) return -1;
    int64_t st_in[CTC_REF_MAX_MODES], ext_out[CTC_REF_MAX_MODES], st_src[CTC_REF_MAX_MODES];
    int seen[CTC_REF_MAX_MODES] = {0};
    int64_t s = 1;
    for (int i = rank - 1; i >= 0; --i) {
        st_in[i] = s;
        s *= extents_in[i];
    }
    int64_t total = s;
    for (int p = 0; p < rank; ++p) {
        if (perm[p] < 0 || perm[p] >= rank || seen[perm[p]]) return -2;
        seen[perm[p]] = 1;
        ext_out[p] = extents_in[perm[p]];
        st_src[p] = st_in[perm[p]];
    }
    if (elem_size != 2 && elem_size != 4 && elem_size != 8) return -3;
#ifdef _OPENMP
    if (n_threads <= 0) n_threads = omp_get_max_threads();
#else
    n_threads = 1;
#endif
    int64_t inner = ext_out[rank - 1];
    int64_t rows = inner ? total / inner : 0;
#pragma omp parallel for schedule(static) num_threads(n_threads)
    for (int64_t row = 0; row < rows; ++row) {
        int64_t r = row, src = 0;
        for (int p = rank - 2; p >= 0; --p) {
            src += (r % ext_out[p]) * st_src[p];
            r /= ext_out[p];
        }
        int64_t dst = row * inner;
        int64_t ss = st_src[rank - 1];
        if (elem_size == 8) {
            const uint64_t *pi = (const uint64_t *)in;
            uint64_t *po = (uint64_t *)out;
            for (int64_t x = 0; x < inner; ++x) po[dst + x] = pi[src + x * ss];
        } else if (elem_size == 4) {
            const uint32_t *pi = (const uint32_t *)in;
            uint32_t *po = (uint32_t *)out;
            for (int64_t x = 0; x < inner; ++x) po[dst + x] = pi[src + x * ss];
        } else {
            const uint16_t *pi = (const uint16_t *)in;
            uint16_t *po = (uint16_t *)out;
            for (int64_t x = 0; x < inner; ++x) po[dst + x] = pi[src + x * ss];
        }
    }
    return 0;
}

int ctc_ref_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
