/* The CPU oracle (oracle/ctc_contract_ref.c) under AddressSanitizer + UndefinedBehaviorSanitizer: seeded random
 * specs on exactly-sized heap buffers, integer-valued data, against an independent evaluation over the whole index
 * space written here.  The oracle is the checker of every GPU parity test; this checks the checker's memory use.
 *   gcc -std=c11 -O1 -g -fopenmp -fsanitize=address,undefined -fno-sanitize-recover=all -Ioracle \
 *       tests/oracle_sanitize.c oracle/ctc_contract_ref.c -lm -o oracle_sanitize && ./oracle_sanitize [cases] [seed] */
#include "ctc_contract_ref.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static uint64_t state = 88172645463325252ull;
static uint64_t rnd(void)
{
    state ^= state << 13;
    state ^= state >> 7;
    state ^= state << 17;
    return state;
}
static int pick(const int *v, int n) { return v[rnd() % (uint64_t)n]; }
static void shuffle(char *s, int n)
{
    for (int i = n - 1; i > 0; --i) {
        int j = (int)(rnd() % (uint64_t)(i + 1));
        char t = s[i];
        s[i] = s[j];
        s[j] = t;
    }
}

static int failures = 0;

static int64_t extent_of(const char *names, const int64_t *ext, int n, char ch)
{
    for (int i = 0; i < n; ++i)
        if (names[i] == ch) return ext[i];
    return -1;
}

static int64_t numel(const char *modes, const char *names, const int64_t *ext, int n)
{
    int64_t v = 1;
    for (const char *p = modes; *p; ++p) v *= extent_of(names, ext, n, *p);
    return v;
}

/* packed offset of the index tuple idx[] (per name) inside an operand with mode string `modes` */
static int64_t offset_in(const char *modes, const char *names, const int64_t *ext, const int64_t *idx, int n)
{
    int64_t off = 0;
    for (const char *p = modes; *p; ++p)
        for (int i = 0; i < n; ++i)
            if (names[i] == *p) off = off * ext[i] + idx[i];
    return off;
}

static void one_case(int c)
{
    static const int counts[] = {0, 1, 1, 2, 2, 3}, batches[] = {0, 0, 1, 2}, small[] = {1, 2, 3, 4, 5, 7};
    char names[13] = "abcdefghijkl";
    shuffle(names, 12);
    const int n_fa = pick(counts, 6), n_fb = pick(counts, 6), n_k = pick(counts, 6), n_bt = pick(batches, 4);
    char a[16] = "", b[16] = "", cm[16] = "";
    int at = 0;
    char fa[4] = "", fb[4] = "", ks[4] = "", bs[4] = "";
    memcpy(fa, names + at, (size_t)n_fa); at += n_fa;
    memcpy(fb, names + at, (size_t)n_fb); at += n_fb;
    memcpy(ks, names + at, (size_t)n_k); at += n_k;
    memcpy(bs, names + at, (size_t)n_bt); at += n_bt;
    strcat(a, fa); strcat(a, ks); strcat(a, bs);
    strcat(b, fb); strcat(b, ks); strcat(b, bs);
    strcat(cm, fa); strcat(cm, fb); strcat(cm, bs);
    if (!a[0] || !b[0]) return;
    shuffle(a, (int)strlen(a));
    shuffle(b, (int)strlen(b));
    shuffle(cm, (int)strlen(cm));
    char used[16];
    int n = 0;
    for (int i = 0; i < at; ++i) used[n++] = names[i];
    used[n] = 0;
    int64_t ext[16];
    int64_t space = 1;
    for (int i = 0; i < n; ++i) {
        ext[i] = pick(small, 6);
        space *= ext[i];
    }
    if (space > 200000) return;
    char spec[64];
    snprintf(spec, sizeof spec, "%s,%s->%s", a, b, cm);
    const int64_t na = numel(a, used, ext, n), nb = numel(b, used, ext, n), nc = numel(cm, used, ext, n);
    double *A = malloc((size_t)na * sizeof(double)), *B = malloc((size_t)nb * sizeof(double));
    double *C = malloc((size_t)nc * sizeof(double)), *W = malloc((size_t)nc * sizeof(double));
    for (int64_t i = 0; i < na; ++i) A[i] = (double)((i * 7 + 3) % 11 - 5);
    for (int64_t i = 0; i < nb; ++i) B[i] = (double)((i * 5 + 8) % 13 - 6);
    for (int64_t i = 0; i < nc; ++i) C[i] = W[i] = (double)(i % 9 - 4);
    const double alpha = (double)(1 + (int)(rnd() % 3)), beta = (double)((int)(rnd() % 3) - 1);
    char err[256] = "";
    const int threads = 1 + (int)(rnd() % 4);
    int rc = ctc_ref_contract(spec, n, used, ext, NULL, NULL, NULL, CTC_REF_F64, alpha, A, B, beta, C, threads, err, sizeof err);
    if (rc != 0) {
        fprintf(stderr, "FAIL #%d %s: %s\n", c, spec, err);
        ++failures;
    } else {
        for (int64_t i = 0; i < nc; ++i) W[i] *= beta;
        int64_t idx[16] = {0};
        for (;;) {
            W[offset_in(cm, used, ext, idx, n)] += alpha * A[offset_in(a, used, ext, idx, n)] * B[offset_in(b, used, ext, idx, n)];
            int d = 0;
            for (; d < n; ++d) {
                if (++idx[d] < ext[d]) break;
                idx[d] = 0;
            }
            if (d == n) break;
        }
        for (int64_t i = 0; i < nc; ++i)
            if (W[i] != C[i]) {
                fprintf(stderr, "FAIL #%d %s: element %lld is %g, expected %g\n", c, spec, (long long)i, C[i], W[i]);
                ++failures;
                break;
            }
    }
    /* the same through f32 and bf16 buffers: exact for these small integers as long as the sums stay below 2^8 */
    float *Af = malloc((size_t)na * sizeof(float)), *Bf = malloc((size_t)nb * sizeof(float)), *Cf = malloc((size_t)nc * sizeof(float));
    for (int64_t i = 0; i < na; ++i) Af[i] = (float)A[i];
    for (int64_t i = 0; i < nb; ++i) Bf[i] = (float)B[i];
    for (int64_t i = 0; i < nc; ++i) Cf[i] = 0.f;
    rc = ctc_ref_contract(spec, n, used, ext, NULL, NULL, NULL, CTC_REF_F32, 1.0, Af, Bf, 0.0, Cf, threads, err, sizeof err);
    if (rc != 0) {
        fprintf(stderr, "FAIL #%d %s f32: %s\n", c, spec, err);
        ++failures;
    }
    uint16_t *Ah = malloc((size_t)na * 2), *Bh = malloc((size_t)nb * 2), *Ch = malloc((size_t)nc * 2);
    for (int64_t i = 0; i < na; ++i) Ah[i] = ctc_ref_f32_to_bf16(Af[i]);
    for (int64_t i = 0; i < nb; ++i) Bh[i] = ctc_ref_f32_to_bf16(Bf[i]);
    memset(Ch, 0, (size_t)nc * 2);
    rc = ctc_ref_contract(spec, n, used, ext, NULL, NULL, NULL, CTC_REF_BF16, 1.0, Ah, Bh, 0.0, Ch, threads, err, sizeof err);
    if (rc != 0) {
        fprintf(stderr, "FAIL #%d %s bf16: %s\n", c, spec, err);
        ++failures;
    }
    /* a permutation of A on exactly-sized buffers, there and back */
    const int ra = (int)strlen(a);
    int perm[16], inv[16];
    int64_t ea[16], eo[16];
    for (int i = 0; i < ra; ++i) {
        perm[i] = i;
        ea[i] = extent_of(used, ext, n, a[i]);
    }
    for (int i = ra - 1; i > 0; --i) {
        int j = (int)(rnd() % (uint64_t)(i + 1)), t = perm[i];
        perm[i] = perm[j];
        perm[j] = t;
    }
    for (int i = 0; i < ra; ++i) {
        inv[perm[i]] = i;
        eo[i] = ea[perm[i]];
    }
    double *P = malloc((size_t)na * sizeof(double)), *Q = malloc((size_t)na * sizeof(double));
    if (ctc_ref_permute(A, P, ra, ea, perm, 8, threads) != 0 || ctc_ref_permute(P, Q, ra, eo, inv, 8, threads) != 0 ||
        memcmp(A, Q, (size_t)na * sizeof(double)) != 0) {
        fprintf(stderr, "FAIL #%d permutation of %s and back\n", c, a);
        ++failures;
    }
    free(A); free(B); free(C); free(W); free(Af); free(Bf); free(Cf); free(Ah); free(Bh); free(Ch); free(P); free(Q);
}

int main(int argc, char **argv)
{
    const int cases = argc > 1 ? atoi(argv[1]) : 500;
    if (argc > 2) state ^= (uint64_t)atoll(argv[2]) * 0x9E3779B97F4A7C15ull;
    for (int c = 0; c < cases; ++c) one_case(c);
    /* malformed specs come back as errors */
    static const char *bad[] = {"", "ab", "ab,->", "ab,bc->ad", "aa,ab->b", "ab,bc->acc", "ab,bc", "ab,bc->ac->x", ",->", "abcdefghijklmnopq,a->b"};
    for (size_t i = 0; i < sizeof bad / sizeof bad[0]; ++i) {
        ctc_ref_classification cl;
        char err[128];
        (void)ctc_ref_classify(bad[i], &cl, err, sizeof err);
        (void)ctc_ref_classify(bad[i], &cl, err, 1);
    }
    /* generators write exactly n elements */
    for (int n = 0; n < 40; n += 13) {
        double *d = malloc((size_t)(n ? n : 1) * 8);
        float *f = malloc((size_t)(n ? n : 1) * 4);
        uint16_t *h = malloc((size_t)(n ? n : 1) * 2);
        ctc_ref_fill_f64(d, n, 1);
        ctc_ref_fill_f32(f, n, 2);
        ctc_ref_fill_bf16(h, n, 3);
        free(d); free(f); free(h);
    }
    printf("%d cases: %d failures\n", cases, failures);
    return failures ? 1 : 0;
}
