// Host-side check of the planners under AddressSanitizer + UndefinedBehaviorSanitizer (no GPU, no HIP):
// seeded random contraction specs and permutations go through build_plan / shard_plan / plan_permute, and
// the offset tables they produce are EXECUTED on the host in exact integer arithmetic against a direct
// evaluation over the index space, so an offset outside an operand or a wrong table entry shows either as a
// sanitizer report or as a mismatch.
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all -Ictc_amd/csrc \
//       tools/planner_sanitize.cpp ctc_amd/csrc/planner.cpp ctc_amd/csrc/permute_plan.cpp -o /tmp/planner_sanitize
//   /tmp/planner_sanitize [cases] [seed]
#include "permute_plan.hpp"
#include "planner.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <set>
#include <string>
#include <utility>
#include <vector>

using ctc::Plan;

static std::mt19937_64 rng;
static int pick(std::initializer_list<int> v) { return *(v.begin() + rng() % v.size()); }
static int64_t val(int64_t i, int salt) { return (int64_t)((i * 7 + salt) % 11) - 5; }

static int failures = 0, ran_contractions = 0, ran_shards = 0, ran_permutations = 0, ran_padded = 0, took_skinny = 0, took_vec = 0;
#define EXPECT(cond, ...)                        \
    do {                                         \
        if (!(cond)) {                           \
            ++failures;                          \
            fprintf(stderr, "FAIL %s: ", #cond); \
            fprintf(stderr, __VA_ARGS__);        \
            fprintf(stderr, "\n");               \
            return;                              \
        }                                        \
    } while (0)

// C computed through the plan's tables (what the kernels do)
static void by_tables(const Plan &p, const std::vector<int64_t> &A, const std::vector<int64_t> &B, std::vector<int64_t> &C)
{
    const std::vector<int64_t> &ga = p.swapped ? B : A, &gb = p.swapped ? A : B;
    for (int64_t b = 0; b < p.batch; ++b)
        for (int64_t m = 0; m < p.M; ++m)
            for (int64_t n = 0; n < p.N; ++n) {
                int64_t s = 0;
                for (int64_t k = 0; k < p.K; ++k)
                    s += ga.at((size_t)(p.bat_a[b] + p.row_a[m] + p.k_a[k])) * gb.at((size_t)(p.bat_b[b] + p.col_b[n] + p.k_b[k]));
                C.at((size_t)(p.bat_c[b] + p.row_c[m] + p.col_c[n])) += s;
            }
}

// C computed over the index space of all modes with the modes' strides
static void direct(const Plan &p, const std::vector<int64_t> &A, const std::vector<int64_t> &B, std::vector<int64_t> &C)
{
    const size_t nm = p.modes.size();
    std::vector<int64_t> idx(nm, 0);
    for (;;) {
        int64_t oa = 0, ob = 0, oc = 0;
        for (size_t i = 0; i < nm; ++i) {
            oa += idx[i] * p.modes[i].sa;
            ob += idx[i] * p.modes[i].sb;
            oc += idx[i] * p.modes[i].sc;
        }
        C.at((size_t)oc) += A.at((size_t)oa) * B.at((size_t)ob);
        size_t d = 0;
        for (; d < nm; ++d) {
            if (++idx[d] < p.modes[d].extent) break;
            idx[d] = 0;
        }
        if (d == nm) break;
    }
}

static void check_flags(const Plan &p, const std::string &what)
{
    auto affine = [&](int g) {
        if (p.K % g) return false;
        for (int64_t k = 0; k < p.K; ++k) {
            const int64_t k0 = k / g * g;
            if (p.k_a[k] != p.k_a[k0] + (k - k0) * p.ka_in || p.k_b[k] != p.k_b[k0] + (k - k0) * p.kb_in) return false;
        }
        return true;
    };
    if (p.k4_affine) EXPECT(affine(4), "%s: k4_affine", what.c_str());
    if (p.k8_affine) EXPECT(affine(8), "%s: k8_affine", what.c_str());
    if (p.k16_affine) EXPECT(affine(16), "%s: k16_affine", what.c_str());
    if (p.k32_affine) EXPECT(affine(32), "%s: k32_affine", what.c_str());
    if (p.k_run > 0) EXPECT(affine((int)p.k_run), "%s: k_run %lld", what.c_str(), (long long)p.k_run);
    if (p.ka_stride)
        for (int64_t k = 0; k < p.K; ++k) EXPECT(p.k_a[k] == k * p.ka_stride, "%s: ka_stride", what.c_str());
    if (p.kb_stride)
        for (int64_t k = 0; k < p.K; ++k) EXPECT(p.k_b[k] == k * p.kb_stride, "%s: kb_stride", what.c_str());
    EXPECT((int64_t)p.row_a.size() == p.M && (int64_t)p.col_b.size() == p.N && (int64_t)p.k_a.size() == p.K &&
               (int64_t)p.k_b.size() == p.K && (int64_t)p.row_c.size() == p.M && (int64_t)p.col_c.size() == p.N,
           "%s: table sizes", what.c_str());
    if (p.M > 0) EXPECT(p.max_row_a == *std::max_element(p.row_a.begin(), p.row_a.end()), "%s: max_row_a", what.c_str());
    if (p.N > 0) EXPECT(p.max_col_b == *std::max_element(p.col_b.begin(), p.col_b.end()), "%s: max_col_b", what.c_str());
    // vector layouts: the elements of a vector are adjacent in memory
    const int va = p.vec_a, vb = p.vec_b;
    if (p.layout_a == ctc::kLayoutKVec && va > 1)
        for (int64_t k = 0; k + va <= p.K; k += va)
            for (int i = 1; i < va; ++i) EXPECT(p.k_a[k + i] == p.k_a[k] + i, "%s: k_vec A at k %lld", what.c_str(), (long long)k);
    if (p.layout_a == ctc::kLayoutOuterVec && va > 1)
        for (int64_t m = 0; m + va <= p.M; m += va)
            for (int i = 1; i < va; ++i) EXPECT(p.row_a[m + i] == p.row_a[m] + i, "%s: outer_vec A at m %lld", what.c_str(), (long long)m);
    if (p.layout_b == ctc::kLayoutKVec && vb > 1)
        for (int64_t k = 0; k + vb <= p.K; k += vb)
            for (int i = 1; i < vb; ++i) EXPECT(p.k_b[k + i] == p.k_b[k] + i, "%s: k_vec B at k %lld", what.c_str(), (long long)k);
    if (p.layout_b == ctc::kLayoutOuterVec && vb > 1)
        for (int64_t n = 0; n + vb <= p.N; n += vb)
            for (int i = 1; i < vb; ++i) EXPECT(p.col_b[n + i] == p.col_b[n] + i, "%s: outer_vec B at n %lld", what.c_str(), (long long)n);
    if (p.vec_c > 1)
        for (int64_t n = 0; n + p.vec_c <= p.N; n += p.vec_c)
            for (int i = 1; i < p.vec_c; ++i) EXPECT(p.col_c[n + i] == p.col_c[n] + i, "%s: vec_c at n %lld", what.c_str(), (long long)n);
    // skinny split of N: the (B offset, C offset) pairs of (x, y, o) are those of the columns
    if (p.kernel == ctc::kKernelSkinnyTranspose) {
        EXPECT(p.sk_X * p.sk_Y * p.sk_O == p.N, "%s: skinny split", what.c_str());
        std::multiset<std::pair<int64_t, int64_t>> want, got;
        for (int64_t n = 0; n < p.N; ++n) want.insert({p.col_b[n], p.col_c[n]});
        for (int64_t o = 0; o < p.sk_O; ++o)
            for (int64_t y = 0; y < p.sk_Y; ++y)
                for (int64_t x = 0; x < p.sk_X; ++x)
                    got.insert({p.sk_bx.at((size_t)x) + p.sk_by.at((size_t)y) + p.sk_bo.at((size_t)o),
                                p.sk_cx.at((size_t)x) + p.sk_cy.at((size_t)y) + p.sk_co.at((size_t)o)});
        EXPECT(want == got, "%s: skinny tables", what.c_str());
    }
}

static void run_plan(const Plan &p, int64_t na, int64_t nb, int64_t nc, const std::string &what)
{
    std::vector<int64_t> A((size_t)na), B((size_t)nb), C1((size_t)nc, 0), C2((size_t)nc, 0);
    for (int64_t i = 0; i < na; ++i) A[(size_t)i] = val(i, 3);
    for (int64_t i = 0; i < nb; ++i) B[(size_t)i] = val(i, 8);
    if (getenv("PLANNER_SANITIZE_SELFTEST") && p.K > 1) {  // a wrong table entry must be noticed
        Plan q = p;
        std::swap(q.k_a.front(), q.k_a.back());
        std::swap(q.k_b.front(), q.k_b.back());
        q.k_b[0] = q.k_b[1];
        by_tables(q, A, B, C1);
    } else {
        by_tables(p, A, B, C1);
    }
    direct(p, A, B, C2);
    EXPECT(C1 == C2, "%s: tables and direct evaluation differ", what.c_str());
    check_flags(p, what);
}

static void one_contraction(int n)
{
    const int n_fa = pick({0, 1, 1, 2, 3}), n_fb = pick({0, 1, 1, 2, 3}), n_k = pick({0, 1, 1, 2, 3}), n_bt = pick({0, 0, 1, 2});
    std::string names = "abcdefghijkl";
    std::shuffle(names.begin(), names.end(), rng);
    int at = 0;
    auto take = [&](int c) {
        std::string s = names.substr((size_t)at, (size_t)c);
        at += c;
        return s;
    };
    std::string fa = take(n_fa), fb = take(n_fb), ks = take(n_k), bs = take(n_bt);
    std::string a = fa + ks + bs, b = fb + ks + bs, c = fa + fb + bs;
    if (a.empty() || b.empty()) return;
    std::shuffle(a.begin(), a.end(), rng);
    std::shuffle(b.begin(), b.end(), rng);
    std::shuffle(c.begin(), c.end(), rng);
    std::map<char, int64_t> ext;
    const int style = (int)(rng() % 6);
    for (char ch : a + b)
        ext[ch] = style == 0 ? pick({1, 2, 3, 5}) : style == 1 ? pick({2, 4, 8}) : style == 2 ? pick({4, 8, 16}) : style == 3 ? pick({3, 7, 9})
                                                                                           : style == 4 ? pick({2, 6, 14, 22}) : pick({1, 4, 32});
    auto numel = [&](const std::string &s) {
        int64_t v = 1;
        for (char ch : s) v *= ext[ch];
        return v;
    };
    int64_t space = 1;
    for (auto &kv : ext) space *= kv.second;
    if (space > 300000 || numel(a) > 100000 || numel(b) > 100000 || numel(c) > 100000) return;
    std::string spec = a + "," + b + "->" + c;
    std::string mode_names;
    std::vector<int64_t> extents;
    for (auto &kv : ext) {
        mode_names.push_back(kv.first);
        extents.push_back(kv.second);
    }
    const int dtype = pick({0, 0, 1, 2});
    Plan p;
    std::string err = ctc::build_plan(spec, (int)mode_names.size(), mode_names.data(), extents.data(), nullptr, nullptr, nullptr, dtype, p);
    const std::string what = "#" + std::to_string(n) + " " + spec + " dtype " + std::to_string(dtype);
    EXPECT(err.empty(), "%s: %s", what.c_str(), err.c_str());
    EXPECT(p.size_a == numel(a) && p.size_b == numel(b) && p.size_c == numel(c), "%s: sizes", what.c_str());
    run_plan(p, numel(a), numel(b), numel(c), what);
    ++ran_contractions;
    took_skinny += p.kernel != ctc::kKernelGemm;
    took_vec += p.layout_a != ctc::kLayoutScalar || p.layout_b != ctc::kLayoutScalar;
    (void)ctc::describe(p);
    // a slab of a random mode: the shard's tables, applied behind the returned offsets, must give the slab
    const char mode = mode_names[rng() % mode_names.size()];
    const int nranks = pick({2, 3, 4});
    if (ext[mode] % nranks == 0) {
        const int rank = (int)(rng() % (unsigned)nranks);
        Plan s;
        int64_t oa = 0, ob = 0, oc = 0;
        err = ctc::shard_plan(p, mode, rank, nranks, s, oa, ob, oc);
        EXPECT(err.empty(), "%s: shard %c: %s", what.c_str(), mode, err.c_str());
        // run the shard on the full operands shifted by the offsets, compare with the direct evaluation
        // restricted to the slab of `mode`
        std::vector<int64_t> A((size_t)numel(a)), B((size_t)numel(b)), C1((size_t)numel(c), 0), C2((size_t)numel(c), 0);
        for (size_t i = 0; i < A.size(); ++i) A[i] = val((int64_t)i, 3);
        for (size_t i = 0; i < B.size(); ++i) B[i] = val((int64_t)i, 8);
        const std::vector<int64_t> &ga = s.swapped ? B : A, &gb = s.swapped ? A : B;
        const int64_t sa = s.swapped ? ob : oa, sb = s.swapped ? oa : ob;
        for (int64_t bt = 0; bt < s.batch; ++bt)
            for (int64_t m = 0; m < s.M; ++m)
                for (int64_t nn = 0; nn < s.N; ++nn) {
                    int64_t acc = 0;
                    for (int64_t k = 0; k < s.K; ++k)
                        acc += ga.at((size_t)(sa + s.bat_a[bt] + s.row_a[m] + s.k_a[k])) * gb.at((size_t)(sb + s.bat_b[bt] + s.col_b[nn] + s.k_b[k]));
                    C1.at((size_t)(oc + s.bat_c[bt] + s.row_c[m] + s.col_c[nn])) += acc;
                }
        // direct, slab only
        size_t mi = 0;
        for (; mi < p.modes.size(); ++mi)
            if (p.modes[mi].name == mode) break;
        const int64_t per = ext[mode] / nranks, lo = rank * per, hi = lo + per;
        const size_t nm = p.modes.size();
        std::vector<int64_t> idx(nm, 0);
        for (;;) {
            if (idx[mi] >= lo && idx[mi] < hi) {
                int64_t xa = 0, xb = 0, xc = 0;
                for (size_t i = 0; i < nm; ++i) {
                    xa += idx[i] * p.modes[i].sa;
                    xb += idx[i] * p.modes[i].sb;
                    xc += idx[i] * p.modes[i].sc;
                }
                C2.at((size_t)xc) += A.at((size_t)xa) * B.at((size_t)xb);
            }
            size_t d = 0;
            for (; d < nm; ++d) {
                if (++idx[d] < p.modes[d].extent) break;
                idx[d] = 0;
            }
            if (d == nm) break;
        }
        EXPECT(C1 == C2, "%s: shard of %c, rank %d of %d", what.c_str(), mode, rank, nranks);
        check_flags(s, what + " shard");
        ++ran_shards;
    }
}

static void one_permutation(int n)
{
    const int rank = 1 + (int)(rng() % 6);
    std::vector<int64_t> ext((size_t)rank);
    for (auto &e : ext) e = pick({1, 2, 3, 4, 5, 8, 12, 16, 17, 33});
    int64_t total = 1;
    for (auto e : ext) total *= e;
    if (total > 200000) return;
    std::vector<int> perm((size_t)rank);
    for (int i = 0; i < rank; ++i) perm[(size_t)i] = i;
    std::shuffle(perm.begin(), perm.end(), rng);
    const int es = pick({2, 4, 8});
    // every other case: the last few output modes form rows that lie a padded pitch apart
    int row_modes = 0;
    int64_t row_len = 1, pitch = 0;
    if (rng() % 2) {
        row_modes = 1 + (int)(rng() % (unsigned)rank);
        for (int p = rank - row_modes; p < rank; ++p) row_len *= ext[(size_t)perm[(size_t)p]];
        const int64_t V = 16 / es;
        pitch = (row_len + V - 1) / V * V + V * (int64_t)(rng() % 3);
        if (row_len < 2) row_modes = 0;
    }
    ctc::PermutePlan pp;
    std::string err = ctc::plan_permute(rank, ext.data(), perm.data(), es, pp, row_modes, row_modes ? pitch : 0);
    const std::string what = "permutation #" + std::to_string(n) + (row_modes ? " (padded rows)" : "");
    EXPECT(err.empty(), "%s: %s", what.c_str(), err.c_str());
    EXPECT(pp.total == total && pp.X * pp.Y * pp.O == total, "%s: sizes", what.c_str());
    const int64_t rows = total / row_len, out_span = row_modes ? (rows - 1) * pitch + row_len : total;
    // out[out offset] = in[in offset] for every (x, y, o) must be the permutation
    std::vector<int64_t> out((size_t)out_span, -1);
    for (int64_t o = 0; o < pp.O; ++o)
        for (int64_t y = 0; y < pp.Y; ++y)
            for (int64_t x = 0; x < pp.X; ++x) {
                const int64_t i = pp.in_x.at((size_t)x) + pp.in_y.at((size_t)y) + pp.in_o.at((size_t)o);
                const int64_t j = pp.out_x.at((size_t)x) + pp.out_y.at((size_t)y) + pp.out_o.at((size_t)o);
                EXPECT(i >= 0 && i < total && j >= 0 && j < out_span, "%s: offset outside the tensor", what.c_str());
                EXPECT(out[(size_t)j] == -1, "%s: an output element written twice", what.c_str());
                out[(size_t)j] = i;
            }
    // direct: out index (p0 .. p_{r-1}) with extents ext[perm[p]] holds in index whose mode perm[p] is p's coordinate
    std::vector<int64_t> in_stride((size_t)rank, 1), oext((size_t)rank);
    for (int d = rank - 2; d >= 0; --d) in_stride[(size_t)d] = in_stride[(size_t)d + 1] * ext[(size_t)d + 1];
    for (int d = 0; d < rank; ++d) oext[(size_t)d] = ext[(size_t)perm[(size_t)d]];
    std::vector<int64_t> idx((size_t)rank, 0);
    for (int64_t j = 0; j < total; ++j) {
        int64_t i = 0;
        for (int d = 0; d < rank; ++d) i += idx[(size_t)d] * in_stride[(size_t)perm[(size_t)d]];
        const int64_t at = row_modes ? (j / row_len) * pitch + j % row_len : j;  // packed output index j -> padded offset
        EXPECT(out[(size_t)at] == i, "%s: element %lld comes from %lld, expected %lld", what.c_str(), (long long)j, (long long)out[(size_t)at], (long long)i);
        for (int d = rank - 1; d >= 0; --d) {
            if (++idx[(size_t)d] < oext[(size_t)d]) break;
            idx[(size_t)d] = 0;
        }
    }
    // contiguous-run claims: the first in_run entries of the x table step by one in the input, the first out_run
    // entries of the y (or, with a common fastest mode, x) table step by one in the output
    const std::vector<int64_t> &oy = pp.same_fast ? pp.out_x : pp.out_y;
    for (int64_t k = 1; k < pp.in_run && k < (int64_t)pp.in_x.size(); ++k) EXPECT(pp.in_x[(size_t)k] == pp.in_x[0] + k, "%s: in_run", what.c_str());
    for (int64_t k = 1; k < pp.out_run && k < (int64_t)oy.size(); ++k) EXPECT(oy[(size_t)k] == oy[0] + k, "%s: out_run", what.c_str());
    ++ran_permutations;
    ran_padded += row_modes > 0;
}

int main(int argc, char **argv)
{
    const int cases = argc > 1 ? atoi(argv[1]) : 2000;
    rng.seed(argc > 2 ? (uint64_t)atoll(argv[2]) : 0xC7C5A11ull);
    for (int n = 0; n < cases; ++n) {
        one_contraction(n);
        one_permutation(n);
    }
    // malformed input must come back as an error message, not as a crash
    const char *bad[] = {"", "ab", "ab,->", "ab,bc->ad", "aa,ab->b", "ab,bc->acc", "ab,bc", "a b,bc->ac", "ab,bc->ac->x", ",->", "ab,cd->abcd e"};
    for (const char *s : bad) {
        Plan p;
        const int64_t e[4] = {2, 3, 4, 5};
        (void)ctc::build_plan(s, 4, "abcd", e, nullptr, nullptr, nullptr, 0, p);
    }
    printf("%d contractions (%d with a vector layout, %d on the skinny kernels), %d shards, %d permutations (%d with padded rows): %d failures\n", ran_contractions, took_vec,
           took_skinny, ran_shards, ran_permutations, ran_padded, failures);
    return failures ? 1 : 0;
}
