// The host side of the C ABI (api.cpp, sharded.cpp, planner.cpp, permute_plan.cpp) under AddressSanitizer +
// UndefinedBehaviorSanitizer, on a machine WITHOUT a GPU: every entry point that needs no device is driven with
// seeded random specs, short and exact output buffers, invalid arguments; entry points that need a device must
// come back with an error code.  Built and run by tools/api_sanitize.sh.
#include "ctc_fastr.h"
#include "ctc_tc.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <random>
#include <string>
#include <vector>

static std::mt19937_64 rng;
static int pick(std::initializer_list<int> v) { return *(v.begin() + rng() % v.size()); }
static int failures = 0;
#define EXPECT(cond, ...)                                  \
    do {                                                   \
        if (!(cond)) {                                     \
            ++failures;                                    \
            fprintf(stderr, "FAIL line %d %s: ", __LINE__, #cond); \
            fprintf(stderr, __VA_ARGS__);                  \
            fprintf(stderr, "\n");                         \
        }                                                  \
    } while (0)

static void queries(const ctc_tc_plan *p, const std::string &what)
{
    for (int q = -1; q <= 30; ++q) {
        const int64_t n = ctc_tc_plan_get(p, q, nullptr, 0);
        if (q < 0 || q > CTC_TC_Q_SPANS) {
            EXPECT(n < 0, "%s: query %d must fail", what.c_str(), q);
            continue;
        }
        if (q == CTC_TC_Q_SCHEDULE) {
            EXPECT(n < 0, "%s: the schedule needs a device", what.c_str());
            continue;
        }
        EXPECT(n >= 0, "%s: query %d: %s", what.c_str(), q, ctc_tc_last_error());
        if (n <= 0) continue;
        // exact buffer, short buffer, one-element buffer: never more than cap values are written (the heap
        // blocks have exactly cap elements, so a value too many is a sanitizer report)
        for (int64_t cap : {n, n / 2, (int64_t)1}) {
            if (cap <= 0) continue;
            std::vector<int64_t> buf((size_t)cap, -77);
            const int64_t got = ctc_tc_plan_get(p, q, buf.data(), cap);
            EXPECT(got == n, "%s: query %d reports %lld then %lld values", what.c_str(), q, (long long)n, (long long)got);
        }
    }
    char small[8];
    const int len = ctc_tc_plan_describe(p, small, sizeof small);
    EXPECT(len > 0 && strlen(small) < sizeof small, "%s: describe into 8 bytes", what.c_str());
    std::vector<char> big((size_t)len + 1);
    const int len2 = ctc_tc_plan_describe(p, big.data(), big.size());
    EXPECT(len2 == len && (int)strlen(big.data()) == len, "%s: describe %d vs %d", what.c_str(), len, len2);
    (void)ctc_tc_plan_describe(p, nullptr, 0);
    EXPECT(ctc_tc_ttgt_workspace_bytes(p) >= 0, "%s: ttgt workspace", what.c_str());
}

static void one_case(int n)
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
    const int style = (int)(rng() % 5);
    for (char ch : a + b)
        ext[ch] = style == 0 ? pick({1, 2, 3, 5}) : style == 1 ? pick({2, 4, 8, 64}) : style == 2 ? pick({16, 24, 40, 96, 49, 1025}) : style == 3 ? pick({3, 7, 9, 33})
                                                                                                                                        : pick({2, 6, 14, 22});
    int64_t space = 1;
    for (auto &kv : ext) space = space > ((int64_t)1 << 40) ? space : space * kv.second;
    auto numel = [&](const std::string &m) {
        int64_t v = 1;
        for (char ch : m) v = v > ((int64_t)1 << 40) ? v : v * ext[ch];
        return v;
    };
    if (space > (int64_t)1 << 34 || numel(a) > 1 << 22 || numel(b) > 1 << 22 || numel(c) > 1 << 22) return;  // the offset tables are per row / column
    const std::string spec = a + "," + b + "->" + c;
    std::string mode_names;
    std::vector<int64_t> extents;
    for (auto &kv : ext) {
        mode_names.push_back(kv.first);
        extents.push_back(kv.second);
    }
    // strides: packed (NULL), or packed with the fastest mode padded by a few elements
    auto padded = [&](const std::string &modes, int pad) {
        std::vector<int64_t> s(modes.size(), 1);
        int64_t run = 1;
        for (int i = (int)modes.size() - 1; i >= 0; --i) {
            s[(size_t)i] = run;
            run *= ext[modes[(size_t)i]] + (i == (int)modes.size() - 1 ? pad : 0);
        }
        return s;
    };
    const bool strided = rng() % 3 == 0;
    std::vector<int64_t> sa = padded(a, pick({0, 1, 2})), sb = padded(b, pick({0, 1, 4})), sc = padded(c, pick({0, 1, 3}));
    const int dtype = pick({0, 0, 1, 2});
    const std::string what = "#" + std::to_string(n) + " " + spec;
    ctc_tc_plan *p = nullptr;
    int rc = ctc_tc_plan_create(spec.c_str(), (int)mode_names.size(), mode_names.data(), extents.data(), strided ? sa.data() : nullptr,
                                strided ? sb.data() : nullptr, strided && !c.empty() ? sc.data() : nullptr, dtype, &p);
    EXPECT(rc == CTC_TC_OK && p != nullptr, "%s: %d %s", what.c_str(), rc, ctc_tc_last_error());
    if (rc != CTC_TC_OK) return;
    queries(p, what);
    {   // the plan whose GEMM an execution launches: the plan itself or a sub-plan it owns
        ctc_tc_plan *g = ctc_tc_plan_gemm_plan(p);
        EXPECT(g != nullptr, "%s: gemm plan: %s", what.c_str(), ctc_tc_last_error());
        if (g && g != p) queries(g, what + " gemm plan");
        EXPECT(ctc_tc_plan_gemm_plan(p) == g, "%s: gemm plan handed out twice", what.c_str());
    }
    // every mode, a few rank counts: evenly divisible or an error, never a crash
    for (char mode : mode_names) {
        const int nr = pick({1, 2, 3, 4, 8});
        const int rank = (int)(rng() % (unsigned)nr);
        ctc_tc_plan *s = nullptr;
        int64_t oa = -1, ob = -1, oc = -1;
        rc = ctc_tc_plan_shard(p, mode, rank, nr, &s, &oa, &ob, &oc);
        if (ext[mode] % nr == 0) {
            EXPECT(rc == CTC_TC_OK && s != nullptr && oa >= 0 && ob >= 0 && oc >= 0, "%s: shard %c %d/%d: %s", what.c_str(), mode, rank, nr, ctc_tc_last_error());
        } else {
            EXPECT(rc != CTC_TC_OK && s == nullptr, "%s: shard %c of %d ranks must fail", what.c_str(), mode, nr);
        }
        if (s) {
            queries(s, what + " shard");
            EXPECT(ctc_tc_plan_destroy(s) == CTC_TC_OK, "destroy shard");
        }
    }
    ctc_tc_plan *s = nullptr;
    int64_t o1, o2, o3;
    EXPECT(ctc_tc_plan_shard(p, '?', 0, 2, &s, &o1, &o2, &o3) != CTC_TC_OK, "%s: unknown mode", what.c_str());
    EXPECT(ctc_tc_plan_shard(p, mode_names[0], 2, 2, &s, &o1, &o2, &o3) != CTC_TC_OK, "%s: rank out of range", what.c_str());
    EXPECT(ctc_tc_plan_shard(p, mode_names[0], 0, 0, &s, &o1, &o2, &o3) != CTC_TC_OK, "%s: zero ranks", what.c_str());
    // no device here: these must return an error code
    double dummy[4] = {0, 0, 0, 0};
    EXPECT(ctc_tc_execute(p, 1.0, dummy, dummy, 0.0, dummy, nullptr) != CTC_TC_OK, "%s: execute without a device", what.c_str());
    EXPECT(ctc_tc_execute_noalloc(p, 1.0, dummy, dummy, 0.0, dummy, nullptr) != CTC_TC_OK, "%s: noalloc without a device", what.c_str());
    EXPECT(ctc_tc_plan_prepare(p) != CTC_TC_OK, "%s: prepare without a device", what.c_str());
    EXPECT(ctc_tc_execute(p, 1.0, nullptr, dummy, 0.0, dummy, nullptr) != CTC_TC_OK, "%s: null operand", what.c_str());
    // contracted-mode sharding objects: host only
    if (!strided)
        for (char mode : ks) {
            for (int reduce : {CTC_TC_REDUCE_SCATTER, CTC_TC_ALL_REDUCE}) {
                const int nr = pick({1, 2, 4}), chunks = pick({1, 2, 4, 7});
                const int rank = (int)(rng() % (unsigned)nr);
                ctc_tc_sharded *sh = nullptr;
                rc = ctc_tc_sharded_create(spec.c_str(), (int)mode_names.size(), mode_names.data(), extents.data(), dtype, mode, rank, nr, chunks, reduce, &sh);
                if (ext[mode] % nr != 0) {
                    EXPECT(rc != CTC_TC_OK && sh == nullptr, "%s: sharded on %c over %d ranks must fail", what.c_str(), mode, nr);
                    continue;
                }
                // documented refusals: a scalar result; a reduce-scatter whose |C| the ranks do not divide
                if (rc != CTC_TC_OK && (c.empty() || (reduce == CTC_TC_REDUCE_SCATTER && numel(c) % nr != 0))) {
                    EXPECT(sh == nullptr && strlen(ctc_tc_last_error()) > 0, "%s: refusal without a message", what.c_str());
                    continue;
                }
                EXPECT(rc == CTC_TC_OK && sh != nullptr, "%s: sharded create %c: %s", what.c_str(), mode, ctc_tc_last_error());
                if (!sh) continue;
                const int64_t nch = ctc_tc_sharded_get(sh, CTC_TC_SQ_CHUNKS, 0);
                EXPECT(nch >= 1 && nch <= chunks, "%s: %lld chunks of at most %d", what.c_str(), (long long)nch, chunks);
                int64_t covered = 0;
                for (int64_t ch = -1; ch <= nch; ++ch)
                    for (int q : {0, 1, 2, 3, 4, 15, 16, 17, 18, 19, 20, 21}) {
                        const int64_t v = ctc_tc_sharded_get(sh, q, ch);
                        if (q == CTC_TC_SQ_CHUNK_ELEMS && ch >= 0 && ch < nch) covered += v;
                        if (q >= 16 && q <= 20 && (ch < 0 || ch >= nch)) EXPECT(v < 0, "%s: chunk %lld of %lld, query %d", what.c_str(), (long long)ch, (long long)nch, q);
                    }
                EXPECT(covered == ctc_tc_sharded_get(sh, CTC_TC_SQ_C_ELEMS, 0), "%s: the chunks cover C", what.c_str());
                for (int64_t ch = -1; ch <= nch; ++ch) {
                    ctc_tc_plan *cp = ctc_tc_sharded_chunk_plan(sh, (int)ch);
                    EXPECT((cp != nullptr) == (ch >= 0 && ch < nch), "%s: chunk plan %lld", what.c_str(), (long long)ch);
                    if (cp) queries(cp, what + " chunk");
                }
                EXPECT(ctc_tc_sharded_prepare(sh) != CTC_TC_OK, "%s: sharded prepare without a device", what.c_str());
                EXPECT(ctc_tc_sharded_destroy(sh) == CTC_TC_OK, "destroy sharded");
            }
        }
    EXPECT(ctc_tc_plan_destroy(p) == CTC_TC_OK, "destroy");
}

static void bad_arguments()
{
    ctc_tc_plan *p = nullptr;
    const int64_t e4[4] = {2, 3, 4, 5};
    const char *bad[] = {"", "ab", "ab,->", "ab,bc->ad", "aa,ab->b", "ab,bc->acc", "ab,bc", "a b,bc->ac", "ab,bc->ac->x", ",->", "ab,cd->abcd e", "AB,BC->AC", "ab,bc->ac,"};
    for (const char *s : bad) {
        p = nullptr;
        const int rc = ctc_tc_plan_create(s, 4, "abcd", e4, nullptr, nullptr, nullptr, 0, &p);
        if (rc == CTC_TC_OK) ctc_tc_plan_destroy(p);  // a spec this list calls bad may be legal; it must not crash
        else EXPECT(p == nullptr && strlen(ctc_tc_last_error()) > 0, "'%s'", s);
    }
    EXPECT(ctc_tc_plan_create(nullptr, 4, "abcd", e4, nullptr, nullptr, nullptr, 0, &p) != CTC_TC_OK, "null spec");
    EXPECT(ctc_tc_plan_create("ab,bc->ac", 4, "abcd", e4, nullptr, nullptr, nullptr, 0, nullptr) != CTC_TC_OK, "null result");
    EXPECT(ctc_tc_plan_create("ab,bc->ac", 3, nullptr, e4, nullptr, nullptr, nullptr, 0, &p) != CTC_TC_OK, "null names");
    EXPECT(ctc_tc_plan_create("ab,bc->ac", 3, "abc", nullptr, nullptr, nullptr, nullptr, 0, &p) != CTC_TC_OK, "null extents");
    EXPECT(ctc_tc_plan_create("ab,bc->ac", 2, "ab", e4, nullptr, nullptr, nullptr, 0, &p) != CTC_TC_OK, "a letter without an extent");
    EXPECT(ctc_tc_plan_create("ab,bc->ac", 3, "abc", e4, nullptr, nullptr, nullptr, 7, &p) != CTC_TC_OK, "unknown dtype");
    EXPECT(ctc_tc_plan_create("ab,bc->ac", 0, "", e4, nullptr, nullptr, nullptr, 0, &p) != CTC_TC_OK, "no extents");
    EXPECT(ctc_tc_plan_create("ab,bc->ac", -3, "abc", e4, nullptr, nullptr, nullptr, 0, &p) != CTC_TC_OK, "negative count");
    const int64_t neg[3] = {2, -3, 4}, zero[3] = {2, 0, 4}, huge[3] = {(int64_t)1 << 40, (int64_t)1 << 40, (int64_t)1 << 40};
    EXPECT(ctc_tc_plan_create("ab,bc->ac", 3, "abc", neg, nullptr, nullptr, nullptr, 0, &p) != CTC_TC_OK, "negative extent");
    if (ctc_tc_plan_create("ab,bc->ac", 3, "abc", zero, nullptr, nullptr, nullptr, 0, &p) == CTC_TC_OK) ctc_tc_plan_destroy(p);
    if (ctc_tc_plan_create("ab,bc->ac", 3, "abc", huge, nullptr, nullptr, nullptr, 0, &p) == CTC_TC_OK) ctc_tc_plan_destroy(p);
    // more modes than the library takes
    {
        const std::string many = "abcdefghijklmnopq";  // 17 letters
        std::vector<int64_t> e(many.size(), 2);
        const std::string spec = many + "," + many + "->" + many;
        if (ctc_tc_plan_create(spec.c_str(), (int)many.size(), many.c_str(), e.data(), nullptr, nullptr, nullptr, 0, &p) == CTC_TC_OK) ctc_tc_plan_destroy(p);
    }
    EXPECT(ctc_tc_plan_get(nullptr, 0, nullptr, 0) < 0, "get on a null plan");
    EXPECT(ctc_tc_plan_gemm_plan(nullptr) == nullptr, "gemm plan of a null plan");
    EXPECT(ctc_tc_plan_destroy(nullptr) == CTC_TC_OK || true, "destroy null");
    (void)ctc_tc_plan_describe(nullptr, nullptr, 0);
    EXPECT(ctc_tc_ttgt_workspace_bytes(nullptr) <= 0, "ttgt workspace of a null plan");
    EXPECT(ctc_tc_sharded_get(nullptr, 0, 0) < 0, "sharded get on null");
    EXPECT(ctc_tc_sharded_chunk_plan(nullptr, 0) == nullptr, "chunk plan of null");
    (void)ctc_tc_sharded_destroy(nullptr);
    // options
    for (const char *name : {"stream_mode", "stream_max_k", "stream_f64_dma", "dma_gemm", "stream_k", "stream_k_group", "launches_classic",
                             "launches_dma_gemm", "launches_stream", "launches_stream_f64_dma", "launches_bf16_ring"}) {
        const int64_t v = ctc_tc_get_option(name);
        EXPECT(v >= 0, "option %s", name);
        if (strncmp(name, "launches", 8) != 0) {
            EXPECT(ctc_tc_set_option(name, v) == CTC_TC_OK, "set %s", name);
            (void)ctc_tc_set_option(name, -5);
            (void)ctc_tc_set_option(name, (int64_t)1 << 40);
            EXPECT(ctc_tc_set_option(name, v) == CTC_TC_OK, "restore %s", name);
        }
    }
    EXPECT(ctc_tc_get_option("no_such_option") < 0 && ctc_tc_set_option("no_such_option", 1) != CTC_TC_OK, "unknown option");
    EXPECT(ctc_tc_get_option(nullptr) < 0 && ctc_tc_set_option(nullptr, 1) != CTC_TC_OK, "null option");
    // device entry points without a device
    double d[4] = {0, 0, 0, 0};
    const int64_t pe[3] = {2, 3, 4};
    const int perm[3] = {2, 0, 1}, bad_perm[3] = {0, 0, 1};
    EXPECT(ctc_tc_permute(d, d, 3, pe, perm, 8, nullptr) != CTC_TC_OK, "permute without a device");
    EXPECT(ctc_tc_permute(d, d, 3, pe, bad_perm, 8, nullptr) != CTC_TC_OK, "permute with a repeated mode");
    EXPECT(ctc_tc_permute(d, d, 3, pe, perm, 3, nullptr) != CTC_TC_OK, "permute with element size 3");
    EXPECT(ctc_tc_permute(d, d, 0, pe, perm, 8, nullptr) != CTC_TC_OK || true, "permute of rank 0");
    EXPECT(ctc_tc_permute(d, d, 40, pe, perm, 8, nullptr) != CTC_TC_OK, "permute of rank 40");
    EXPECT(ctc_tc_fill(d, 4, 1, 0, 0, nullptr) != CTC_TC_OK, "fill without a device");
    EXPECT(ctc_tc_grouped_contract(nullptr, 3, 0, 1.0, d, d, 0.0, d, nullptr) != CTC_TC_OK, "grouped without descriptors");
    EXPECT(ctc_tc_device_count() == 0, "device count");
    EXPECT(ctc_tc_version() > 0 && ctc_tc_noalloc_scratch_bytes() > 0, "version / scratch size");
    // the scoring / alignment entry points: argument checks come before anything touches a device
    {
        const uint8_t *u8 = reinterpret_cast<const uint8_t *>(d);
        const int32_t *i32 = reinterpret_cast<const int32_t *>(d);
        int32_t *o32 = reinterpret_cast<int32_t *>(d);
        const int32_t ncons[2] = {2, 1}, reads[2] = {3, 4}, lens[3] = {200, 300, 250}, neg[2] = {2, -1}, neg_len[3] = {200, -5, 250};
        EXPECT(ctc_fastr_best_offsets(u8, u8, u8, 100, i32, i32, 0, ncons, reads, lens, o32, o32, nullptr) == CTC_TC_OK, "no regions");
        EXPECT(ctc_fastr_best_offsets(u8, u8, u8, 100, i32, i32, -1, ncons, reads, lens, o32, o32, nullptr) == CTC_TC_EINVAL, "negative regions");
        EXPECT(ctc_fastr_best_offsets(u8, u8, u8, 0, i32, i32, 2, ncons, reads, lens, o32, o32, nullptr) == CTC_TC_EINVAL, "zero stride");
        EXPECT(ctc_fastr_best_offsets(nullptr, u8, u8, 100, i32, i32, 2, ncons, reads, lens, o32, o32, nullptr) == CTC_TC_EINVAL, "null ref");
        EXPECT(ctc_fastr_best_offsets(u8, u8, u8, 100, i32, i32, 2, neg, reads, lens, o32, o32, nullptr) == CTC_TC_EINVAL, "negative count");
        EXPECT(ctc_fastr_best_offsets(u8, u8, u8, 100, i32, i32, 2, ncons, neg, lens, o32, o32, nullptr) == CTC_TC_EINVAL, "negative reads");
        EXPECT(ctc_fastr_best_offsets(u8, u8, u8, 100, i32, i32, 2, ncons, reads, neg_len, o32, o32, nullptr) == CTC_TC_EINVAL, "negative length");
        EXPECT(ctc_fastr_best_offsets(u8, u8, u8, 100, i32, i32, 2, ncons, reads, lens, o32, o32, nullptr) == CTC_TC_EDEVICE, "scoring without a device");
        EXPECT(ctc_fastr_sw_workspace_bytes(0, 100, 100) == 0 && ctc_fastr_sw_workspace_bytes(10, 100, 100) > 0, "alignment workspace size");
        uint8_t *o8 = reinterpret_cast<uint8_t *>(d);
        const int64_t *i64 = reinterpret_cast<const int64_t *>(d);
        EXPECT(ctc_fastr_sw_align(u8, i64, i32, u8, 100, i32, 0, 1, -1, -2, -1, 100, 100, o32, o32, o8, o32, 8, d, 1 << 20, nullptr) == CTC_TC_OK, "no pairs");
        EXPECT(ctc_fastr_sw_align(u8, i64, i32, u8, 100, i32, 4, 1, -1, -2, -1, 40000, 100, o32, o32, o8, o32, 8, d, 1 << 20, nullptr) == CTC_TC_EUNSUPPORTED, "window too long");
        EXPECT(ctc_fastr_sw_align(u8, i64, i32, u8, 100, i32, 4, 1, -1, -2, -1, 100, 200, o32, o32, o8, o32, 8, d, 1 << 20, nullptr) == CTC_TC_EINVAL, "read longer than its slot");
        EXPECT(ctc_fastr_sw_align(u8, i64, i32, u8, 100, i32, 4, 1, -1, -2, -1, 100, 100, o32, o32, o8, o32, 8, d, 16, nullptr) == CTC_TC_EINVAL, "workspace too small");
        EXPECT(ctc_fastr_sw_align(u8, i64, i32, u8, 100, i32, 4, 1, -1, -2, -1, 100, 100, o32, o32, o8, o32, 0, d, 1 << 20, nullptr) == CTC_TC_EINVAL, "no room for operations");
        EXPECT(ctc_fastr_sw_align(u8, i64, i32, u8, 100, i32, 4, 1, -1, -2, -1, 100, 100, o32, o32, o8, o32, 8, nullptr, 1 << 20, nullptr) == CTC_TC_EINVAL, "null workspace");
    }
    unsigned char id[CTC_TC_UNIQUE_ID_BYTES];
    (void)ctc_tc_comm_unique_id(id);
    (void)ctc_tc_comm_destroy(nullptr);
}

int main(int argc, char **argv)
{
    const int cases = argc > 1 ? atoi(argv[1]) : 1000;
    rng.seed(argc > 2 ? (uint64_t)atoll(argv[2]) : 0xC7CA91ull);
    bad_arguments();
    for (int n = 0; n < cases; ++n) one_case(n);
    printf("%d cases: %d failures\n", cases, failures);
    return failures ? 1 : 0;
}
