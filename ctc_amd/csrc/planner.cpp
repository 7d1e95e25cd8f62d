// Contraction planner (host only).  See planner.hpp.
#include "planner.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <sstream>

namespace ctc {

namespace {

int find_mode(const Plan &p, char ch)
{
    for (size_t i = 0; i < p.modes.size(); ++i)
        if (p.modes[i].name == ch) return (int)i;
    return -1;
}

bool is_letter(char ch) { return (ch >= 'a' && ch <= 'z') || (ch >= 'A' && ch <= 'Z'); }

std::string parse_operand(const std::string &text, Plan &p, std::vector<int> &out, bool may_add)
{
    out.clear();
    for (char ch : text) {
        if (ch == ' ') continue;
        if (!is_letter(ch)) return std::string("malformed spec: unexpected character '") + ch + "'";
        int id = find_mode(p, ch);
        if (id < 0) {
            if (!may_add) return std::string("malformed spec: output mode '") + ch + "' is in neither input";
            if ((int)p.modes.size() >= kMaxModes) return "too many distinct modes";
            Mode m;
            m.name = ch;
            p.modes.push_back(m);
            id = (int)p.modes.size() - 1;
        }
        if (std::find(out.begin(), out.end(), id) != out.end())
            return std::string("malformed spec: mode '") + ch + "' repeated inside one operand";
        out.push_back(id);
    }
    return "";
}

int position_of(const std::vector<int> &v, int id)
{
    auto it = std::find(v.begin(), v.end(), id);
    return it == v.end() ? -1 : (int)(it - v.begin());
}

void packed_strides(const Plan &p, const std::vector<int> &modes, std::vector<int64_t> &st)
{
    st.assign(modes.size(), 0);
    int64_t s = 1;
    for (int i = (int)modes.size() - 1; i >= 0; --i) {
        st[i] = s;
        s *= p.modes[modes[i]].extent;
    }
}

// stable order of `ids` by ascending stride (fastest first), extent-1 modes dropped
std::vector<int> fastest_first(const Plan &p, const std::vector<int> &ids,
                               const std::vector<int64_t> &stride_of_mode)
{
    std::vector<int> v;
    for (int id : ids)
        if (p.modes[id].extent > 1) v.push_back(id);
    std::stable_sort(v.begin(), v.end(),
                     [&](int x, int y) { return stride_of_mode[x] < stride_of_mode[y]; });
    return v;
}

int unit_stride_mode(const Plan &p, const std::vector<int> &operand_modes,
                     const std::vector<int64_t> &stride_of_mode)
{
    for (int id : operand_modes)
        if (p.modes[id].extent > 1 && stride_of_mode[id] == 1) return id;
    return -1;
}

void build_table(const Plan &p, const std::vector<int> &order_fastest_first,
                 const std::vector<int64_t> &stride_of_mode, int64_t count, std::vector<int64_t> &tab)
{
    tab.assign((size_t)count, 0);
    std::vector<int64_t> idx(order_fastest_first.size(), 0);
    int64_t off = 0;
    for (int64_t i = 0; i < count; ++i) {
        tab[(size_t)i] = off;
        for (size_t d = 0; d < order_fastest_first.size(); ++d) {
            int id = order_fastest_first[d];
            off += stride_of_mode[id];
            if (++idx[d] < p.modes[id].extent) break;
            off -= idx[d] * stride_of_mode[id];
            idx[d] = 0;
        }
    }
}

int vector_width(const Plan &p, const std::vector<int> &operand_modes,
                 const std::vector<int64_t> &stride_of_mode, int unit_mode, int vmax)
{
    if (vmax <= 1 || unit_mode < 0) return 1;
    if (p.modes[unit_mode].extent % vmax) return 1;
    for (int id : operand_modes) {
        if (id == unit_mode || p.modes[id].extent <= 1) continue;
        if (stride_of_mode[id] % vmax) return 1;
    }
    return vmax;
}

constexpr int64_t kBf16BigTileMinBlocks = 192;

void choose_tile(Plan &p)
{
    Tile t;
    auto blocks = [&](int bm, int bn) {
        return ((p.M + bm - 1) / bm) * ((p.N + bn - 1) / bn) * p.batch;
    };
    // Large tile when it still yields well over one block per CU, or when cutting its tiles along
    // K (the scheduler does that for tiles that would leave workgroup slots idle) yields that many
    // chunks of at least kBigTileMinChunk contracted elements each: on mid-sized products the large
    // tile split along K measures up to 13 % faster than the small tile (1024 x 4096 x 16384 fp64:
    // 64 -> 71 TFLOP/s), while short chunks lose to it (1024^3: 32 vs 35).  The host does not know
    // the device: 512 slots (two workgroups on each of 256 CUs) stand for it here.
    // Otherwise the small tile keeps the CUs busy on modest problems.
    constexpr int64_t kBigTileMinBlocks = 384, kAssumedSlots = 512, kBigTileMinChunk = 512;
    const int64_t big_blocks = blocks(128, 128);
    int64_t big_chunks = big_blocks;
    // (a batched launch's tiles form one list that is cut like a single product's: GemmParams::flat_batch.  With
    // element loads on an operand a batch keeps the small tile, which is cut the same way: 40 x 129 x 131 x 2049
    // runs at 20.6 TFLOP/s on it and at 13.4 on the large one, fp32 4 x 700 x 900 x 1101 at 70.4 / 57.5)
    static const bool batched_too = getenv("CTC_TC_NO_BATCHED_CHUNKS") == nullptr;
    const bool cut = p.batch == 1 || (batched_too && p.layout_a != kLayoutScalar && p.layout_b != kLayoutScalar);
    if (cut && big_blocks > 0 && big_blocks < kBigTileMinBlocks) {
        const int64_t by_k = p.K / kBigTileMinChunk, by_slots = (2 * kAssumedSlots) / big_blocks;
        const int64_t f = by_k < by_slots ? by_k : by_slots;
        // Fewer large tiles than seven eighths of the CUs cannot be cut two or three ways into an even load (192
        // tiles are 1.5 or 2.25 chunks per CU), and left whole they leave CUs idle: 2048 x 1536 x 1100 runs at 43.9
        // TFLOP/s on 192 large tiles and at 56.0 on 768 small ones (fp32 71.7 / 91.0), K = 1536 at 52.7 / 58.2;
        // from 224 tiles on the large tile is ahead (1792 x 2048 x 1280: 60.1 / 55.7).
        // With element loads on an operand (odd unit-stride extents) the large tile runs on the register-staged
        // kernel: fp64 then keeps the small tile below 256 large tiles, where the padded K-major copies begin
        // (1681 x 1763 x 1035: 37.6 against 49.3 TFLOP/s, 1849^3: 45.5 against 52.9); fp32 takes those copies
        // from 1024 x 1024 on, and they need the large tile (1601 x 1603 x 1537, both operands by elements: 70.4
        // through the copies against 49.5 on the small tile), so there the count alone decides as before.
        const bool vectors = p.layout_a != kLayoutScalar && p.layout_b != kLayoutScalar;
        const int64_t few_blocks = vectors ? 224 : (p.dtype == kF64 ? 256 : 0);
        static const bool old_rule = getenv("CTC_TC_OLD_TILE_RULE") != nullptr;  // A/B runs: the count of chunks alone
        bool even = old_rule || big_blocks >= few_blocks || f >= 4;
        // ... unless some cut up to f ways loads the CUs evenly (169 tiles three ways are 1.98 chunks per CU: the
        // headline contraction at extent 40, fp32 94 TFLOP/s on the large tile against 84 - 90 on the small one)
        constexpr int64_t kAssumedCus = kAssumedSlots / 2;
        for (int64_t c = 2; vectors && !even && c <= f; ++c) {
            const int64_t chunks = big_blocks * c, per_cu = (chunks + kAssumedCus - 1) / kAssumedCus;
            if (chunks >= kBigTileMinBlocks && 20 * chunks >= 17 * per_cu * kAssumedCus) even = true;  // 85 % of the busiest CU's load
        }
        if (f > 1 && even) big_chunks = big_blocks * f;
    }
    // What the large tile multiplies beyond C's edges against what the small one does: from 15 % more on, the small
    // tile is ahead at every size measured (144 x 4096 x 4096: 33.5 -> 37.8 TFLOP/s, fp32 52.3 -> 60.5; 8208 x 400 x 1024:
    // 40.5 -> 46.0, fp32 57.0 -> 76.4; at 12 - 14 % more the large tile still leads by up to 5 %).
    auto padded = [&](int64_t edge) { return (double)((p.M + edge - 1) / edge * edge) * (double)((p.N + edge - 1) / edge * edge); };
    static const bool mind_padding = getenv("CTC_TC_OLD_TILE_RULE") == nullptr;
    // (under a K below 2048 from 10 % more on: 16 x 858 x 546 x 89 22.8 -> 30.3, 64 x 580 x 404 x 217 34.7 -> 40.9, fp32
    // 32 x 400 x 1120 x 288 74.1 -> 84.9)
    const bool too_much_padding = mind_padding && padded(128) >= (p.K >= 2048 ? 1.15 : 1.10) * padded(64);
    if (p.M >= 96 && p.N >= 96 && !too_much_padding && (big_blocks >= kBigTileMinBlocks || big_chunks >= kBigTileMinBlocks)) {
        t.bm = 128; t.bn = 128;
    } else {
        t.bm = 64; t.bn = 64;
    }
    // diagnostic knob for A/B runs
    if (const char *force = getenv("CTC_TC_FORCE_TILE")) {
        if (atoi(force) == 64) { t.bm = 64; t.bn = 64; }
        if (atoi(force) == 128) { t.bm = 128; t.bn = 128; }
    }
    t.bk = 16;
    t.threads = 256;
    if (p.dtype == kBF16) {
        // Square tiles, K-tile 64.  The 256 tile moves half the operand bytes per flop from L2
        // to LDS; it needs 16-byte-vector layouts on both operands and enough tiles for the
        // 256 CUs (one workgroup per CU).
        t.bm = 128; t.bn = 128; t.bk = 64;
        const bool vectors = p.layout_a != kLayoutScalar && p.layout_b != kLayoutScalar;
        bool big = vectors && blocks(256, 256) >= kBf16BigTileMinBlocks;
        if (const char *force = getenv("CTC_TC_FORCE_TILE")) {
            if (atoi(force) == 128) big = false;
            if (atoi(force) == 256) big = vectors;
        }
        if (big) { t.bm = 256; t.bn = 256; t.threads = 512; }
    }
    t.group = 8;
    if (const char *g = getenv("CTC_TC_GROUP")) { if (atoi(g) > 0) t.group = atoi(g); }
    p.tile = t;
}

}  // namespace

std::string parse_and_classify(const std::string &spec, Plan &p)
{
    p = Plan();
    p.spec = spec;
    size_t comma = spec.find(',');
    size_t arrow = spec.find("->");
    size_t arrow_len = 2;
    if (arrow == std::string::npos) {
        arrow = spec.find('=');
        arrow_len = 1;
    }
    if (comma == std::string::npos || arrow == std::string::npos || comma > arrow)
        return "malformed spec: expected \"A,B->C\"";
    std::string e;
    if (!(e = parse_operand(spec.substr(0, comma), p, p.a_modes, true)).empty()) return e;
    if (!(e = parse_operand(spec.substr(comma + 1, arrow - comma - 1), p, p.b_modes, true)).empty()) return e;
    if (!(e = parse_operand(spec.substr(arrow + arrow_len), p, p.c_modes, false)).empty()) return e;

    for (size_t id = 0; id < p.modes.size(); ++id) {
        Mode &m = p.modes[id];
        m.pa = position_of(p.a_modes, (int)id);
        m.pb = position_of(p.b_modes, (int)id);
        m.pc = position_of(p.c_modes, (int)id);
        bool ia = m.pa >= 0, ib = m.pb >= 0, ic = m.pc >= 0;
        if (ia && ib && ic) m.cls = kBatch;
        else if (ia && ib) m.cls = kContracted;
        else if (ia && ic) m.cls = kFreeM;
        else if (ib && ic) m.cls = kFreeN;
        else return std::string("mode '") + m.name + "' appears in one operand only";
    }
    for (int id : p.c_modes) {
        int cls = p.modes[id].cls;
        if (cls == kBatch) p.canon_batch.push_back(id);
        else if (cls == kFreeM) p.canon_m.push_back(id);
        else p.canon_n.push_back(id);
    }
    for (int id : p.a_modes)
        if (p.modes[id].cls == kContracted) p.canon_k.push_back(id);

    for (int id : p.canon_batch) p.perm_a.push_back(p.modes[id].pa);
    for (int id : p.canon_m) p.perm_a.push_back(p.modes[id].pa);
    for (int id : p.canon_k) p.perm_a.push_back(p.modes[id].pa);
    for (int id : p.canon_batch) p.perm_b.push_back(p.modes[id].pb);
    for (int id : p.canon_k) p.perm_b.push_back(p.modes[id].pb);
    for (int id : p.canon_n) p.perm_b.push_back(p.modes[id].pb);
    std::vector<int> gemm_order;
    gemm_order.insert(gemm_order.end(), p.canon_batch.begin(), p.canon_batch.end());
    gemm_order.insert(gemm_order.end(), p.canon_m.begin(), p.canon_m.end());
    gemm_order.insert(gemm_order.end(), p.canon_n.begin(), p.canon_n.end());
    for (int id : p.c_modes) p.perm_c.push_back(position_of(gemm_order, id));
    return "";
}

std::string build_plan(const std::string &spec, int n_ext, const char *mode_names,
                       const int64_t *extents, const int64_t *strides_a, const int64_t *strides_b,
                       const int64_t *strides_c, int dtype, Plan &p)
{
    std::string e = parse_and_classify(spec, p);
    if (!e.empty()) return e;
    if (dtype != kF64 && dtype != kF32 && dtype != kBF16) return "unknown dtype";
    p.dtype = dtype;
    for (Mode &m : p.modes) {
        m.extent = -1;
        for (int j = 0; j < n_ext; ++j)
            if (mode_names[j] == m.name) m.extent = extents[j];
        if (m.extent < 0) return std::string("missing extent for mode '") + m.name + "'";
        if (m.extent == 0) return std::string("zero extent for mode '") + m.name + "'";
    }
    // Sizes stay far inside int64: an operand has at most 2^46 elements (packed count, and the span its strides
    // address), a GEMM dimension at most 2^31 - 1 entries (one int64 table entry per row / column / k / batch).
    constexpr int64_t kMaxElems = int64_t(1) << 46, kMaxDim = 0x7FFFFFFF;
    auto product_within = [&](const std::vector<int> &modes, int64_t limit) {
        int64_t n = 1;
        for (int id : modes) {
            if (p.modes[id].extent > limit / n) return false;
            n *= p.modes[id].extent;
        }
        return true;
    };
    if (!product_within(p.a_modes, kMaxElems) || !product_within(p.b_modes, kMaxElems) || !product_within(p.c_modes, kMaxElems))
        return "operand too large (more than 2^46 elements)";
    if (!product_within(p.canon_batch, kMaxDim) || !product_within(p.canon_m, kMaxDim) || !product_within(p.canon_n, kMaxDim) ||
        !product_within(p.canon_k, kMaxDim))
        return "contraction too large (a GEMM dimension of 2^31 or more)";
    auto span_within = [&](const std::vector<int> &modes, const int64_t *given) {
        if (!given) return true;
        int64_t span = 1;
        for (size_t i = 0; i < modes.size(); ++i) {
            const int64_t e1 = p.modes[modes[i]].extent - 1, s = given[i];
            if (s < 0) return true;  // reported below
            if (e1 > 0 && s > (kMaxElems - span) / e1) return false;
            span += e1 * s;
        }
        return true;
    };
    if (!span_within(p.a_modes, strides_a) || !span_within(p.b_modes, strides_b) || !span_within(p.c_modes, strides_c))
        return "operand too large (its strides address more than 2^46 elements)";
    std::vector<int64_t> st;
    auto assign = [&](const std::vector<int> &modes, const int64_t *given, int which) {
        if (given) st.assign(given, given + modes.size()); else packed_strides(p, modes, st);
        for (size_t i = 0; i < modes.size(); ++i) {
            Mode &m = p.modes[modes[i]];
            (which == 0 ? m.sa : which == 1 ? m.sb : m.sc) = st[i];
        }
    };
    assign(p.a_modes, strides_a, 0);
    p.strides_a = st;
    assign(p.b_modes, strides_b, 1);
    p.strides_b = st;
    assign(p.c_modes, strides_c, 2);
    p.strides_c = st;
    for (const Mode &m : p.modes)
        if (m.sa < 0 || m.sb < 0 || m.sc < 0) return "negative strides are not supported";

    auto count = [&](const std::vector<int> &modes) {
        int64_t n = 1;
        for (int id : modes) n *= p.modes[id].extent;
        return n;
    };
    p.size_a = count(p.a_modes);
    p.size_b = count(p.b_modes);
    p.size_c = count(p.c_modes);

    const size_t nm = p.modes.size();
    std::vector<int64_t> sa(nm), sb(nm), sc(nm);
    for (size_t i = 0; i < nm; ++i) {
        sa[i] = p.modes[i].sa;
        sb[i] = p.modes[i].sb;
        sc[i] = p.modes[i].sc;
    }

    // Role assignment: the operand that owns C's unit-stride mode becomes GEMM "B",
    // so that the MFMA result tile (lanes along N) stores contiguous runs of C.
    int c_unit = unit_stride_mode(p, p.c_modes, sc);
    p.swapped = (c_unit >= 0 && p.modes[c_unit].cls == kFreeM);
    const std::vector<int64_t> &ga = p.swapped ? sb : sa;
    const std::vector<int64_t> &gb = p.swapped ? sa : sb;
    const std::vector<int> &ga_modes = p.swapped ? p.b_modes : p.a_modes;
    const std::vector<int> &gb_modes = p.swapped ? p.a_modes : p.b_modes;
    const std::vector<int> &m_ids = p.swapped ? p.canon_n : p.canon_m;
    const std::vector<int> &n_ids = p.swapped ? p.canon_m : p.canon_n;

    p.ex_n = fastest_first(p, n_ids, sc);
    p.ex_m = fastest_first(p, m_ids, ga);
    p.ex_batch = fastest_first(p, p.canon_batch, sc);
    int a_unit = unit_stride_mode(p, ga_modes, ga);
    int b_unit = unit_stride_mode(p, gb_modes, gb);
    bool a_unit_is_k = a_unit >= 0 && p.modes[a_unit].cls == kContracted;
    bool b_unit_is_k = b_unit >= 0 && p.modes[b_unit].cls == kContracted;
    p.ex_k = fastest_first(p, p.canon_k, (a_unit_is_k || !b_unit_is_k) ? ga : gb);

    p.M = count(p.ex_m);
    p.N = count(p.ex_n);
    p.K = count(p.ex_k);
    p.batch = count(p.ex_batch);

    const int vmax = 16 / dtype_size(dtype);
    p.layout_a = kLayoutScalar;
    p.vec_a = 1;
    if (a_unit >= 0) {
        int v = vector_width(p, ga_modes, ga, a_unit, vmax);
        if (v > 1 && !p.ex_k.empty() && a_unit == p.ex_k[0]) { p.layout_a = kLayoutKVec; p.vec_a = v; }
        else if (v > 1 && !p.ex_m.empty() && a_unit == p.ex_m[0]) { p.layout_a = kLayoutOuterVec; p.vec_a = v; }
    }
    p.layout_b = kLayoutScalar;
    p.vec_b = 1;
    if (b_unit >= 0) {
        int v = vector_width(p, gb_modes, gb, b_unit, vmax);
        if (v > 1 && !p.ex_k.empty() && b_unit == p.ex_k[0]) { p.layout_b = kLayoutKVec; p.vec_b = v; }
        else if (v > 1 && !p.ex_n.empty() && b_unit == p.ex_n[0]) { p.layout_b = kLayoutOuterVec; p.vec_b = v; }
    }

    // C can be stored in 16-byte vectors when its unit-stride mode leads the N modes and
    // every other C stride keeps the vectors aligned
    p.vec_c = 1;
    if (c_unit >= 0 && !p.ex_n.empty() && c_unit == p.ex_n[0] && vector_width(p, p.c_modes, sc, c_unit, vmax) == vmax)
        p.vec_c = vmax;

    {
        const int64_t per64 = 64 / dtype_size(dtype);
        p.c_sectors_whole = p.vec_c > 1 && p.modes[c_unit].extent % per64 == 0;
        for (int id : p.c_modes)
            if (id != c_unit && p.modes[id].extent > 1 && sc[id] % per64 != 0) p.c_sectors_whole = false;
    }

    build_table(p, p.ex_m, ga, p.M, p.row_a);
    build_table(p, p.ex_k, ga, p.K, p.k_a);
    build_table(p, p.ex_k, gb, p.K, p.k_b);
    auto affine = [&](const std::vector<int64_t> &tab) -> int64_t {
        if (tab.size() < 2) return 0;
        const int64_t s = tab[1];
        for (size_t k = 0; k < tab.size(); ++k)
            if (tab[k] != (int64_t)k * s) return 0;
        return s;
    };
    p.ka_stride = affine(p.k_a);
    p.kb_stride = affine(p.k_b);
    {
        auto affine_groups = [](const std::vector<int64_t> &tab, int64_t K, int group, int64_t &step) {
            if (K < group || K % group != 0) return false;
            step = tab[1] - tab[0];
            for (int64_t g = 0; g < K; g += group)
                for (int i = 1; i < group; ++i)
                    if (tab[g + i] != tab[g] + i * step) return false;
            return true;
        };
        p.k32_affine = affine_groups(p.k_a, p.K, 32, p.ka_in) && affine_groups(p.k_b, p.K, 32, p.kb_in);
        p.k16_affine = affine_groups(p.k_a, p.K, 16, p.ka_in) && affine_groups(p.k_b, p.K, 16, p.kb_in);
        p.k8_affine = affine_groups(p.k_a, p.K, 8, p.ka_in) && affine_groups(p.k_b, p.K, 8, p.kb_in);
        p.k4_affine = affine_groups(p.k_a, p.K, 4, p.ka_in) && affine_groups(p.k_b, p.K, 4, p.kb_in);
        const int64_t run = p.ex_k.empty() ? 0 : p.modes[p.ex_k[0]].extent;
        if (run >= 2 && run <= 0x7FFFFFFF && p.K % run == 0 && affine_groups(p.k_a, p.K, (int)run, p.ka_in) &&
            affine_groups(p.k_b, p.K, (int)run, p.kb_in))
            p.k_run = run;
    }
    build_table(p, p.ex_n, gb, p.N, p.col_b);
    for (int64_t v : p.row_a) p.max_row_a = v > p.max_row_a ? v : p.max_row_a;
    for (int64_t v : p.col_b) p.max_col_b = v > p.max_col_b ? v : p.max_col_b;
    build_table(p, p.ex_m, sc, p.M, p.row_c);
    build_table(p, p.ex_n, sc, p.N, p.col_c);
    build_table(p, p.ex_batch, ga, p.batch, p.bat_a);
    build_table(p, p.ex_batch, gb, p.batch, p.bat_b);
    build_table(p, p.ex_batch, sc, p.batch, p.bat_c);

    choose_tile(p);

    // Pathological strides for the fused kernel: a batch mode is the unit-stride mode of A, B and C, so
    // every element of every tile is a strided scalar access.  Explicit transposes to
    // [batch, M, K] / [batch, K, N] and back measure up to twice as fast there (64 x 256^3 fp64: 12.9 vs
    // 24.9 TFLOP/s), so ctc_tc_execute takes that path when the operands are packed and beta == 0.
    {
        auto packed = [&](const std::vector<int> &modes, const std::vector<int64_t> &strides) {
            int64_t s = 1;
            for (size_t i = modes.size(); i-- > 0;) {
                if (strides[i] != s) return false;
                s *= p.modes[modes[i]].extent;
            }
            return true;
        };
        const bool batch_fastest = a_unit >= 0 && a_unit == b_unit && a_unit == c_unit && p.modes[a_unit].cls == kBatch;
        p.prefer_transposes = batch_fastest && p.modes[a_unit].extent > 1 && p.M > 1 && p.N > 1 &&
                              packed(p.a_modes, p.strides_a) && packed(p.b_modes, p.strides_b) && packed(p.c_modes, p.strides_c);
    }

    // Element loads under a large contraction (see Plan::pad_transposes).  Thresholds from tools/odd_extent_probe.py:
    // the copies cost two passes over A and B at 3 - 4 TB/s, the GEMM gains 10 - 15 % in fp64 and 50 - 60 % in
    // fp32 (whose element loads are 4 bytes wide); bf16 element loads run at a quarter of the vector kernels' rate.
    // Problems on the small tile lose 10 - 30 % with the copies and keep the fused kernel.
    {
        auto packed = [&](const std::vector<int> &modes, const std::vector<int64_t> &strides) {
            int64_t s = 1;
            for (size_t i = modes.size(); i-- > 0;) {
                if (strides[i] != s) return false;
                s *= p.modes[modes[i]].extent;
            }
            return true;
        };
        static const bool off = getenv("CTC_TC_NO_PAD_TRANSPOSES") != nullptr;
        const int64_t unit = 64 / dtype_size(dtype);
        // CTC_TC_PAD_MIN_MN / CTC_TC_PAD_MIN_K: calibration knobs
        static const int64_t env_mn = getenv("CTC_TC_PAD_MIN_MN") ? atoll(getenv("CTC_TC_PAD_MIN_MN")) : 0;
        static const int64_t env_k = getenv("CTC_TC_PAD_MIN_K") ? atoll(getenv("CTC_TC_PAD_MIN_K")) : 0;
        // fp32 with an operand that is read by elements ACROSS its rows (its unit-stride mode is a free one, of an odd
        // extent): the copies pay from 512 x 512 on (979 x 2363 x 6616 `ka,kb`: 63 -> 99 TFLOP/s, `ak,kb` 81 -> 100;
        // 515 x 5001 x 1025 `ka,kb` 35 -> 57; two operands read along K tie or lose there, fp64 ties)
        const bool a_across = p.layout_a == kLayoutScalar && a_unit >= 0 && !a_unit_is_k;
        const bool b_across = p.layout_b == kLayoutScalar && b_unit >= 0 && !b_unit_is_k;
        const int64_t f32_mn = (dtype == kF32 && (a_across || b_across)) ? 512 : 1024;
        const int64_t min_mn = env_mn > 0 ? env_mn : (dtype == kF64 ? 2048 : f32_mn), min_k = env_k > 0 ? env_k : (dtype == kF64 ? 1024 : 512);
        p.k_padded = (p.K + unit - 1) / unit * unit;
        // bf16 also when the k tables are affine in groups shorter than 16 only (the fastest contracted mode's
        // extent is no multiple of 16): the ring kernel (K-tiles of 32, one table entry per tile or half tile) cannot
        // take the operands as they are, and one fused contracted mode is affine throughout
        // (measured on the headline contraction, tools/bf16_groups_probe.py: extent 56 loses 16 %, 72 / 88 / 104 gain
        // 4 / 12 / 28 %: hence the larger size)
        const bool bf16_short_groups = dtype == kBF16 && !p.k32_affine && !p.k16_affine && p.M >= 4096 && p.N >= 4096 &&
                                       p.K >= 4096 && getenv("CTC_TC_NO_PAD_BF16_GROUPS") == nullptr;
        // (fp64 operands with vectors but ragged K-tiles - extents 66, 70, 74 ... - gain -1 ... +4 % with the copies:
        // they keep the fused kernel)
        p.pad_transposes = !off && !p.prefer_transposes && (p.layout_a == kLayoutScalar || p.layout_b == kLayoutScalar || bf16_short_groups) &&
                           p.tile.bm >= 128 && p.M >= min_mn && p.N >= min_mn && p.K >= min_k &&
                           packed(p.a_modes, p.strides_a) && packed(p.b_modes, p.strides_b);
    }

    // Skinny path: a handful of output rows over a long N.  Needs 16-byte vectors along
    // an N mode in both B and C.
    p.kernel = kKernelGemm;
    const bool c_unit_in_n = c_unit >= 0 && p.modes[c_unit].cls != kBatch && p.modes[c_unit].cls != kContracted;
    const bool b_unit_in_n = b_unit >= 0 && std::find(p.ex_n.begin(), p.ex_n.end(), b_unit) != p.ex_n.end();
    if (dtype != kBF16 && p.M <= kSkinnyMaxM && p.M * p.K <= kSkinnyMaxMK && p.N >= 4096 && c_unit_in_n && b_unit_in_n &&
        getenv("CTC_TC_NO_SKINNY") == nullptr) {
        const int vb = vector_width(p, gb_modes, gb, b_unit, vmax);
        const int vc = vector_width(p, p.c_modes, sc, c_unit, vmax);
        if (vb == vmax && vc == vmax) {
            if (b_unit == c_unit) {
                p.kernel = kKernelSkinnySame;
            } else {
                p.kernel = kKernelSkinnyTranspose;
                const int64_t want = 16 * vmax;  // patch edge
                std::vector<int> by_b = fastest_first(p, p.ex_n, gb), by_c = p.ex_n;  // ex_n is in C order
                std::vector<int> xm, ym, om;
                std::vector<char> used(nm, 0);
                int64_t vol = 1;
                for (int id : by_c) {
                    if (vol >= want || id == b_unit) break;
                    ym.push_back(id);
                    used[id] = 1;
                    vol *= p.modes[id].extent;
                }
                vol = 1;
                for (int id : by_b) {
                    if (vol >= want || used[id]) break;
                    xm.push_back(id);
                    used[id] = 1;
                    vol *= p.modes[id].extent;
                }
                for (int id : by_c)
                    if (!used[id]) om.push_back(id);
                p.sk_X = count(xm);
                p.sk_Y = count(ym);
                p.sk_O = count(om);
                build_table(p, xm, gb, p.sk_X, p.sk_bx);
                build_table(p, xm, sc, p.sk_X, p.sk_cx);
                build_table(p, ym, gb, p.sk_Y, p.sk_by);
                build_table(p, ym, sc, p.sk_Y, p.sk_cy);
                build_table(p, om, gb, p.sk_O, p.sk_bo);
                build_table(p, om, sc, p.sk_O, p.sk_co);
            }
        }
    }
    return "";
}

std::string shard_plan(const Plan &full, char mode, int rank, int nranks, Plan &shard,
                       int64_t &off_a, int64_t &off_b, int64_t &off_c)
{
    int id = find_mode(full, mode);
    if (id < 0) return std::string("unknown mode '") + mode + "'";
    if (nranks <= 0 || rank < 0 || rank >= nranks) return "bad rank/nranks";
    const Mode &m = full.modes[id];
    if (m.extent % nranks) return "extent of the sharded mode must be a multiple of the rank count";
    std::string names;
    std::vector<int64_t> ext;
    for (const Mode &x : full.modes) {
        names.push_back(x.name);
        ext.push_back(x.name == mode ? x.extent / nranks : x.extent);
    }
    std::string e = build_plan(full.spec, (int)names.size(), names.data(), ext.data(),
                               full.strides_a.data(), full.strides_b.data(), full.strides_c.data(),
                               full.dtype, shard);
    if (!e.empty()) return e;
    int64_t first = (m.extent / nranks) * rank;
    off_a = first * m.sa;
    off_b = first * m.sb;
    off_c = first * m.sc;
    return "";
}

std::string describe(const Plan &p)
{
    static const char *cls_name[] = {"batch", "contracted", "free_m", "free_n"};
    static const char *lay_name[] = {"k_vec", "outer_vec", "scalar"};
    static const char *dt_name[] = {"f64", "f32", "bf16"};
    std::ostringstream o;
    o << "spec " << p.spec << " dtype " << dt_name[p.dtype] << "\n";
    o << "modes";
    for (const Mode &m : p.modes) o << " " << m.name << "=" << m.extent << ":" << cls_name[m.cls];
    o << "\n";
    auto names = [&](const std::vector<int> &v) {
        std::string s;
        for (int id : v) s.push_back(p.modes[id].name);
        return s.empty() ? std::string("-") : s;
    };
    auto ints = [&](const std::vector<int> &v) {
        std::string s;
        for (size_t i = 0; i < v.size(); ++i) s += (i ? "," : "") + std::to_string(v[i]);
        return s.empty() ? std::string("-") : s;
    };
    o << "canonical batch " << names(p.canon_batch) << " m " << names(p.canon_m) << " n "
      << names(p.canon_n) << " k " << names(p.canon_k) << "\n";
    o << "perm_a " << ints(p.perm_a) << " perm_b " << ints(p.perm_b) << " perm_c " << ints(p.perm_c) << "\n";
    o << "exec swapped " << (p.swapped ? 1 : 0) << " batch " << names(p.ex_batch) << " m "
      << names(p.ex_m) << " n " << names(p.ex_n) << " k " << names(p.ex_k) << " (fastest first)\n";
    o << "gemm M " << p.M << " N " << p.N << " K " << p.K << " batch " << p.batch << "\n";
    o << "layout_a " << lay_name[p.layout_a] << " vec " << p.vec_a << " layout_b "
      << lay_name[p.layout_b] << " vec " << p.vec_b << " c_vec " << p.vec_c << "\n";
    static const char *kern_name[] = {"mfma_gemm", "skinny_same", "skinny_transpose"};
    o << "kernel " << kern_name[p.kernel];
    if (p.kernel == kKernelSkinnyTranspose) o << " x " << p.sk_X << " y " << p.sk_Y << " o " << p.sk_O;
    o << "\n";
    o << "tile " << p.tile.bm << "x" << p.tile.bn << "x" << p.tile.bk << " threads " << p.tile.threads
      << " group " << p.tile.group << "\n";
    o << "transposes " << (p.prefer_transposes ? "explicit (A, B, C)" : p.pad_transposes ? "A and B into padded K-major rows" : "fused into the loads");
    if (p.pad_transposes) o << " of " << p.k_padded;
    o << "\n";
    return o.str();
}

}  // namespace ctc
