// C ABI of libctc_tc.so (declared in include/ctc_tc.h): thin host layer between
// plain-C callers and the HIP kernels.  No torch types, no CUDA shims.
#include "ctc_tc.h"

#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <list>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "api_internal.hpp"
#include "kernels.hpp"
#include "permute_plan.hpp"
#include "planner.hpp"

using ctc_internal::DeviceTables;

namespace {

thread_local std::string g_last_error;

int fail(int code, const std::string &msg)
{
    g_last_error = msg;
    return code;
}

int fail_hip(hipError_t e, const char *what)
{
    return fail(CTC_TC_EDEVICE, std::string(what) + ": " + hipGetErrorString(e));
}

}  // namespace

namespace ctc_internal {
// shared with the other translation units of the library
int set_last_error(int code, const std::string &msg) { return fail(code, msg); }
int set_last_hip_error(hipError_t e, const char *what) { return fail_hip(e, what); }
}  // namespace ctc_internal

namespace {

struct PermuteTables {
    int64_t *base = nullptr;
    ctc::PermuteParams params;
};

}  // namespace

namespace {

// Many small products under a batch mode (M, N <= 32): the tiled kernels would multiply mostly padding (a 64 x 64
// tile per 16 x 16 product: 1.9 TFLOP/s for 100,000 of them); the grouped kernel gives every product a wave of
// its own (4.8).  It reads operands with one unit stride and one leading dimension each, so the plan's tables must
// say exactly that: A(m, k) at m lda + k or at k lda + m, B(k, n) at k ldb + n or at n ldb + k, C(m, n) at
// m ldc + n.  Measured against the tiled kernel (tools/small_batched_probe.py): fp64 wins up to 32 x 32
// (x 1.3 - 4.5), loses from 40 x 40 on; fp32 ties at 32^3.
std::atomic<int> g_small_batched{1};  // 0: never (A/B runs)
std::atomic<int> g_small_equal{getenv("CTC_TC_NO_SMALL_EQUAL") ? 0 : 1};  // four products of at most 16 x 16 per wave
std::atomic<int64_t> g_launches_grouped{0};

// offsets a[i] + b[j] of an extent-na by extent-nb operand: unit stride along i and one leading dimension along j
// (ld >= na)?  An extent of one fits any stride.
bool unit_then_ld(const std::vector<int64_t> &a, int64_t na, const std::vector<int64_t> &b, int64_t nb, int64_t &ld)
{
    ld = nb > 1 ? b[1] : na;
    if (ld < na || ld > 0x7FFFFFFF) return false;
    for (int64_t i = 0; i < na; ++i)
        if (a[(size_t)i] != i) return false;
    for (int64_t j = 0; j < nb; ++j)
        if (b[(size_t)j] != j * ld) return false;
    return true;
}

bool small_batched(const ctc::Plan &p, int64_t &lda, int64_t &ldb, int64_t &ldc, bool &a_outer, bool &b_kmajor)
{
    if (p.kernel != ctc::kKernelGemm || (p.dtype != ctc::kF64 && p.dtype != ctc::kF32) || p.batch < 512) return false;
    if (p.M < 1 || p.N < 1 || p.K < 1 || p.M > 32 || p.N > 32) return false;
    // how long a K: products of at most 16 x 16 gain at every K measured (50,000 x 16 x 16 x 128: x 2.2, 10,000 x 16 x 16 x
    // 1024: x 1.9, fp32 x 1.5 - 1.6), 24 x 24 fp64 still x 1.4 at K = 96; 32 x 32 ties from K = 128 on and fp32 loses there
    const int64_t small = p.M > p.N ? p.M : p.N;
    const int64_t max_k = small <= 16 ? 4096 : (small <= 24 && p.dtype == ctc::kF64 ? 128 : 64);
    if (p.K > max_k) return false;
    if (p.dtype == ctc::kF32 && p.M * p.N > 768 && p.K > 16) return false;
    if (unit_then_ld(p.k_a, p.K, p.row_a, p.M, lda)) a_outer = false;       // [m][k]
    else if (unit_then_ld(p.row_a, p.M, p.k_a, p.K, lda)) a_outer = true;   // [k][m]
    else return false;
    if (unit_then_ld(p.col_b, p.N, p.k_b, p.K, ldb)) b_kmajor = false;      // [k][n]
    else if (unit_then_ld(p.k_b, p.K, p.col_b, p.N, ldb)) b_kmajor = true;  // [n][k]
    else return false;
    return unit_then_ld(p.col_c, p.N, p.row_c, p.M, ldc);
}

int upload_tables(ctc_tc_plan *pl, int device, DeviceTables &out, bool no_alloc = false)
{
    std::lock_guard<std::mutex> lock(pl->mu);
    auto it = pl->tables.find(device);
    if (it != pl->tables.end() && it->second.base) {
        out = it->second;
        return CTC_TC_OK;
    }
    if (no_alloc)
        return fail(CTC_TC_EINVAL, "the plan's tables are not resident on this device: call ctc_tc_plan_prepare first");
    const ctc::Plan &p = pl->plan;
    // k tables padded for the kernel's look-ahead (see kKTablePad)
    // Pad entry K+j repeats entry K-P+(j mod P), P = elements per 16-byte vector, so
    // that a vector load through a pad entry re-reads the operand's last whole vector
    // (same alignment, in bounds) instead of straddling the end of a row.
    std::vector<int64_t> k_a = p.k_a, k_b = p.k_b;
    {
        const int64_t vmax = 16 / ctc::dtype_size(p.dtype);
        const int64_t period = (p.K % vmax == 0) ? vmax : 1;
        for (int j = 0; j < ctc::kKTablePad; ++j) {
            k_a.push_back(p.k_a[(size_t)(p.K - period + (j % period))]);
            k_b.push_back(p.k_b[(size_t)(p.K - period + (j % period))]);
        }
    }
    constexpr int NT = 15;
    const std::vector<int64_t> *src[NT] = {&p.row_a, &k_a, &k_b, &p.col_b, &p.row_c,
                                           &p.col_c, &p.bat_a, &p.bat_b, &p.bat_c,
                                           &p.sk_bx, &p.sk_cx, &p.sk_by, &p.sk_cy, &p.sk_bo, &p.sk_co};
    size_t total = 0;
    size_t start[NT];
    for (int i = 0; i < NT; ++i) {
        start[i] = total;
        total += (src[i]->size() + 1) & ~size_t(1);  // keep every table 16-byte aligned
    }
    std::vector<int64_t> host(total, 0);
    for (int i = 0; i < NT; ++i) std::copy(src[i]->begin(), src[i]->end(), host.begin() + start[i]);
    DeviceTables t;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&t.base), total * sizeof(int64_t));
    if (e != hipSuccess) return fail_hip(e, "hipMalloc(offset tables)");
    e = hipMemcpy(t.base, host.data(), total * sizeof(int64_t), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        (void)hipFree(t.base);
        return fail_hip(e, "hipMemcpy(offset tables)");
    }
    t.row_a = t.base + start[0];
    t.k_a = t.base + start[1];
    t.k_b = t.base + start[2];
    t.col_b = t.base + start[3];
    t.row_c = t.base + start[4];
    t.col_c = t.base + start[5];
    t.bat_a = t.base + start[6];
    t.bat_b = t.base + start[7];
    t.bat_c = t.base + start[8];
    for (int i = 0; i < 6; ++i) t.sk[i] = t.base + start[9 + i];
    int64_t lda = 0, ldb = 0, ldc = 0;
    if (small_batched(p, lda, ldb, ldc, t.small_a_outer, t.small_b_kmajor)) {
        std::vector<ctc::SmallGemm> desc((size_t)p.batch);
        for (int64_t z = 0; z < p.batch; ++z)
            desc[(size_t)z] = ctc::SmallGemm{(int32_t)p.M, (int32_t)p.N, (int32_t)p.K, (int32_t)lda, (int32_t)ldb, (int32_t)ldc,
                                             p.bat_a[(size_t)z], p.bat_b[(size_t)z], p.bat_c[(size_t)z]};
        e = hipMalloc(reinterpret_cast<void **>(&t.small), desc.size() * sizeof(ctc::SmallGemm));
        if (e == hipSuccess) e = hipMemcpy(t.small, desc.data(), desc.size() * sizeof(ctc::SmallGemm), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            if (t.small) (void)hipFree(t.small);
            (void)hipFree(t.base);
            return fail_hip(e, "descriptors of the batch's small products");
        }
    }
    pl->tables[device] = t;
    out = t;
    return CTC_TC_OK;
}

void release_tables(std::map<int, DeviceTables> &tables)
{
    for (auto &kv : tables) {
        if (kv.second.base) (void)hipFree(kv.second.base);
        if (kv.second.ws) (void)hipFree(kv.second.ws);
        if (kv.second.small) (void)hipFree(kv.second.small);
        for (void *old : kv.second.retired) (void)hipFree(old);
        if (kv.second.ws_done) (void)hipEventDestroy(kv.second.ws_done);
    }
    tables.clear();
}

// Offset tables of mode permutations, cached per (device, element size, extents, permutation) and bounded: beyond
// kPermCacheMax entries the least recently used one is freed (hipFree waits for the device, so a
// kernel that still reads the evicted tables finishes first; a caller cycling through more shapes
// than that pays this wait, everyone else never reaches it).
constexpr size_t kPermCacheMax = 128;
struct PermCacheEntry {
    PermuteTables tables;
    std::list<std::string>::iterator lru;
};
std::mutex g_perm_mu;
std::map<std::string, PermCacheEntry> g_perm_cache;  // key: device|element size|extents|perm
std::list<std::string> g_perm_lru;                   // most recently used first

int get_permute_tables(int device, int rank, const int64_t *extents, const int *perm, int elem_size, PermuteTables &out,
                       int out_row_modes = 0, int64_t out_row_pitch = 0)
{
    std::string key = std::to_string(device) + "|" + std::to_string(elem_size) + "|";
    for (int i = 0; i < rank; ++i) key += std::to_string(extents[i]) + ",";
    key += "|";
    for (int i = 0; i < rank; ++i) key += std::to_string(perm[i]) + ",";
    if (out_row_modes > 0) key += "|rows " + std::to_string(out_row_modes) + " pitch " + std::to_string(out_row_pitch);
    std::lock_guard<std::mutex> lock(g_perm_mu);
    auto it = g_perm_cache.find(key);
    if (it != g_perm_cache.end()) {
        g_perm_lru.splice(g_perm_lru.begin(), g_perm_lru, it->second.lru);
        out = it->second.tables;
        return CTC_TC_OK;
    }
    while (g_perm_cache.size() >= kPermCacheMax) {
        auto victim = g_perm_cache.find(g_perm_lru.back());
        (void)hipFree(victim->second.tables.base);
        g_perm_cache.erase(victim);
        g_perm_lru.pop_back();
    }
    ctc::PermutePlan pp;
    std::string e = ctc::plan_permute(rank, extents, perm, elem_size, pp, out_row_modes, out_row_pitch);
    if (!e.empty()) return fail(CTC_TC_EINVAL, e);
    const std::vector<int64_t> *src[6] = {&pp.in_x, &pp.out_x, &pp.in_y, &pp.out_y, &pp.in_o, &pp.out_o};
    size_t total = 0, start[6];
    for (int i = 0; i < 6; ++i) {
        start[i] = total;
        total += (src[i]->size() + 1) & ~size_t(1);
    }
    std::vector<int64_t> host(total, 0);
    for (int i = 0; i < 6; ++i) std::copy(src[i]->begin(), src[i]->end(), host.begin() + start[i]);
    PermuteTables t;
    hipError_t he = hipMalloc(reinterpret_cast<void **>(&t.base), total * sizeof(int64_t));
    if (he != hipSuccess) return fail_hip(he, "hipMalloc(permute tables)");
    he = hipMemcpy(t.base, host.data(), total * sizeof(int64_t), hipMemcpyHostToDevice);
    if (he != hipSuccess) {
        (void)hipFree(t.base);
        return fail_hip(he, "hipMemcpy(permute tables)");
    }
    t.params.in_x = t.base + start[0];
    t.params.out_x = t.base + start[1];
    t.params.in_y = t.base + start[2];
    t.params.out_y = t.base + start[3];
    t.params.in_o = t.base + start[4];
    t.params.out_o = t.base + start[5];
    t.params.X = pp.X;
    t.params.Y = pp.Y;
    t.params.O = pp.O;
    t.params.same_fast = pp.same_fast ? 1 : 0;
    // A workgroup walks four consecutive o values when that lengthens the contiguous runs on one
    // side (the next o lies right behind the y run in the output, or behind the x run in the
    // input); otherwise one, which measures 9 % faster (128^4 fp32 abdc <- abcd: 5.2 -> 5.7 TB/s).
    t.params.o_per_block = (pp.O > 1 && (pp.out_o[1] == pp.Y || pp.in_o[1] == pp.X)) ? 4 : 1;
    t.params.vec_mode = pp.vec_mode;
    t.params.in_run = pp.in_run;
    t.params.out_run = pp.out_run;
    g_perm_lru.push_front(key);
    g_perm_cache[key] = PermCacheEntry{t, g_perm_lru.begin()};
    out = t;
    return CTC_TC_OK;
}

int current_device(int &device)
{
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) return fail_hip(e, "hipGetDevice (is a gfx950 GPU visible?)");
    return CTC_TC_OK;
}

template <typename V> int64_t copy_out(const V &v, int64_t *out, int64_t cap)
{
    const int64_t n = (int64_t)v.size();
    if (out)
        for (int64_t i = 0; i < n && i < cap; ++i) out[i] = (int64_t)v[(size_t)i];
    return n;
}

// Caller holds the plan's mutex.  A workspace that has to grow is replaced, not freed: kernels in
// flight may still use the old one, and an execution call must not wait for the device.
int grow_workspace(DeviceTables &dt, size_t bytes)
{
    if (dt.ws_bytes < bytes) {
        void *fresh = nullptr;
        hipError_t e = hipMalloc(&fresh, bytes);
        if (e != hipSuccess) return fail_hip(e, "hipMalloc(workspace)");
        if (dt.ws) dt.retired.push_back(dt.ws);
        dt.ws = fresh;
        dt.ws_bytes = bytes;
    }
    if (!dt.ws_done) {
        hipError_t e = hipEventCreateWithFlags(&dt.ws_done, hipEventDisableTiming);
        if (e != hipSuccess) return fail_hip(e, "hipEventCreate");
    }
    return CTC_TC_OK;
}

// batch entries one launch covers: the batch index is a grid dimension
constexpr int64_t kMaxBatchPerLaunch = 65535;

// When the persistent deferred-store kernel takes over from the classic one: the classic kernel loses
// |C| / (write bandwidth) per launch whatever K is, the persistent one runs its K loop with one wave per
// SIMD and leaves the last, partly filled round of tiles uncut.
// Measured on MI355X (tools/stream_cmp.py): the persistent kernel wins up to K = 128 in fp64 (K = 64: 41 ->
// 49 TFLOP/s, K = 128: 56 -> 59) and loses from K = 256 on (K = 512: 67 -> 64); the fp32 form wins at K = 64
// (44 -> 55) and is level at K = 128.
constexpr int64_t kStreamMaxK = 128;
constexpr double kStreamMinTiles = 2048.0;
int initial_stream_mode()
{
    const char *m = getenv("CTC_TC_STREAM");  // 0: never, force: whenever it can run the launch
    return !m ? 0 : (m[0] == '0' ? 1 : (m[0] == 'f' ? 2 : 0));
}
std::atomic<int> g_stream_mode{initial_stream_mode()};
std::atomic<int64_t> g_launches_classic{0}, g_launches_stream{0}, g_launches_stream_dma{0};  // since load, for tests
std::atomic<int> g_stream_f64_dma{1};  // 0: fp64 takes the register-staged persistent kernel too (A/B runs)
std::atomic<int> g_dma_gemm{1};        // 0: tiled GEMMs never take the LDS-DMA kernel (A/B runs)
std::atomic<int64_t> g_launches_dma_gemm{0};
bool plan_dma_ok(const ctc::Plan &p)
{
    return g_dma_gemm.load() != 0 && p.max_row_a < (int64_t(1) << 28) && p.max_col_b < (int64_t(1) << 28);
}
bool plan_k8_affine(const ctc::Plan &p) { return plan_dma_ok(p) && p.k8_affine; }
std::atomic<int64_t> g_stream_max_k{getenv("CTC_TC_STREAM_MAX_K") ? atoll(getenv("CTC_TC_STREAM_MAX_K")) : kStreamMaxK};

// The part of a launch's parameters that comes from the plan alone (shape, what the LDS-DMA kernels need
// to know about the k tables): everything schedule_gemm looks at.  One place, so that a schedule reported
// ahead of time (ctc_tc_plan_prepare, CTC_TC_Q_SCHEDULE) is the schedule of the kernel the launch takes.
void shape_from_plan(const ctc::Plan &p, int nbatch, ctc::GemmParams &g)
{
    g.M = p.M; g.N = p.N; g.K = p.K;
    g.batch = nbatch;
    g.group = p.tile.group;
    // one table entry per K-tile of 8 and 32-bit lane offsets: what the LDS-DMA kernel needs
    g.k8_affine = plan_k8_affine(p) ? 1 : 0;
    g.k4_affine = (plan_dma_ok(p) && p.k4_affine) ? 1 : 0;
    g.k16_affine = (g.k8_affine && p.k16_affine) ? 1 : 0;
    g.k32_affine = (g.k16_affine && p.k32_affine) ? 1 : 0;
    g.ka_in = p.ka_in;
    g.kb_in = p.kb_in;
    g.k_run = plan_dma_ok(p) ? p.k_run : 0;
}

// The split-K workspace a tiled-GEMM plan can ask for on the current device: the largest over the
// layout variants an execution may fall back to (unaligned operand bases take element loads).
size_t worst_splitk_bytes(const ctc::Plan &p)
{
    if (p.kernel != ctc::kKernelGemm) return 0;
    size_t worst = 0;
    // a batch of more entries than one launch takes: whole launches and the rest (the tiles of a batched launch
    // are scheduled as one list, so the two differ)
    const int64_t whole = p.batch < kMaxBatchPerLaunch ? p.batch : kMaxBatchPerLaunch, rest = p.batch % whole;
    for (int64_t nb : {whole, rest}) {
        if (nb < 1) continue;
        ctc::GemmParams g;
        shape_from_plan(p, (int)nb, g);
        for (int va = 0; va < 2; ++va)
            for (int vb = 0; vb < 2; ++vb) {
                const int la = va ? ctc::kLayoutScalar : p.layout_a, lb = vb ? ctc::kLayoutScalar : p.layout_b;
                ctc::GemmSchedule sch;
                if (ctc::schedule_gemm(p.dtype, la, lb, p.tile.bm, p.tile.bn, g, &sch) == hipSuccess && sch.ws_bytes > worst)
                    worst = sch.ws_bytes;
            }
    }
    return worst;
}

}  // namespace

namespace ctc_internal {

int launch_plan(ctc_tc_plan *pl, double alpha, const void *A, const void *B, double beta, void *C,
                hipStream_t stream, const Workspace *fixed_ws, bool no_alloc)
{
    const ctc::Plan &p = pl->plan;
    int device = 0;
    int rc = current_device(device);
    if (rc) return rc;
    DeviceTables t;
    rc = upload_tables(pl, device, t, no_alloc);
    if (rc) return rc;

    if (t.small && g_small_batched.load() != 0 && p.batch <= 0x7FFFFFFF) {
        // one wave per product, all of the batch in one launch (GEMM operands: A and B exchanged when the plan says so)
        hipError_t e = ctc::launch_grouped(p.dtype, t.small, (int)p.batch, p.swapped ? B : A, p.swapped ? A : B, C, alpha, beta, stream,
                                           t.small_a_outer, t.small_b_kmajor,
                                           // (four per wave pays where the fixed work per wave dominates: under K = 64 it ties or loses)
                                           p.M <= 16 && p.N <= 16 && p.K <= 32 && g_small_equal.load() != 0);
        if (e != hipSuccess) return fail_hip(e, "grouped kernel launch");
        g_launches_grouped.fetch_add(1);
        return CTC_TC_OK;
    }

    // more batch entries than one grid holds: one launch per slab of the batch tables
    for (int64_t b0 = 0; b0 < p.batch; b0 += kMaxBatchPerLaunch) {
    const int nb = (int)(p.batch - b0 < kMaxBatchPerLaunch ? p.batch - b0 : kMaxBatchPerLaunch);
    const void *ga = p.swapped ? B : A, *gb = p.swapped ? A : B;
    if (p.kernel != ctc::kKernelGemm &&
        ((reinterpret_cast<uintptr_t>(gb) | reinterpret_cast<uintptr_t>(C)) & 15) == 0) {
        ctc::SkinnyParams s;
        s.A = ga; s.B = gb; s.C = C;
        s.row_a = t.row_a; s.k_a = t.k_a; s.k_b = t.k_b; s.row_c = t.row_c;
        s.bat_a = t.bat_a + b0; s.bat_b = t.bat_b + b0; s.bat_c = t.bat_c + b0;
        s.M = (int)p.M; s.K = (int)p.K; s.batch = nb;
        s.alpha = alpha; s.beta = beta;
        if (p.kernel == ctc::kKernelSkinnyTranspose) {
            s.transpose = 1;
            s.bx = t.sk[0]; s.cx = t.sk[1]; s.by = t.sk[2]; s.cy = t.sk[3]; s.bo = t.sk[4]; s.co = t.sk[5];
            s.X = p.sk_X; s.Y = p.sk_Y; s.O = p.sk_O;
        } else {
            s.transpose = 0;
            s.bx = t.col_b; s.cx = t.col_c;
            s.X = p.N;
        }
        s.stream_hint = p.c_sectors_whole && (reinterpret_cast<uintptr_t>(C) & 63) == 0 &&
                        (double)p.N * (double)nb * (double)(p.K + p.M) * (double)ctc::dtype_size(p.dtype) > 256.0 * 1048576.0;
        hipError_t e = ctc::launch_skinny(p.dtype, s, stream);
        if (e != hipSuccess) return fail_hip(e, "skinny contraction kernel launch");
        continue;
    }

    ctc::GemmParams g;
    g.A = ga;
    g.B = gb;
    g.C = C;
    g.row_a = t.row_a; g.k_a = t.k_a; g.k_b = t.k_b; g.col_b = t.col_b;
    g.row_c = t.row_c; g.col_c = t.col_c;
    g.bat_a = t.bat_a + b0; g.bat_b = t.bat_b + b0; g.bat_c = t.bat_c + b0;
    shape_from_plan(p, nb, g);
    g.alpha = alpha; g.beta = beta;
    // 16-byte stores of C need an aligned base as well
    g.c_vec = (p.vec_c > 1 && (reinterpret_cast<uintptr_t>(C) & 15) == 0) ? p.vec_c : 1;
    // a result larger than the 256 MiB Infinity Cache cannot stay resident for a consumer anyway
    // ... and the hint pays only when the tiles' stores cover whole 64-byte sectors (free extents 42 / 44 / 46 under
    // the config-4 contraction: 25 - 30 TFLOP/s with the hint, 33 - 35 without; 40 / 48: 41 / 49 with, 38 without)
    g.c_stream = (double)p.M * (double)p.N * (double)p.batch * (double)ctc::dtype_size(p.dtype) > 256.0 * 1048576.0 &&
                 p.c_sectors_whole && (reinterpret_cast<uintptr_t>(C) & 63) == 0;
    {
        static const int force = getenv("CTC_TC_C_STREAM") ? atoi(getenv("CTC_TC_C_STREAM")) : -1;  // A/B runs
        if (force >= 0) g.c_stream = force;
    }

    // vector loads need 16-byte aligned bases; otherwise fall back to element loads
    int la = p.layout_a, lb = p.layout_b;
    if (la != ctc::kLayoutScalar && (reinterpret_cast<uintptr_t>(g.A) & 15)) la = ctc::kLayoutScalar;
    if (lb != ctc::kLayoutScalar && (reinterpret_cast<uintptr_t>(g.B) & 15)) lb = ctc::kLayoutScalar;

    // A short K under a large C: the persistent kernel that deals the stores of a finished tile out behind
    // the MFMAs of the next one (gemm_stream_kernels.hip); see ctc_tc_set_option("stream_mode").
    {
        const int mode = g_stream_mode.load();
        const bool off = mode == 1, force = mode == 2;
        const int64_t max_k = g_stream_max_k.load();
        const double tiles128 = ((double)p.M / 128.0) * ((double)p.N / 128.0);
        if (!off && ctc::stream_gemm_supported(p.dtype, la, lb, p.tile.bm, p.tile.bn, g) &&
            (force || (p.K <= (p.dtype == ctc::kF32 && max_k > 96 ? 96 : max_k) && tiles128 >= kStreamMinTiles))) {
            g.ka_stride = p.ka_stride;
            g.kb_stride = p.kb_stride;
            const bool dma = p.dtype == ctc::kF64 && g_stream_f64_dma.load() != 0 && ctc::stream_gemm_f64_dma_supported(la, lb, g);
            // fp64 without the LDS-DMA form (a K that is no multiple of its K-tile, k offsets with more than one
            // stride): the register-staged persistent form is slower than the one-tile-per-workgroup kernels
            // (K = 40 under 31,250 tiles: 19.8 against 36.6 TFLOP/s), so only a forced mode takes it
            // the fp64 LDS-DMA form stores with the streaming hint unconditionally: not for a C whose runs end
            // inside 64-byte sectors
            if ((dma && (g.c_stream || force)) || p.dtype != ctc::kF64 || (force && !dma)) {
                hipError_t es = dma ? ctc::launch_gemm_stream_f64_dma(la, lb, g, stream)
                                    : ctc::launch_gemm_stream(p.dtype, la, lb, g, stream);
                if (es != hipSuccess) return fail_hip(es, "persistent contraction kernel launch");
                (dma ? g_launches_stream_dma : g_launches_stream).fetch_add(1);
                continue;
            }
        }
    }

    if (ctc::dma_gemm_supported(p.dtype, la, lb, p.tile.bm, p.tile.bn, g)) g_launches_dma_gemm.fetch_add(1);
    else g_launches_classic.fetch_add(1);
    ctc::GemmSchedule sch;
    hipError_t e = ctc::schedule_gemm(p.dtype, la, lb, p.tile.bm, p.tile.bn, g, &sch);
    if (e != hipSuccess) return fail_hip(e, "contraction schedule");
    if (sch.ws_bytes > 0 && fixed_ws) {
        // caller-provided workspace: nothing is allocated, nothing is ordered (the caller's business)
        if (sch.ws_bytes > fixed_ws->fixed_bytes)
            return fail(CTC_TC_EUNSUPPORTED, "this contraction needs " + std::to_string(sch.ws_bytes) +
                                                 " bytes of split-K scratch, the fixed scratch buffer holds " +
                                                 std::to_string(fixed_ws->fixed_bytes));
        e = ctc::launch_gemm(p.dtype, la, lb, p.tile.bm, p.tile.bn, g, sch, fixed_ws->fixed, stream);
        if (e != hipSuccess) return fail_hip(e, "contraction kernel launch");
        continue;
    }
    if (sch.ws_bytes > 0) {
        std::lock_guard<std::mutex> lock(pl->mu);
        DeviceTables &dt = pl->tables[device];
        if (dt.ws_used && dt.ws_stream != stream) {
            e = hipStreamWaitEvent(stream, dt.ws_done, 0);
            if (e != hipSuccess) return fail_hip(e, "hipStreamWaitEvent");
        }
        int wrc = grow_workspace(dt, sch.ws_bytes);
        if (wrc) return wrc;
        pl->pads_zeroed.erase(device);  // partial sums overwrite what a padded-copy execution had cleared there
        e = ctc::launch_gemm(p.dtype, la, lb, p.tile.bm, p.tile.bn, g, sch, dt.ws, stream);
        if (e != hipSuccess) return fail_hip(e, "contraction kernel launch");
        (void)hipEventRecord(dt.ws_done, stream);
        dt.ws_stream = stream;
        dt.ws_used = true;
        continue;
    }
    e = ctc::launch_gemm(p.dtype, la, lb, p.tile.bm, p.tile.bn, g, sch, nullptr, stream);
    if (e != hipSuccess) return fail_hip(e, "contraction kernel launch");
    }
    return CTC_TC_OK;
}

}  // namespace ctc_internal

using ctc_internal::launch_plan;

namespace {

// ---- fixed scratch of ctc_tc_execute_noalloc ---------------------------------------------------------
// One buffer per GPU, part of the library's code object, so the entry point never allocates.  It takes
// the split-K partial sums (every chunk slab is written in full before the fix-up kernel reads it, so
// nothing has to be cleared between calls).  `scratch` is a host-side stand-in for the symbol: the
// device address is looked up once per device (scratch_setup, reached from ctc_tc_plan_prepare) and only
// that address is ever handed to a kernel.
// 64 MiB: one 128 x 128 fp64 slab per workgroup slot of the chip (512): the most a stream-K schedule of the
// LDS-DMA kernel asks for, and what the extent-96 headline contraction needs; schedules that need more (many
// equal K chunks) are refused.
constexpr size_t kScratchBytes = size_t(64) << 20;
constexpr int kMaxDevices = 64;
static __device__ __align__(256) unsigned char scratch[kScratchBytes];
std::atomic<void *> g_scratch_ptr[kMaxDevices];  // written once per device (scratch_setup), read by every execution
std::mutex g_scratch_mu;

// caller has made `device` current
int scratch_setup(int device)
{
    if (device < 0 || device >= kMaxDevices) return CTC_TC_OK;  // such a device simply has no no-alloc path
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (g_scratch_ptr[device].load()) return CTC_TC_OK;
    void *ptr = nullptr;
    hipError_t e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return fail_hip(e, "hipGetSymbolAddress(scratch)");
    g_scratch_ptr[device].store(ptr);
    return CTC_TC_OK;
}

// same contraction on operands already in [batch,M,K] / [batch,K,N] / [batch,M,N] order
int ensure_packed_plan(ctc_tc_plan *plan)
{
    const ctc::Plan &p = plan->plan;
    std::lock_guard<std::mutex> lock(plan->mu);
    if (plan->packed) return CTC_TC_OK;
    std::string sa, sb, sc, names;
    std::vector<int64_t> ext;
    auto add = [&](std::string &s, const std::vector<int> &ids) {
        for (int id : ids) s.push_back(p.modes[id].name);
    };
    add(sa, p.canon_batch); add(sa, p.canon_m); add(sa, p.canon_k);
    add(sb, p.canon_batch); add(sb, p.canon_k); add(sb, p.canon_n);
    add(sc, p.canon_batch); add(sc, p.canon_m); add(sc, p.canon_n);
    for (auto &m : p.modes) { names.push_back(m.name); ext.push_back(m.extent); }
    std::unique_ptr<ctc_tc_plan> q(new ctc_tc_plan());
    std::string e = ctc::build_plan(sa + "," + sb + "->" + sc, (int)names.size(), names.data(),
                                    ext.data(), nullptr, nullptr, nullptr, p.dtype, q->plan);
    if (!e.empty()) return fail(CTC_TC_EINVAL, e);
    plan->packed = std::move(q);
    return CTC_TC_OK;
}

// Plan::pad_transposes: the GEMM on the padded K-major copies.  Operand modes: batch, the M (N) modes in canonical
// order, ONE contracted mode of extent k_padded (the contracted modes of the original, fused in A's order and
// padded with zeros); C as the caller gave it, strides included.
std::atomic<int> g_pad_transposes{1};          // 0: never (A/B runs)
std::atomic<int64_t> g_padded_executions{0};  // since load, for tests
int ensure_padded_plan(ctc_tc_plan *plan)
{
    const ctc::Plan &p = plan->plan;
    std::lock_guard<std::mutex> lock(plan->mu);
    if (plan->padded) return CTC_TC_OK;
    char kname = 0;
    for (const char *c = "abcdefghijklmnopqrstuvwxyzABCDEFGHIJKLMNOPQRSTUVWXYZ"; *c && !kname; ++c) {
        bool used = false;
        for (auto &m : p.modes) used = used || m.name == *c;
        if (!used) kname = *c;
    }
    if (!kname) return fail(CTC_TC_EUNSUPPORTED, "no letter left for the fused contracted mode");
    std::string sa, sb, sc, names;
    std::vector<int64_t> ext;
    auto add = [&](std::string &s, const std::vector<int> &ids) {
        for (int id : ids) s.push_back(p.modes[id].name);
    };
    add(sa, p.canon_batch); add(sa, p.canon_m); sa.push_back(kname);
    add(sb, p.canon_batch); add(sb, p.canon_n); sb.push_back(kname);
    add(sc, p.c_modes);
    for (auto &m : p.modes)
        if (m.cls != ctc::kContracted) { names.push_back(m.name); ext.push_back(m.extent); }
    names.push_back(kname);
    ext.push_back(p.k_padded);
    std::unique_ptr<ctc_tc_plan> q(new ctc_tc_plan());
    std::string e = ctc::build_plan(sa + "," + sb + "->" + sc, (int)names.size(), names.data(), ext.data(), nullptr, nullptr,
                                    p.strides_c.empty() ? nullptr : p.strides_c.data(), p.dtype, q->plan);
    if (!e.empty()) return fail(CTC_TC_EINVAL, e);
    plan->perm_bn.clear();
    for (int id : p.canon_batch) plan->perm_bn.push_back(p.modes[id].pb);
    for (int id : p.canon_n) plan->perm_bn.push_back(p.modes[id].pb);
    for (int id : p.canon_k) plan->perm_bn.push_back(p.modes[id].pb);
    plan->padded = std::move(q);
    return CTC_TC_OK;
}

// bytes of the two padded copies
struct PaddedBytes {
    int64_t a, b;
};
PaddedBytes padded_bytes(const ctc::Plan &p)
{
    auto rows = [&](const std::vector<int> &ids) {
        int64_t n = 1;
        for (int id : p.canon_batch) n *= p.modes[id].extent;
        for (int id : ids) n *= p.modes[id].extent;
        return n;
    };
    const int64_t es = ctc::dtype_size(p.dtype);
    auto pad = [](int64_t b) { return (b + 255) & ~int64_t(255); };
    return {pad(rows(p.canon_m) * p.k_padded * es), pad(rows(p.canon_n) * p.k_padded * es)};
}

// extents of the three explicit transposes: A as given, B as given, [batch,M,N] in GEMM order
struct TransposeShapes {
    std::vector<int64_t> ea, eb, eg;
};
TransposeShapes transpose_shapes(const ctc::Plan &p)
{
    auto ext_of = [&](const std::vector<int> &ids) {
        std::vector<int64_t> e;
        for (int id : ids) e.push_back(p.modes[id].extent);
        return e;
    };
    TransposeShapes t;
    t.ea = ext_of(p.a_modes);
    t.eb = ext_of(p.b_modes);
    std::vector<int> gemm_order;
    gemm_order.insert(gemm_order.end(), p.canon_batch.begin(), p.canon_batch.end());
    gemm_order.insert(gemm_order.end(), p.canon_m.begin(), p.canon_m.end());
    gemm_order.insert(gemm_order.end(), p.canon_n.begin(), p.canon_n.end());
    t.eg = ext_of(gemm_order);
    return t;
}

// The workspace of the padded copies on `device`, grown and ordered like the split-K workspace; the elements
// between a row's end and the next row are cleared once per workspace (the copies never write them, both
// operands hold zeros there, so they add 0 x 0 to every sum).  Caller holds no lock.
int padded_workspace(ctc_tc_plan *plan, int device, hipStream_t s, bool sync_clear, char *&wa, char *&wb)
{
    const PaddedBytes pb = padded_bytes(plan->plan);
    std::lock_guard<std::mutex> lock(plan->mu);
    DeviceTables &dt = plan->tables[device];
    if (dt.ws_used && dt.ws_stream != s && !sync_clear) {
        hipError_t e = hipStreamWaitEvent(s, dt.ws_done, 0);
        if (e != hipSuccess) return fail_hip(e, "hipStreamWaitEvent");
    }
    int rc = grow_workspace(dt, (size_t)(pb.a + pb.b));
    if (rc) return rc;
    if (plan->pads_zeroed[device] != dt.ws) {
        hipError_t e = sync_clear ? hipMemset(dt.ws, 0, (size_t)(pb.a + pb.b)) : hipMemsetAsync(dt.ws, 0, (size_t)(pb.a + pb.b), s);
        if (e != hipSuccess) return fail_hip(e, "clearing the padded copies");
        plan->pads_zeroed[device] = dt.ws;
    }
    wa = static_cast<char *>(dt.ws);
    wb = wa + pb.a;
    return CTC_TC_OK;
}

// tables of the two copies: A -> [batch, M, K | pad], B -> [batch, N, K | pad]
int padded_permute_tables(ctc_tc_plan *plan, int device, PermuteTables &ta, PermuteTables &tb)
{
    const ctc::Plan &p = plan->plan;
    const TransposeShapes ts = transpose_shapes(p);
    const int es = (int)ctc::dtype_size(p.dtype), nk = (int)p.canon_k.size();
    int rc = get_permute_tables(device, (int)ts.ea.size(), ts.ea.data(), p.perm_a.data(), es, ta, nk, p.k_padded);
    if (rc) return rc;
    return get_permute_tables(device, (int)ts.eb.size(), ts.eb.data(), plan->perm_bn.data(), es, tb, nk, p.k_padded);
}

// Plan::pad_transposes: copy A and B into padded K-major rows, multiply those, C in place.  Executions of one
// plan are serialised on the host here and ordered on the device through the workspace's event.
int execute_padded(ctc_tc_plan *plan, double alpha, const void *A, const void *B, double beta, void *C, hipStream_t s)
{
    int device = 0;
    int rc = current_device(device);
    if (rc) return rc;
    rc = ensure_padded_plan(plan);
    if (rc) return rc;
    std::lock_guard<std::mutex> serial(plan->transposes_mu);
    char *wa = nullptr, *wb = nullptr;
    rc = padded_workspace(plan, device, s, false, wa, wb);
    if (rc) return rc;
    PermuteTables ta, tb;
    rc = padded_permute_tables(plan, device, ta, tb);
    if (rc) return rc;
    const int es = (int)ctc::dtype_size(plan->plan.dtype);
    ctc::PermuteParams pa = ta.params, pbm = tb.params;
    pa.in = A;
    pa.out = wa;
    pbm.in = B;
    pbm.out = wb;
    hipError_t e = ctc::launch_permute(es, pa, s);
    if (e == hipSuccess) e = ctc::launch_permute(es, pbm, s);
    if (e != hipSuccess) return fail_hip(e, "copy into padded rows");
    rc = launch_plan(plan->padded.get(), alpha, wa, wb, beta, C, s);
    {
        std::lock_guard<std::mutex> lock(plan->mu);
        DeviceTables &dt = plan->tables[device];
        (void)hipEventRecord(dt.ws_done, s);
        dt.ws_stream = s;
        dt.ws_used = true;
    }
    if (rc == CTC_TC_OK) g_padded_executions.fetch_add(1);
    return rc;
}

bool takes_padded_path(const ctc_tc_plan *plan) { return plan->plan.pad_transposes && g_pad_transposes.load() != 0; }

}  // namespace

extern "C" {

int ctc_tc_version(void) { return 100; }

const char *ctc_tc_last_error(void) { return g_last_error.c_str(); }

int ctc_tc_set_option(const char *name, int64_t value)
{
    if (!name) return fail(CTC_TC_EINVAL, "null option name");
    const std::string n(name);
    if (n == "stream_mode") {
        if (value < 0 || value > 2) return fail(CTC_TC_EINVAL, "stream_mode: 0 automatic, 1 never, 2 whenever possible");
        g_stream_mode.store((int)value);
        return CTC_TC_OK;
    }
    if (n == "dma_gemm") {
        g_dma_gemm.store(value != 0);
        return CTC_TC_OK;
    }
    if (n == "stream_f64_dma") {
        g_stream_f64_dma.store(value != 0);
        return CTC_TC_OK;
    }
    if (n == "stream_max_k") {
        g_stream_max_k.store(value);
        return CTC_TC_OK;
    }
    if (n == "stream_k") {
        ctc::dma_gemm_set_stream_k((int)value);
        return CTC_TC_OK;
    }
    if (n == "stream_k_group") {
        ctc::dma_gemm_set_stream_k_group((int)value);
        return CTC_TC_OK;
    }
    if (n == "pad_transposes") {
        g_pad_transposes.store(value != 0);
        return CTC_TC_OK;
    }
    if (n == "small_batched") {
        g_small_batched.store(value != 0);
        return CTC_TC_OK;
    }
    return fail(CTC_TC_EINVAL, "unknown option '" + n + "'");
}

int64_t ctc_tc_get_option(const char *name)
{
    if (!name) return fail(CTC_TC_EINVAL, "null option name");
    const std::string n(name);
    if (n == "stream_mode") return g_stream_mode.load();
    if (n == "stream_max_k") return g_stream_max_k.load();
    if (n == "stream_f64_dma") return g_stream_f64_dma.load();
    if (n == "dma_gemm") return g_dma_gemm.load();
    if (n == "stream_k") return ctc::dma_gemm_get_stream_k();
    if (n == "stream_k_group") return ctc::dma_gemm_get_stream_k_group();
    if (n == "pad_transposes") return g_pad_transposes.load();
    if (n == "small_batched") return g_small_batched.load();
    if (n == "launches_grouped") return g_launches_grouped.load();
    if (n == "padded_executions") return g_padded_executions.load();
    if (n == "launches_dma_gemm") return g_launches_dma_gemm.load();
    if (n == "launches_bf16_ring") return ctc::bf16_ring_launches();
    if (n == "launches_classic") return g_launches_classic.load();
    if (n == "launches_stream") return g_launches_stream.load();
    if (n == "launches_stream_f64_dma") return g_launches_stream_dma.load();
    return fail(CTC_TC_EINVAL, "unknown option '" + n + "'");
}

int ctc_tc_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int ctc_tc_plan_create(const char *spec, int n_ext, const char *mode_names, const int64_t *extents,
                       const int64_t *strides_a, const int64_t *strides_b, const int64_t *strides_c,
                       int dtype, ctc_tc_plan **plan)
{
    CTC_ABI_TRY
    if (!spec || !plan || n_ext < 0 || (n_ext > 0 && (!mode_names || !extents)))
        return fail(CTC_TC_EINVAL, "null argument");
    std::unique_ptr<ctc_tc_plan> pl(new ctc_tc_plan());
    std::string e = ctc::build_plan(spec, n_ext, mode_names, extents, strides_a, strides_b, strides_c,
                                    dtype, pl->plan);
    if (!e.empty()) return fail(e.find("spec") != std::string::npos || e.find("mode") != std::string::npos
                                    ? CTC_TC_ESPEC : CTC_TC_EINVAL, e);
    *plan = pl.release();
    return CTC_TC_OK;
    CTC_ABI_CATCH
}

int ctc_tc_plan_destroy(ctc_tc_plan *plan)
{
    if (!plan) return CTC_TC_OK;
    release_tables(plan->tables);
    if (plan->packed) release_tables(plan->packed->tables);
    delete plan;
    return CTC_TC_OK;
}

int ctc_tc_plan_describe(const ctc_tc_plan *plan, char *buf, size_t buflen)
{
    CTC_ABI_TRY
    if (!plan) return fail(CTC_TC_EINVAL, "null plan");
    std::string s = ctc::describe(plan->plan);
    if (buf && buflen) {
        size_t n = s.size() < buflen - 1 ? s.size() : buflen - 1;
        memcpy(buf, s.data(), n);
        buf[n] = 0;
    }
    return (int)s.size();
    CTC_ABI_CATCH
}

int64_t ctc_tc_plan_get(const ctc_tc_plan *plan, int query, int64_t *out, int64_t cap)
{
    CTC_ABI_TRY
    if (!plan) return fail(CTC_TC_EINVAL, "null plan");
    const ctc::Plan &p = plan->plan;
    std::vector<int64_t> v;
    switch (query) {
    case CTC_TC_Q_NAMES: for (auto &m : p.modes) v.push_back((int64_t)m.name); break;
    case CTC_TC_Q_CLASSES: for (auto &m : p.modes) v.push_back(m.cls); break;
    case CTC_TC_Q_EXTENTS: for (auto &m : p.modes) v.push_back(m.extent); break;
    case CTC_TC_Q_CANON_BATCH: return copy_out(p.canon_batch, out, cap);
    case CTC_TC_Q_CANON_M: return copy_out(p.canon_m, out, cap);
    case CTC_TC_Q_CANON_N: return copy_out(p.canon_n, out, cap);
    case CTC_TC_Q_CANON_K: return copy_out(p.canon_k, out, cap);
    case CTC_TC_Q_PERM_A: return copy_out(p.perm_a, out, cap);
    case CTC_TC_Q_PERM_B: return copy_out(p.perm_b, out, cap);
    case CTC_TC_Q_PERM_C: return copy_out(p.perm_c, out, cap);
    case CTC_TC_Q_SWAPPED: v.push_back(p.swapped ? 1 : 0); break;
    case CTC_TC_Q_EXEC_BATCH: return copy_out(p.ex_batch, out, cap);
    case CTC_TC_Q_EXEC_M: return copy_out(p.ex_m, out, cap);
    case CTC_TC_Q_EXEC_N: return copy_out(p.ex_n, out, cap);
    case CTC_TC_Q_EXEC_K: return copy_out(p.ex_k, out, cap);
    case CTC_TC_Q_GEMM_DIMS: v = {p.M, p.N, p.K, p.batch}; break;
    case CTC_TC_Q_TILE:
        v = {p.tile.bm, p.tile.bn, p.tile.bk, p.tile.threads, p.layout_a, p.layout_b,
             p.vec_a, p.vec_b, p.tile.group, p.kernel, p.vec_c, p.prefer_transposes ? 1 : (p.pad_transposes ? 2 : 0)};
        break;
    case CTC_TC_Q_ROW_A: return copy_out(p.row_a, out, cap);
    case CTC_TC_Q_K_A: return copy_out(p.k_a, out, cap);
    case CTC_TC_Q_K_B: return copy_out(p.k_b, out, cap);
    case CTC_TC_Q_COL_B: return copy_out(p.col_b, out, cap);
    case CTC_TC_Q_ROW_C: return copy_out(p.row_c, out, cap);
    case CTC_TC_Q_COL_C: return copy_out(p.col_c, out, cap);
    case CTC_TC_Q_BAT_A: return copy_out(p.bat_a, out, cap);
    case CTC_TC_Q_BAT_B: return copy_out(p.bat_b, out, cap);
    case CTC_TC_Q_BAT_C: return copy_out(p.bat_c, out, cap);
    case CTC_TC_Q_FLOPS: v.push_back(2 * p.M * p.N * p.K * p.batch); break;
    case CTC_TC_Q_SCHEDULE: {
        // needs a device: the split of the last round of tiles depends on the resident workgroups
        if (p.kernel != ctc::kKernelGemm) return fail(CTC_TC_EUNSUPPORTED, "not a tiled GEMM plan");
        int device = 0;
        int rc = current_device(device);
        if (rc) return rc;
        // the schedule of the kernel a launch with aligned operand bases takes (launch_plan fills the same
        // fields from the plan), and the largest workspace over the fall-backs to element loads
        ctc::GemmParams g;
        shape_from_plan(p, (int)(p.batch < kMaxBatchPerLaunch ? p.batch : kMaxBatchPerLaunch), g);
        ctc::GemmSchedule sch;
        hipError_t e = ctc::schedule_gemm(p.dtype, p.layout_a, p.layout_b, p.tile.bm, p.tile.bn, g, &sch);
        if (e != hipSuccess) return fail_hip(e, "contraction schedule");
        v = {sch.nbm, sch.nbn, sch.dp_tiles, sch.tail_tiles, sch.split, (int64_t)sch.ws_bytes,
             (int64_t)ctc::dma_gemm_kind(p.dtype, p.layout_a, p.layout_b, p.tile.bm, p.tile.bn, g),
             (int64_t)worst_splitk_bytes(p), sch.sk_units, sch.flat_batch};
        break;
    }
    case CTC_TC_Q_SPANS: {
        int64_t a = 1, b = 1, c = 1;  // a mode an operand lacks has stride 0 there
        for (auto &m : p.modes) {
            a += (m.extent - 1) * m.sa;
            b += (m.extent - 1) * m.sb;
            c += (m.extent - 1) * m.sc;
        }
        v = {a, b, c};
        break;
    }
    default: return fail(CTC_TC_EINVAL, "unknown query");
    }
    return copy_out(v, out, cap);
    CTC_ABI_CATCH
}

int ctc_tc_plan_shard(const ctc_tc_plan *plan, char mode, int rank, int nranks, ctc_tc_plan **shard,
                      int64_t *off_a, int64_t *off_b, int64_t *off_c)
{
    CTC_ABI_TRY
    if (!plan || !shard) return fail(CTC_TC_EINVAL, "null argument");
    std::unique_ptr<ctc_tc_plan> pl(new ctc_tc_plan());
    int64_t oa = 0, ob = 0, oc = 0;
    std::string e = ctc::shard_plan(plan->plan, mode, rank, nranks, pl->plan, oa, ob, oc);
    if (!e.empty()) return fail(CTC_TC_EINVAL, e);
    if (off_a) *off_a = oa;
    if (off_b) *off_b = ob;
    if (off_c) *off_c = oc;
    *shard = pl.release();
    return CTC_TC_OK;
    CTC_ABI_CATCH
}

ctc_tc_plan *ctc_tc_plan_gemm_plan(ctc_tc_plan *plan)
{
    try {
        if (!plan) {
            fail(CTC_TC_EINVAL, "null plan");
            return nullptr;
        }
        if (plan->plan.prefer_transposes) return ensure_packed_plan(plan) == CTC_TC_OK ? plan->packed.get() : nullptr;
        if (takes_padded_path(plan)) return ensure_padded_plan(plan) == CTC_TC_OK ? plan->padded.get() : nullptr;
        return plan;
    } catch (const std::exception &e) {
        fail(CTC_TC_ENOMEM, std::string("building the GEMM plan: ") + e.what());
        return nullptr;
    }
}

int ctc_tc_plan_prepare(ctc_tc_plan *plan)
{
    CTC_ABI_TRY
    if (!plan) return fail(CTC_TC_EINVAL, "null plan");
    int device = 0;
    int rc = current_device(device);
    if (rc) return rc;
    rc = scratch_setup(device);
    if (rc) return rc;
    DeviceTables t;
    rc = upload_tables(plan, device, t);
    if (rc) return rc;
    const ctc::Plan &p = plan->plan;
    size_t bytes = worst_splitk_bytes(p);
    if (p.prefer_transposes) {
        // everything the explicit-transpose execution touches: the packed plan with its own tables and
        // split-K workspace, the three permutations' tables, the workspace of the three packed tensors
        rc = ensure_packed_plan(plan);
        if (rc) return rc;
        rc = ctc_tc_plan_prepare(plan->packed.get());
        if (rc) return rc;
        TransposeShapes ts = transpose_shapes(p);
        PermuteTables unused;
        const int es = (int)ctc::dtype_size(p.dtype);
        rc = get_permute_tables(device, (int)ts.ea.size(), ts.ea.data(), p.perm_a.data(), es, unused);
        if (rc == CTC_TC_OK) rc = get_permute_tables(device, (int)ts.eb.size(), ts.eb.data(), p.perm_b.data(), es, unused);
        if (rc == CTC_TC_OK) rc = get_permute_tables(device, (int)ts.eg.size(), ts.eg.data(), p.perm_c.data(), es, unused);
        if (rc) return rc;
        const size_t ttgt = (size_t)ctc_tc_ttgt_workspace_bytes(plan);
        if (ttgt > bytes) bytes = ttgt;
    }
    if (takes_padded_path(plan)) {
        // everything the padded-copy execution touches: the GEMM plan on the copies with its tables and split-K
        // workspace, the two copies' tables, the workspace of the copies (cleared here)
        rc = ensure_padded_plan(plan);
        if (rc) return rc;
        rc = ctc_tc_plan_prepare(plan->padded.get());
        if (rc) return rc;
        PermuteTables ta, tb;
        rc = padded_permute_tables(plan, device, ta, tb);
        if (rc) return rc;
        char *wa = nullptr, *wb = nullptr;
        return padded_workspace(plan, device, nullptr, true, wa, wb);
    }
    if (bytes > 0) {
        std::lock_guard<std::mutex> lock(plan->mu);
        rc = grow_workspace(plan->tables[device], bytes);
    }
    return rc;
    CTC_ABI_CATCH
}

int ctc_tc_execute(ctc_tc_plan *plan, double alpha, const void *A, const void *B, double beta, void *C,
                   void *stream)
{
    CTC_ABI_TRY
    if (!plan || !A || !B || !C) return fail(CTC_TC_EINVAL, "null argument");
    static const bool no_auto = getenv("CTC_TC_NO_AUTO_TTGT") != nullptr;
    if (plan->plan.prefer_transposes && beta == 0.0 && !no_auto) {
        // Pathological strides (see Plan::prefer_transposes): transpose A and B, packed GEMM, transpose
        // into C, through a workspace the plan owns on this device.  Executions of one plan are
        // serialised on the host here and ordered on the device through the workspace's event.
        int device = 0;
        int rc = current_device(device);
        if (rc) return rc;
        DeviceTables t;
        rc = upload_tables(plan, device, t);
        if (rc) return rc;
        std::lock_guard<std::mutex> serial(plan->transposes_mu);
        hipStream_t s = static_cast<hipStream_t>(stream);
        const int64_t bytes = ctc_tc_ttgt_workspace_bytes(plan);
        void *ws = nullptr;
        {
            std::lock_guard<std::mutex> lock(plan->mu);
            DeviceTables &dt = plan->tables[device];
            if (dt.ws_used && dt.ws_stream != s) {
                hipError_t e = hipStreamWaitEvent(s, dt.ws_done, 0);
                if (e != hipSuccess) return fail_hip(e, "hipStreamWaitEvent");
            }
            rc = grow_workspace(dt, (size_t)bytes);
            if (rc) return rc;
            ws = dt.ws;
        }
        rc = ctc_tc_execute_ttgt(plan, alpha, A, B, 0.0, C, ws, bytes, stream);
        if (rc) return rc;
        std::lock_guard<std::mutex> lock(plan->mu);
        DeviceTables &dt = plan->tables[device];
        (void)hipEventRecord(dt.ws_done, s);
        dt.ws_stream = s;
        dt.ws_used = true;
        return CTC_TC_OK;
    }
    if (takes_padded_path(plan)) return execute_padded(plan, alpha, A, B, beta, C, static_cast<hipStream_t>(stream));
    return launch_plan(plan, alpha, A, B, beta, C, static_cast<hipStream_t>(stream));
    CTC_ABI_CATCH
}

int64_t ctc_tc_noalloc_scratch_bytes(void) { return (int64_t)kScratchBytes; }

int ctc_tc_execute_noalloc(ctc_tc_plan *plan, double alpha, const void *A, const void *B, double beta, void *C,
                           void *stream)
{
    CTC_ABI_TRY
    if (!plan || !A || !B || !C) return fail(CTC_TC_EINVAL, "null argument");
    int device = 0;
    int rc = current_device(device);
    if (rc) return rc;
    if (device < 0 || device >= kMaxDevices)
        return fail(CTC_TC_EUNSUPPORTED, "ctc_tc_execute_noalloc: device ordinal " + std::to_string(device) +
                                             " is beyond the " + std::to_string(kMaxDevices) + " devices the fixed scratch buffers cover");
    void *const scratch_dev = g_scratch_ptr[device].load();
    if (scratch_dev == nullptr)
        return fail(CTC_TC_EINVAL, "ctc_tc_execute_noalloc: call ctc_tc_plan_prepare on this device first "
                                   "(it resolves the scratch buffer's address and uploads the plan's tables)");
    // always the fused kernel (a plan that prefers explicit transposes would need three packed tensors
    // of workspace); split-K partial sums go to this device's fixed scratch buffer
    ctc_internal::Workspace ws;
    ws.fixed = scratch_dev;
    ws.fixed_bytes = kScratchBytes;
    return launch_plan(plan, alpha, A, B, beta, C, static_cast<hipStream_t>(stream), &ws, /*no_alloc=*/true);
    CTC_ABI_CATCH
}

int ctc_tc_permute(const void *in, void *out, int rank, const int64_t *extents_in, const int *perm,
                   int elem_size, void *stream)
{
    CTC_ABI_TRY
    if (!in || !out || !extents_in || !perm) return fail(CTC_TC_EINVAL, "null argument");
    if (in == out) return fail(CTC_TC_EINVAL, "in-place permutation is not supported");
    int device = 0;
    int rc = current_device(device);
    if (rc) return rc;
    PermuteTables t;
    if (elem_size != 2 && elem_size != 4 && elem_size != 8) return fail(CTC_TC_EINVAL, "element size must be 2, 4 or 8 bytes");
    rc = get_permute_tables(device, rank, extents_in, perm, elem_size, t);
    if (rc) return rc;
    ctc::PermuteParams p = t.params;
    p.in = in;
    p.out = out;
    hipError_t e = ctc::launch_permute(elem_size, p, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(e, "permute kernel launch");
    return CTC_TC_OK;
    CTC_ABI_CATCH
}

int64_t ctc_tc_ttgt_workspace_bytes(const ctc_tc_plan *plan)
{
    if (!plan) return fail(CTC_TC_EINVAL, "null plan");
    const ctc::Plan &p = plan->plan;
    const int64_t es = ctc::dtype_size(p.dtype);
    auto pad = [](int64_t b) { return (b + 255) & ~int64_t(255); };
    return pad(p.size_a * es) + pad(p.size_b * es) + pad(p.size_c * es);
}

int ctc_tc_execute_ttgt(ctc_tc_plan *plan, double alpha, const void *A, const void *B, double beta,
                        void *C, void *workspace, int64_t workspace_bytes, void *stream)
{
    CTC_ABI_TRY
    if (!plan || !A || !B || !C || !workspace) return fail(CTC_TC_EINVAL, "null argument");
    if (beta != 0.0) return fail(CTC_TC_EUNSUPPORTED, "the explicit-transpose path supports beta == 0 only");
    const ctc::Plan &p = plan->plan;
    for (size_t i = 0; i < p.strides_a.size(); ++i) {
        // the explicit transposes assume packed operands
        int64_t s = 1;
        for (size_t j = i + 1; j < p.a_modes.size(); ++j) s *= p.modes[p.a_modes[j]].extent;
        if (p.strides_a[i] != s) return fail(CTC_TC_EUNSUPPORTED, "explicit-transpose path needs packed operands");
    }
    for (size_t i = 0; i < p.strides_b.size(); ++i) {
        int64_t s = 1;
        for (size_t j = i + 1; j < p.b_modes.size(); ++j) s *= p.modes[p.b_modes[j]].extent;
        if (p.strides_b[i] != s) return fail(CTC_TC_EUNSUPPORTED, "explicit-transpose path needs packed operands");
    }
    for (size_t i = 0; i < p.strides_c.size(); ++i) {
        int64_t s = 1;
        for (size_t j = i + 1; j < p.c_modes.size(); ++j) s *= p.modes[p.c_modes[j]].extent;
        if (p.strides_c[i] != s) return fail(CTC_TC_EUNSUPPORTED, "explicit-transpose path needs packed operands");
    }
    if (workspace_bytes < ctc_tc_ttgt_workspace_bytes(plan)) return fail(CTC_TC_EINVAL, "workspace too small");
    const int es = ctc::dtype_size(p.dtype);
    auto pad = [](int64_t b) { return (b + 255) & ~int64_t(255); };
    char *wa = static_cast<char *>(workspace);
    char *wb = wa + pad(p.size_a * es);
    char *wc = wb + pad(p.size_b * es);

    int rc = ensure_packed_plan(plan);
    if (rc) return rc;
    TransposeShapes ts = transpose_shapes(p);
    rc = ctc_tc_permute(A, wa, (int)ts.ea.size(), ts.ea.data(), p.perm_a.data(), es, stream);
    if (rc) return rc;
    rc = ctc_tc_permute(B, wb, (int)ts.eb.size(), ts.eb.data(), p.perm_b.data(), es, stream);
    if (rc) return rc;
    rc = launch_plan(plan->packed.get(), alpha, wa, wb, 0.0, wc, static_cast<hipStream_t>(stream));
    if (rc) return rc;
    // [batch,M,N] -> C
    return ctc_tc_permute(wc, C, (int)ts.eg.size(), ts.eg.data(), p.perm_c.data(), es, stream);
    CTC_ABI_CATCH
}

int ctc_tc_grouped_contract(const ctc_tc_small_gemm *problems, int count, int dtype, double alpha, const void *A,
                            const void *B, double beta, void *C, void *stream)
{
    static_assert(sizeof(ctc_tc_small_gemm) == sizeof(ctc::SmallGemm), "descriptor layouts must agree");
    if (count < 0 || (count > 0 && (!problems || !A || !B || !C))) return fail(CTC_TC_EINVAL, "bad argument");
    if (dtype != CTC_TC_F64 && dtype != CTC_TC_F32) return fail(CTC_TC_EUNSUPPORTED, "grouped contractions: f64 or f32");
    hipError_t e = ctc::launch_grouped(dtype, reinterpret_cast<const ctc::SmallGemm *>(problems), count, A, B, C,
                                       alpha, beta, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(e, "grouped contraction kernel launch");
    return CTC_TC_OK;
}

int ctc_tc_fill(void *dev, int64_t n, uint64_t seed, int64_t first, int dtype, void *stream)
{
    if (!dev || n < 0) return fail(CTC_TC_EINVAL, "bad argument");
    hipError_t e = ctc::launch_fill(dev, n, seed, first, dtype, static_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(e, "fill kernel launch");
    return CTC_TC_OK;
}

int ctc_tc_time_execute(ctc_tc_plan *plan, double alpha, const void *A, const void *B, double beta,
                        void *C, void *stream, int warmup, int iters, double *avg_ms)
{
    CTC_ABI_TRY
    if (!plan || !avg_ms || iters <= 0 || warmup < 0) return fail(CTC_TC_EINVAL, "bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int i = 0; i < warmup; ++i) {
        int rc = ctc_tc_execute(plan, alpha, A, B, beta, C, stream);
        if (rc) return rc;
    }
    hipEvent_t e0, e1;
    hipError_t e = hipEventCreate(&e0);
    if (e != hipSuccess) return fail_hip(e, "hipEventCreate");
    e = hipEventCreate(&e1);
    if (e != hipSuccess) { (void)hipEventDestroy(e0); return fail_hip(e, "hipEventCreate"); }
    (void)hipEventRecord(e0, s);
    int rc = CTC_TC_OK;
    for (int i = 0; i < iters && rc == CTC_TC_OK; ++i) rc = ctc_tc_execute(plan, alpha, A, B, beta, C, stream);
    (void)hipEventRecord(e1, s);
    e = hipEventSynchronize(e1);
    float ms = 0.f;
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (rc) return rc;
    if (e != hipSuccess) return fail_hip(e, "event timing");
    *avg_ms = (double)ms / iters;
    return CTC_TC_OK;
    CTC_ABI_CATCH
}

int ctc_tc_time_execute_each(ctc_tc_plan *plan, double alpha, const void *A, const void *B, double beta,
                             void *C, void *stream, int iters, double *ms_each)
{
    CTC_ABI_TRY
    if (!plan || !ms_each || iters <= 0 || iters > 4096) return fail(CTC_TC_EINVAL, "bad argument");
    hipStream_t s = static_cast<hipStream_t>(stream);
    std::vector<hipEvent_t> ev((size_t)iters + 1, nullptr);
    hipError_t e = hipSuccess;
    for (auto &x : ev)
        if (e == hipSuccess) e = hipEventCreate(&x);
    int rc = CTC_TC_OK;
    if (e == hipSuccess) {
        (void)hipEventRecord(ev[0], s);
        for (int i = 0; i < iters && rc == CTC_TC_OK; ++i) {
            rc = ctc_tc_execute(plan, alpha, A, B, beta, C, stream);
            (void)hipEventRecord(ev[(size_t)i + 1], s);
        }
        e = hipEventSynchronize(ev[(size_t)iters]);
        for (int i = 0; i < iters && e == hipSuccess && rc == CTC_TC_OK; ++i) {
            float ms = 0.f;
            e = hipEventElapsedTime(&ms, ev[(size_t)i], ev[(size_t)i + 1]);
            ms_each[i] = (double)ms;
        }
    }
    for (auto x : ev)
        if (x) (void)hipEventDestroy(x);
    if (rc) return rc;
    if (e != hipSuccess) return fail_hip(e, "event timing");
    return CTC_TC_OK;
    CTC_ABI_CATCH
}

}  // extern "C"
