// Persistent MFMA GEMM for contractions with a short K and a large C (e.g. abcijk = adij,bcdk with
// d = 64: 128 flops per byte of C), where the classic one-tile-per-workgroup kernel spends as long
// storing a C tile as computing it and the two do not overlap.
//
// One workgroup per CU (4 waves, one per SIMD, up to 512 registers each) walks tiles (items of
// 128 x 64 in f64, whose two accumulator sets then take 128 registers, and 128 x 128 in f32; K-tiles
// of 32: with one wave per SIMD a K-tile has to be long enough to cover its successor's loads)
// w, w + G, w + 2G, ... (G = grid size) as ONE stream of K-tiles:
//   - two accumulator sets per wave: while the MFMAs of item X fill one set, the 16-byte stores of
//     item X-1 leave the other, one store every few MFMAs (a burst of stores at the end of a tile
//     stalls the matrix pipe chip-wide; the same stores dealt out behind MFMAs do not);
//   - the operand pipeline (global -> registers -> LDS, two LDS stages) runs across item
//     boundaries, so an item has no prologue: the first K-tiles of item X+1 are loaded while the last
//     ones of item X are multiplied;
//   - the element offsets an item needs (its A rows, B columns, C rows and columns) are staged in a
//     four-slot LDS ring two items ahead.
// Same operand staging, LDS images, fragment pattern and block order as gemm_kernels.hip.
// Restrictions (the caller falls back to the classic kernel otherwise): vector layouts on both
// operands, 16-byte stores legal in C, beta == 0, batch == 1, at least kStreamMinKTiles K-tiles per item.
#include "gemm_device.hpp"
#include "kernels.hpp"

#include <type_traits>

namespace ctc {

namespace {

using namespace gemm_detail;

constexpr int kStreamMinKTiles = 2;
constexpr int kStreamBK = 32;
template <typename T> constexpr int stream_bn() { return sizeof(T) == 8 ? 64 : 128; }

// f(integral_constant<int, I>) for I = 0 .. N-1
template <int I, int N, typename F> __device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>());
        static_for<I + 1, N>(f);
    }
}

template <typename T, int LA, int LB>
__global__ void __launch_bounds__(256, 1) gemm_stream_kernel(const GemmParams p)
{
    constexpr int BM = 128, BN = stream_bn<T>(), BK = kStreamBK, WM = 2, WN = 2, NT = 256;
    constexpr int TM = BM / WM / 16, TN = BN / WN / 16, NMFMA = TM * TN, KSTEPS = BK / 4;
    typedef Traits<T> Tr;
    typedef typename Tr::acc_t acc_t;
    typedef typename Tr::vec_t vec_t;
    typedef Stager<T, LA, BM, BK, NT> SA;
    typedef Stager<T, LB, BN, BK, NT> SB;
    constexpr int STAGE_ELEMS = SA::LDS_ELEMS + SB::LDS_ELEMS;
    constexpr int64_t kNoOffset = INT64_MIN;
    // stores of one item per wave, dealt out over the item's first two K-tiles
    constexpr int NSTORE = TM * Tr::NCV * TN;
    constexpr int STORE_TILES = kStreamMinKTiles;
    constexpr int SPT = NSTORE / STORE_TILES;           // stores per tile
    constexpr int SPACING = NMFMA * KSTEPS / SPT;       // MFMAs between two stores
    static_assert(NSTORE % STORE_TILES == 0 && (NMFMA * KSTEPS) % SPT == 0 && SPACING >= 4, "store schedule");
    static_assert(KSTEPS % 2 == 0, "fragment slots alternate per k-step");

    __shared__ __attribute__((aligned(16))) T smem[2 * STAGE_ELEMS];
    // element offsets of an item: [0,BM) A rows, [BM,BM+BN) B columns (both clamped into the operand),
    // then C rows and C columns in the same order (kNoOffset outside C); slot = item mod 4
    constexpr int NOFF = BM + BN;
    static_assert(NOFF <= NT, "one thread per offset");
    __shared__ int64_t s_off[4][2 * NOFF];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm0 = (wave / WN) * (TM * 16);
    const int wn0 = (wave % WN) * (TN * 16);
    const int frow = lane & 15;
    const int fk = lane >> 4;

    // Workgroup -> stream of tiles.  Blocks b and b + 8 share an XCD: give each XCD a contiguous run
    // of slots, so that the tiles its workgroups hold at one time are neighbours in the grouped order
    // (shared A and B panels in that XCD's L2).
    const int G = (int)gridDim.x;
    const int bid = (int)blockIdx.x;
    const int wslot = (G & 7) == 0 ? (bid & 7) * (G >> 3) + (bid >> 3) : bid;
    const int64_t tiles = (int64_t)p.nbm * p.nbn;
    if (wslot >= tiles) return;
    const int n_items = (int)((tiles - wslot + G - 1) / G);
    const int width = p.group * p.nbn;
    auto item_origin = [&](int it, int64_t &m0, int64_t &n0) {
        const int sid = (int)((int64_t)it * G + wslot);
        const int first_m = (sid / width) * p.group;
        const int gsz = min(p.nbm - first_m, p.group);
        m0 = (int64_t)(first_m + (sid % width) % gsz) * BM;
        n0 = (int64_t)((sid % width) / gsz) * BN;
    };

    const T *A = static_cast<const T *>(p.A) + p.bat_a[0];
    const T *B = static_cast<const T *>(p.B) + p.bat_b[0];
    T *C = static_cast<T *>(p.C) + p.bat_c[0];
    const T alpha = (T)p.alpha;
    const int nk = (int)((p.K + BK - 1) / BK);

    // offsets of one item: fetched into two registers per thread, written to the ring a K-tile later
    int64_t ofs_reg[2];
    auto off_fetch = [&](int it) {
        int64_t m0, n0;
        item_origin(it, m0, n0);
        const bool is_row = t < BM;
        const int64_t g = is_row ? m0 + t : n0 + (t - BM);
        const int64_t lim = is_row ? p.M : p.N;
        const bool ok = g < lim;
        const int64_t clamp_to = lim - (is_row ? SA::ROW_VEC : SB::ROW_VEC);
        ofs_reg[0] = (is_row ? p.row_a : p.col_b)[ok ? g : clamp_to];
        const int64_t oc = (is_row ? p.row_c : p.col_c)[ok ? g : 0];
        ofs_reg[1] = ok ? oc : kNoOffset;
    };
    auto off_commit = [&](int slot) {
        if (t < NOFF) {
            s_off[slot][t] = ofs_reg[0];
            s_off[slot][NOFF + t] = ofs_reg[1];
        }
    };

    SA sa;
    SB sb;
    sa.init_thread(t);
    sb.init_thread(t);

    acc_t acc[2][TM][TN];

    // ---- prologue: ring slots of "item -1" (nothing to store), item 0 and item 1, K-tile 0 into LDS
    // stage 0, K-tile 1 on its way in registers, first fragments of K-tile 0 in registers
    if (t < NOFF) s_off[3][NOFF + t] = kNoOffset;
    off_fetch(0);
    off_commit(0);
    if (n_items > 1) {
        off_fetch(1);
        off_commit(1);
    }
    __syncthreads();
    sa.load_fix(&s_off[0][0]);
    sb.load_fix(&s_off[0][BM]);
    sa.fetch_step(p.k_a, 0);
    sb.fetch_step(p.k_b, 0);
#pragma unroll
    for (int i = 0; i < SA::NV; ++i) sa.issue_one(i, A);
#pragma unroll
    for (int i = 0; i < SB::NV; ++i) sb.issue_one(i, B);
    sa.fetch_step(p.k_a, BK);
    sb.fetch_step(p.k_b, BK);
#pragma unroll
    for (int i = 0; i < SA::NV; ++i) sa.template commit_one<true>(i, smem, 0, p.K);
#pragma unroll
    for (int i = 0; i < SB::NV; ++i) sb.template commit_one<true>(i, smem + SA::LDS_ELEMS, 0, p.K);
#pragma unroll
    for (int i = 0; i < SA::NV; ++i) sa.issue_one(i, A);
#pragma unroll
    for (int i = 0; i < SB::NV; ++i) sb.issue_one(i, B);
    __syncthreads();
    // stream tile 2 is K-tile 2 of item 0, or K-tile 0 of item 1 when an item has two K-tiles
    sa.fetch_step(p.k_a, nk > 2 ? 2 * (int64_t)BK : 0);
    sb.fetch_step(p.k_b, nk > 2 ? 2 * (int64_t)BK : 0);

    T fa[2][TM], fb[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[0][i] = SA::frag(smem, wm0 + 16 * i + frow, fk);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[0][j] = SB::frag(smem + SA::LDS_ELEMS, wn0 + 16 * j + frow, fk);

    int buf = 0;  // LDS stage of the K-tile being multiplied

    // One 16-byte store of the finished item (accumulator set SET), index S in [0, NSTORE).
    // rvoff is carried from the first store of a row group to the other TN - 1.
    int64_t cvoff[TN];
    int64_t rvoff = kNoOffset;
    vec_t sval;
    auto store_prepare = [&](auto set_t, auto s_t, const int64_t *slot) {
        constexpr int SET = decltype(set_t)::value, S = decltype(s_t)::value;
        constexpr int i = S / (Tr::NCV * TN), sub = (S / TN) % Tr::NCV, j = S % TN;
        if (j == 0) rvoff = slot[NOFF + wm0 + 16 * i + Tr::cvec_row(lane, sub)];
        sval = Tr::cvec(acc[SET][i][j], lane, sub) * alpha;
    };
    auto store_issue = [&](auto s_t) {
        constexpr int S = decltype(s_t)::value;
        constexpr int j = S % TN;
        if (rvoff != kNoOffset && cvoff[j] != kNoOffset) {
            vec_t *dst = reinterpret_cast<vec_t *>(C + rvoff + cvoff[j]);
            if (p.c_stream) __builtin_nontemporal_store(sval, dst);
            else *dst = sval;
        }
    };
    auto load_cvoff = [&](const int64_t *slot) {
#pragma unroll
        for (int j = 0; j < TN; ++j) cvoff[j] = slot[NOFF + BM + wn0 + 16 * j + (frow & ~(Tr::CV - 1))];
    };

    // One K-tile of the stream: K-tile kt of item X into accumulator set SET.
    //   VAR 0, 1: K-tile kt == VAR of the item, carrying stores [VAR * SPT, (VAR + 1) * SPT) of item X-1
    //             (set SET ^ 1); VAR 0 also starts the accumulators from zero;
    //   VAR 2:    any later K-tile, no stores.
    // Every other instruction is dealt out behind the MFMAs in a pinned order, as in the classic
    // kernel: fragment reads of the next k-step, the LDS writes of stream tile +1 in k-step KSTEPS-2,
    // the barrier, then first fragments of tile +1, global loads of tile +2 and k offsets of tile +3.
    auto tile_step = [&](auto set_t, auto var_t, int X, int kt) {
        constexpr int SET = decltype(set_t)::value, VAR = decltype(var_t)::value;
        const T *la = smem + buf * STAGE_ELEMS;
        const T *lb = la + SA::LDS_ELEMS;
        T *na = smem + (buf ^ 1) * STAGE_ELEMS;
        T *nb = na + SA::LDS_ELEMS;
        buf ^= 1;
        const bool has_next = X + 1 < n_items, has_next2 = X + 2 < n_items;
        int kt1 = kt + 1, kt3 = kt + 3;
        if (kt1 >= nk) kt1 -= nk;
        if (kt3 >= nk) kt3 -= nk;
        if (kt3 >= nk) kt3 -= nk;  // nk >= 2, so kt + 3 <= 2 nk + 1 wraps at most twice
        const int64_t k_next = (int64_t)kt1 * BK;
        // the loads issued in this tile belong to stream tile +2; when that is the first K-tile of the
        // next item, the stagers switch to the next item's row offsets first
        const bool loader_switch = (kt == nk - 2) && has_next;
        const int64_t *slot_prev = &s_off[(X + 3) & 3][0];  // item X-1
        const int64_t *slot_next = &s_off[(X + 1) & 3][0];
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            const int s = kk & 1;
            const bool last = (kk == KSTEPS - 1);
#ifndef CTC_SX_NOBARRIER
            if (last) __syncthreads();
#endif
            const T *ra = last ? na : la;
            const T *rb = last ? nb : lb;
            const int kq = last ? fk : (kk + 1) * 4 + fk;
            constexpr int NREAD = TM + TN;
            constexpr int NUNIT = SA::NV + SB::NV;
#pragma unroll
            for (int q = 0; q < NMFMA; ++q) {
                if (VAR == 0 && kk == 0)
                    acc[SET][q / TN][q % TN] = Tr::mfma(fa[s][q / TN], fb[s][q % TN], zero_of<acc_t>());
                else
                    acc[SET][q / TN][q % TN] = Tr::mfma(fa[s][q / TN], fb[s][q % TN], acc[SET][q / TN][q % TN]);
                // fragment reads: the first NREAD / 2 MFMAs each carry two
#pragma unroll
                for (int r = 0; r < NREAD; ++r) {
                    if (r / 2 == q) {
                        if (r < TM) fa[s ^ 1][r] = SA::frag(ra, wm0 + 16 * r + frow, kq);
                        else fb[s ^ 1][r - TM] = SB::frag(rb, wn0 + 16 * (r - TM) + frow, kq);
                    }
                }
                {
                    // staging units: the LDS writes of stream tile +1 spread over the k-steps in front of
                    // the barrier (three of them when a K-tile has eight: twelve 1 KB writes per wave
                    // behind five MFMAs would queue up in the LDS), the global loads of tile +2 over
                    // the last k-step, behind its fragment reads
                    constexpr int COMMIT_SPAN = (KSTEPS >= 8 ? 3 : 1) * NMFMA;
                    constexpr int C0 = (KSTEPS - 1) * NMFMA - COMMIT_SPAN;
                    constexpr int Q0 = NREAD / 2;
                    constexpr int QN = NMFMA - Q0;
                    const int idx = kk * NMFMA + q;
#pragma unroll
                    for (int u = 0; u < NUNIT; ++u) {
#ifndef CTC_SX_NOCOMMIT
                        if (idx == C0 + u * COMMIT_SPAN / NUNIT) {
                            if (u < SA::NV) sa.template commit_one<true>(u, na, k_next, p.K);
                            else sb.template commit_one<true>(u - SA::NV, nb, k_next, p.K);
                        }
#endif
#ifndef CTC_SX_NOLOAD
                        if (last && Q0 + u * QN / NUNIT == q) {
                            if (u < SA::NV) sa.issue_one(u, A);
                            else sb.issue_one(u - SA::NV, B);
                        }
#endif
                    }
                }
                if (last && q == NMFMA - 1) {
                    sa.fetch_step(p.k_a, (int64_t)kt3 * BK);
                    sb.fetch_step(p.k_b, (int64_t)kt3 * BK);
                }
                // item bookkeeping, a few instructions each, away from the staging units
                if (VAR == 0 && kk == 0 && q == 4) load_cvoff(slot_prev);
                if (VAR == 0 && kk == 0 && q == NMFMA - 3 && has_next2) off_fetch(X + 2);
                if (VAR == 1 && kk == 0 && q == NMFMA - 3 && has_next2) off_commit((X + 2) & 3);
                if (kk == 0 && q == NMFMA - 1 && loader_switch) {
                    sa.load_fix(slot_next);
                    sb.load_fix(slot_next + BM);
                }
                // stores of the previous item: value one MFMA ahead of the store itself
                if (VAR < STORE_TILES) {
                    const int idx = kk * NMFMA + q;
                    static_for<0, SPT>([&](auto u_t) {
                        constexpr int u = decltype(u_t)::value;
                        constexpr int S = (VAR < STORE_TILES ? VAR : 0) * SPT + u;
                        if (idx == u * SPACING + SPACING / 2 + 1)
                            store_prepare(std::integral_constant<int, SET ^ 1>(), std::integral_constant<int, S>(), slot_prev);
                        if (idx == u * SPACING + SPACING / 2 + 2) store_issue(std::integral_constant<int, S>());
                    });
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    auto run_item = [&](auto set_t, int X) {
        static_assert(STORE_TILES == 2, "one call per store-carrying K-tile below");
        tile_step(set_t, std::integral_constant<int, 0>(), X, 0);
        tile_step(set_t, std::integral_constant<int, 1>(), X, 1);
        for (int kt = STORE_TILES; kt < nk; ++kt) tile_step(set_t, std::integral_constant<int, 2>(), X, kt);
    };

    for (int X = 0; X < n_items; X += 2) {
        run_item(std::integral_constant<int, 0>(), X);
        if (X + 1 < n_items) run_item(std::integral_constant<int, 1>(), X + 1);
    }

    // the last item's stores: nothing left to hide them behind
    auto flush = [&](auto set_t, const int64_t *slot) {
        constexpr int SET = decltype(set_t)::value;
        load_cvoff(slot);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int sub = 0; sub < Tr::NCV; ++sub) {
                const int64_t ro = slot[NOFF + wm0 + 16 * i + Tr::cvec_row(lane, sub)];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const vec_t v = Tr::cvec(acc[SET][i][j], lane, sub) * alpha;
                    if (ro != kNoOffset && cvoff[j] != kNoOffset) {
                        vec_t *dst = reinterpret_cast<vec_t *>(C + ro + cvoff[j]);
                        if (p.c_stream) __builtin_nontemporal_store(v, dst);
                        else *dst = v;
                    }
                }
            }
    };
    const int64_t *slot_last = &s_off[(n_items - 1) & 3][0];
    if ((n_items - 1) & 1) flush(std::integral_constant<int, 1>(), slot_last);
    else flush(std::integral_constant<int, 0>(), slot_last);
}

template <typename T, int LA, int LB> hipError_t launch_stream_one(const GemmParams &p, hipStream_t stream)
{
    static_assert(3 * kStreamBK <= kKTablePad, "the prologue reads the k tables three K-tiles ahead");
    constexpr int BN = stream_bn<T>();
    int dev = 0, cus = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    GemmParams q = p;
    q.nbm = (int)((p.M + 127) / 128);
    q.nbn = (int)((p.N + BN - 1) / BN);
    if (q.group < 1) q.group = 1;
    const int64_t tiles = (int64_t)q.nbm * q.nbn;
    if (tiles > 0x7FFFFFFF) return hipErrorInvalidValue;
    const int grid = (int)(tiles < cus ? tiles : cus);
    hipLaunchKernelGGL((gemm_stream_kernel<T, LA, LB>), dim3((unsigned)grid), dim3(256), 0, stream, q);
    return hipGetLastError();
}

template <typename T> hipError_t launch_stream_layouts(int la, int lb, const GemmParams &p, hipStream_t stream)
{
    switch (la * 2 + lb) {
#ifndef CTC_SX_ONE  // diagnostic builds of the outer/outer instantiation only (tools/build_variant.sh)
    case 0: return launch_stream_one<T, 0, 0>(p, stream);
    case 1: return launch_stream_one<T, 0, 1>(p, stream);
    case 2: return launch_stream_one<T, 1, 0>(p, stream);
#endif
    case 3: return launch_stream_one<T, 1, 1>(p, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace

bool stream_gemm_supported(int dtype, int layout_a, int layout_b, int bm, int bn, const GemmParams &p)
{
    if (dtype != 0 && dtype != 1) return false;
    return bm == 128 && bn == 128 && (layout_a == 0 || layout_a == 1) && (layout_b == 0 || layout_b == 1) &&
           p.batch == 1 && p.beta == 0.0 && p.c_vec > 1 && (p.K + kStreamBK - 1) / kStreamBK >= kStreamMinKTiles;
}

hipError_t launch_gemm_stream(int dtype, int layout_a, int layout_b, const GemmParams &p, hipStream_t stream)
{
    if (!stream_gemm_supported(dtype, layout_a, layout_b, 128, 128, p)) return hipErrorInvalidValue;
    if (dtype == 0) return launch_stream_layouts<double>(layout_a, layout_b, p, stream);
#ifndef CTC_SX_ONE
    return launch_stream_layouts<float>(layout_a, layout_b, p, stream);
#else
    return hipErrorInvalidValue;
#endif
}

}  // namespace ctc
