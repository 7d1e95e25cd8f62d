// MFMA GEMM for tensor contractions on gfx950, with the mode permutation fused
// into the operand loads.
//
// The planner hands over element-offset tables (row/k/col per operand), so the
// kernel never sees tensor modes: A(m,k) lives at A[row_a[m] + k_a[k]].  Each
// operand is staged through registers into a double-buffered LDS tile whose
// shape depends on which GEMM dimension is unit-stride in memory:
//   K-major    (unit stride along K):      LDS image [row][BK + pad], loads are
//                                          16-byte vectors along K;
//   outer-major(unit stride along M or N): LDS image [k][rows + 16], loads are
//                                          16-byte vectors along the rows;
//   scalar     (no usable unit stride):    K-major image, element loads.
// Both images are read back one element per lane in the MFMA 16x16x4 operand
// pattern (lane l: row l&15, k l>>4) without bank conflicts for 8-byte elements:
// the K-major row stride of BK+2 doubles walks the 64 dword banks in steps of 36,
// the outer-major stride of rows+16 doubles offsets consecutive k by 32 banks.
//
// One workgroup = WM x WN waves; each wave owns a (BM/WM) x (BN/WN) block of C as
// TM x TN accumulators of v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32.
// Per K-tile the wave runs its MFMAs back to back and deals every other instruction
// (fragment reads, LDS writes of the next tile, global loads of the tile after it)
// out behind them, one barrier per tile; see tile_step below.
#include "gemm_device.hpp"
#include "kernels.hpp"

#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

// K-tile depth and the shape of the large tile; overridable for diagnostic builds
// (tools/build_variant.sh), e.g. -DCTC_GEMM_BIG_BN=64 -DCTC_GEMM_BK=8.
#ifndef CTC_GEMM_BK
#define CTC_GEMM_BK 16
#endif
#ifndef CTC_GEMM_BIG_BM
#define CTC_GEMM_BIG_BM 128
#endif
#ifndef CTC_GEMM_BIG_BN
#define CTC_GEMM_BIG_BN 128
#endif

namespace ctc {

namespace {

// Diagnostic build (-DCTC_STAMPS): wave 0 of every block sums, per K-tile segment, the
// shader-clock ticks it spends there and the launcher prints the median shares.
// Never defined in the shipped library.
#ifdef CTC_STAMPS
__device__ unsigned long long g_stamp_sums[8 * 8192];
#define CTC_STAMP(var)                                                                      \
    do {                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");         \
        __builtin_amdgcn_sched_barrier(0);                                                  \
    } while (0)
#else
#define CTC_STAMP(var) do { } while (0)
#endif

using namespace gemm_detail;

template <typename T, int LA, int LB, int BM, int BN, int BK, int WM, int WN>
__global__ void __launch_bounds__(64 * WM * WN, 2) gemm_kernel(const GemmParams p)
{
    constexpr int NT = 64 * WM * WN;
    constexpr int TM = BM / WM / 16;
    constexpr int TN = BN / WN / 16;
    constexpr int NMFMA = TM * TN;  // MFMAs per k-step
    typedef Traits<T> Tr;
    typedef typename Tr::acc_t acc_t;
    typedef Stager<T, LA, BM, BK, NT> SA;
    typedef Stager<T, LB, BN, BK, NT> SB;

    __shared__ __attribute__((aligned(16))) T smem[2 * (SA::LDS_ELEMS + SB::LDS_ELEMS)];
    // element offsets of the tile's C rows and columns (kNoOffset: outside C), fetched in the
    // prologue so that the epilogue issues its stores without waiting on any global load
    __shared__ int64_t s_coff[BM + BN];
    constexpr int STAGE_ELEMS = SA::LDS_ELEMS + SB::LDS_ELEMS;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
#ifdef CTC_STAMPS
    unsigned long long st_entry, st_loop0, st_loop1;
    const unsigned long long st_real0 = __builtin_amdgcn_s_memrealtime();
    CTC_STAMP(st_entry);
#endif
    const int wm0 = (wave / WN) * (TM * 16);
    const int wn0 = (wave % WN) * (TN * 16);
#ifdef CTC_CLOCK  // diagnostic build: shader clock while this workgroup runs = core-clock ticks / 100 MHz ticks
    const unsigned long long ck_core0 = __builtin_readcyclecounter(), ck_real0 = __builtin_amdgcn_s_memrealtime();
#endif

    // Block -> tile.  Blocks b and b+8 share an XCD (and its L2): give each XCD a
    // contiguous run of tile ids, then walk tiles in groups of `group` tile rows so
    // that concurrently running blocks share A and B panels.
    // The first dp_tiles tiles run whole (one workgroup each).  The remaining tiles - the
    // last, partly filled round of the grid - are cut along K into `split` chunks so that
    // they occupy every workgroup slot for 1/split of a tile's time; a chunk's raw
    // accumulators go to a workspace and splitk_fixup_kernel sums them in chunk order.
    int sid, chunk = 0;
    bool partial = false;
    {
        const int bid = blockIdx.x;
        if (bid < p.dp_tiles) {
            const int nblk = p.dp_tiles;
            const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
            sid = p.plain_order ? bid : (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        } else {
            const int item = bid - p.dp_tiles;
            sid = p.dp_tiles + item / p.split;
            chunk = item % p.split;
            partial = true;
        }
    }
    // a launch without a batch dimension in its grid: the tile id counts through the batch entries (GemmParams::flat_batch)
    int bz = blockIdx.y;
    const int list_id = sid;  // position in the launch's list of tiles: what a chunk's workspace slab is counted from
    if (p.flat_batch > 0) {
        bz = sid / (p.nbm * p.nbn);
        sid -= bz * (p.nbm * p.nbn);
    }
    const int width = p.group * p.nbn;
    const int first_m = (sid / width) * p.group;
    const int gsz = min(p.nbm - first_m, p.group);
    const int tm = first_m + (sid % width) % gsz;
    const int tn = (sid % width) / gsz;
    const int64_t m0 = (int64_t)tm * BM;
    const int64_t n0 = (int64_t)tn * BN;

    const T *A = static_cast<const T *>(p.A) + p.bat_a[bz];
    const T *B = static_cast<const T *>(p.B) + p.bat_b[bz];
    T *C = static_cast<T *>(p.C) + p.bat_c[bz];

    SA sa;
    SB sb;
    sa.init(p.row_a, m0, p.M, t);
    sb.init(p.col_b, n0, p.N, t);

    acc_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = zero_of<acc_t>();

    const int nk_total = (int)((p.K + BK - 1) / BK);
    const int kt_first = partial ? (int)((int64_t)chunk * nk_total / p.split) : 0;
    const int kt_last = partial ? (int)((int64_t)(chunk + 1) * nk_total / p.split) : nk_total;
#ifdef CTC_EXP_NOLOOP
    const int nk = 1;
#else
    const int nk = kt_last - kt_first;
#endif
    const int64_t k_base = (int64_t)kt_first * BK;
    constexpr int KSTEPS = BK / 4;
    static_assert(KSTEPS % 2 == 0, "fragment slots alternate per k-step");
    const int frow = lane & 15;
    const int fk = lane >> 4;

    // Prologue: tile 0 into LDS buffer 0, tile 1 on its way in registers, first
    // fragments of tile 0 in registers.
    sa.fetch_step(p.k_a, k_base);
    sb.fetch_step(p.k_b, k_base);
#pragma unroll
    for (int i = 0; i < SA::NV; ++i) sa.issue_one(i, A);
#pragma unroll
    for (int i = 0; i < SB::NV; ++i) sb.issue_one(i, B);
    sa.fetch_step(p.k_a, k_base + BK);
    sb.fetch_step(p.k_b, k_base + BK);
    if (!partial) {
        for (int idx = t; idx < BM + BN; idx += NT) {
            const bool is_row = idx < BM;
            const int64_t g = is_row ? m0 + idx : n0 + (idx - BM);
            const bool ok = g < (is_row ? p.M : p.N);
            const int64_t off = (is_row ? p.row_c : p.col_c)[ok ? g : 0];
            s_coff[idx] = ok ? off : kNoOffset;
        }
    }
#pragma unroll
    for (int i = 0; i < SA::NV; ++i) sa.template commit_one<true>(i, smem, k_base, p.K);
#pragma unroll
    for (int i = 0; i < SB::NV; ++i) sb.template commit_one<true>(i, smem + SA::LDS_ELEMS, k_base, p.K);
    // tile 1 leaves for the registers before the barrier: its latency overlaps the wait
#pragma unroll
    for (int i = 0; i < SA::NV; ++i) sa.issue_one(i, A);
#pragma unroll
    for (int i = 0; i < SB::NV; ++i) sb.issue_one(i, B);
    __syncthreads();
    sa.fetch_step(p.k_a, k_base + 2 * (int64_t)BK);
    sb.fetch_step(p.k_b, k_base + 2 * (int64_t)BK);

    T fa[2][TM], fb[2][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i) fa[0][i] = SA::frag(smem, wm0 + 16 * i + frow, fk);
#pragma unroll
    for (int j = 0; j < TN; ++j) fb[0][j] = SB::frag(smem + SA::LDS_ELEMS, wn0 + 16 * j + frow, fk);

    // One K-tile.  A v_mfma_f64_16x16x4_f64 occupies the matrix pipe for 64 cycles, so
    // every other instruction of the tile (LDS reads and writes, global loads, address
    // arithmetic) is dealt out one or two at a time BEHIND an MFMA, in program order
    // pinned with sched_barrier, instead of in clusters that would let the pipe drain:
    //   k-steps 0..KSTEPS-2 : fragment reads of the next k-step;
    //   k-step  KSTEPS-2    : also the LDS writes of tile kt+1 (registers -> other buffer);
    //   barrier, then
    //   k-step  KSTEPS-1    : first fragment reads of tile kt+1, global loads of tile
    //                         kt+2 (a whole tile of MFMAs to land), k offsets of kt+3.
#ifdef CTC_STAMPS
    unsigned long long st_a = 0, st_b = 0, st_c = 0, st_d = 0, st_n = 0;
#endif
    // Three instantiations, run in a straight line (loop, second-to-last tile, last tile):
    // the last two tiles leave out what would reach past the end of K, so that a contraction
    // with few K-tiles does not fetch two more tiles than it uses.
    auto tile_step = [&](auto mask_t, auto next_t, auto issue_t, int kt) {
        constexpr bool HAS1 = decltype(next_t)::value;   // there is a tile kt+1: commit it, read its first fragments
        constexpr bool HAS2 = decltype(issue_t)::value;  // there is a tile kt+2: issue its global loads
        constexpr bool MASK = decltype(mask_t)::value;   // tile kt+1 may cross K
        const int cur = kt & 1;
        const T *la = smem + cur * STAGE_ELEMS;
        const T *lb = la + SA::LDS_ELEMS;
        T *na = smem + (cur ^ 1) * STAGE_ELEMS;
        T *nb = na + SA::LDS_ELEMS;
        const int64_t k_next = k_base + (int64_t)(kt + 1) * BK;
#ifdef CTC_STAMPS
        unsigned long long t0, t1, t2, t3;
        CTC_STAMP(t0);
#endif
#pragma unroll
        for (int kk = 0; kk < KSTEPS; ++kk) {
            const int s = kk & 1;
            const bool last = (kk == KSTEPS - 1);
#ifdef CTC_STAMPS
            if (kk == KSTEPS - 2) CTC_STAMP(t1);
            if (last) CTC_STAMP(t2);
#endif
            if (last) __syncthreads();
#ifdef CTC_STAMPS
            if (last) CTC_STAMP(t3);
#endif
            const T *ra = last ? na : la;
            const T *rb = last ? nb : lb;
            const int kq = last ? fk : (kk + 1) * 4 + fk;
            const bool do_reads = !last || HAS1;
            const bool do_commit = (kk == KSTEPS - 2) && HAS1;
            const bool do_issue = last && HAS2;
            constexpr int NREAD = TM + TN;
            constexpr int NUNIT = SA::NV + SB::NV;
#pragma unroll
            for (int q = 0; q < NMFMA; ++q) {
                acc[q / TN][q % TN] = Tr::mfma(fa[s][q / TN], fb[s][q % TN], acc[q / TN][q % TN]);
                if (do_reads) {
                    // fragment reads: the first NREAD/2 MFMAs each carry two
#pragma unroll
                    for (int r = 0; r < NREAD; ++r) {
                        if (r / 2 == q) {
                            if (r < TM) fa[s ^ 1][r] = SA::frag(ra, wm0 + 16 * r + frow, kq);
                            else fb[s ^ 1][r - TM] = SB::frag(rb, wn0 + 16 * (r - TM) + frow, kq);
                        }
                    }
                }
                if (do_commit || do_issue) {
                    // staging units spread over the MFMAs that follow the fragment reads
                    constexpr int Q0 = NREAD / 2 < NMFMA ? NREAD / 2 : 0;
                    constexpr int QN = NMFMA - Q0;
#pragma unroll
                    for (int u = 0; u < NUNIT; ++u) {
                        if (Q0 + u * QN / NUNIT == q) {
                            if (do_commit) {
                                if (u < SA::NV) sa.template commit_one<MASK>(u, na, k_next, p.K);
                                else sb.template commit_one<MASK>(u - SA::NV, nb, k_next, p.K);
                            } else {
                                if (u < SA::NV) sa.issue_one(u, A);
                                else sb.issue_one(u - SA::NV, B);
                            }
                        }
                    }
                }
                if (do_issue && q == NMFMA - 1) {
                    sa.fetch_step(p.k_a, k_base + (int64_t)(kt + 3) * BK);
                    sb.fetch_step(p.k_b, k_base + (int64_t)(kt + 3) * BK);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#ifdef CTC_STAMPS
        unsigned long long t4;
        CTC_STAMP(t4);
        st_a += t1 - t0; st_b += t2 - t1; st_c += t3 - t2; st_d += t4 - t3; st_n += 1;
#endif
    };

#ifdef CTC_STAMPS
    CTC_STAMP(st_loop0);
#endif
    {
        int kt = 0;
        for (; kt + 2 < nk; ++kt) tile_step(std::false_type(), std::true_type(), std::true_type(), kt);
        if (kt + 1 < nk) {
            // commits the last tile, which may cross K
            tile_step(std::true_type(), std::true_type(), std::false_type(), kt);
            ++kt;
        }
        tile_step(std::false_type(), std::false_type(), std::false_type(), kt);
    }
#ifdef CTC_STAMPS
    CTC_STAMP(st_loop1);
#endif

    store_tile<T, BM, BN, TM, TN>(p, acc, s_coff, C, wm0, wn0, lane, partial, list_id, chunk);
#ifdef CTC_CLOCK
    if (t == 0 && blockIdx.y == 0 && (blockIdx.x == 700 || blockIdx.x == 2500 || blockIdx.x == 4500)) {
        const unsigned long long dc = __builtin_readcyclecounter() - ck_core0, dr = __builtin_amdgcn_s_memrealtime() - ck_real0;
        printf("[clock] block %d: %llu core ticks in %llu x 10 ns -> %.3f GHz\n", (int)blockIdx.x, dc, dr, (double)dc / (10.0 * (double)dr));
    }
#endif
#ifdef CTC_STAMPS
    {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long st_exit;
        CTC_STAMP(st_exit);
        if (t == 0 && blockIdx.x < 8192 && blockIdx.y == 0) {
            unsigned long long *o = g_stamp_sums + 8 * blockIdx.x;
            o[0] = st_a; o[1] = st_b; o[2] = st_c; o[3] = (__builtin_amdgcn_s_memrealtime() - st_real0); o[4] = st_n;
            o[5] = st_loop0 - st_entry; o[6] = st_loop1 - st_loop0; o[7] = st_exit - st_loop1;
        }
    }
#endif
}

// Sums the K chunks of the split tiles in chunk order and writes C = alpha * sum + beta * C.
// One workgroup per (tile, 16 rows); a thread owns two neighbouring columns.
template <typename T, int BM, int BN>
__global__ void __launch_bounds__(256) splitk_fixup_kernel(const GemmParams p)
{
    constexpr int ROWS = 16, VPR = BN / 2, RPP = 256 / VPR;  // vectors per row, rows per pass
    const int tile_rel = blockIdx.x / (BM / ROWS);
    const int rblk = blockIdx.x % (BM / ROWS);
    int sid = p.dp_tiles + tile_rel, bz = 0;
    if (p.flat_batch > 0) {  // the tile id counts through the batch entries (GemmParams::flat_batch)
        bz = sid / (p.nbm * p.nbn);
        sid -= bz * (p.nbm * p.nbn);
    }
    const int width = p.group * p.nbn;
    const int first_m = (sid / width) * p.group;
    const int gsz = min(p.nbm - first_m, p.group);
    const int tm = first_m + (sid % width) % gsz;
    const int tn = (sid % width) / gsz;
    const T *slab0 = static_cast<const T *>(p.ws) + (int64_t)tile_rel * p.split * (BM * BN);
    T *C = static_cast<T *>(p.C) + p.bat_c[bz];
    const T alpha = (T)p.alpha, beta = (T)p.beta;
    const int cv = threadIdx.x % VPR;
    const int64_t n = (int64_t)tn * BN + 2 * cv;
    const bool ok0 = n < p.N, ok1 = n + 1 < p.N;
    const int64_t c0 = p.col_c[ok0 ? n : 0], c1 = p.col_c[ok1 ? n + 1 : 0];
    for (int rr = threadIdx.x / VPR; rr < ROWS; rr += RPP) {
        const int row = rblk * ROWS + rr;
        const int64_t m = (int64_t)tm * BM + row;
        if (m >= p.M) continue;
        // eight chunks at a time into separate sums: eight loads in flight instead of one (a tile cut
        // several hundred ways otherwise spends longer here than in the GEMM); fixed order of additions
        T a0[8], a1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a0[u] = a1[u] = 0;
        int c = 0;
        for (; c + 8 <= p.split; c += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const T *src = slab0 + (int64_t)(c + u) * (BM * BN) + row * BN + 2 * cv;
                a0[u] += src[0];
                a1[u] += src[1];
            }
        }
        for (int u = 0; c < p.split; ++c, ++u) {
            const T *src = slab0 + (int64_t)c * (BM * BN) + row * BN + 2 * cv;
            a0[u] += src[0];
            a1[u] += src[1];
        }
        const T s0 = ((a0[0] + a0[1]) + (a0[2] + a0[3])) + ((a0[4] + a0[5]) + (a0[6] + a0[7]));
        const T s1 = ((a1[0] + a1[1]) + (a1[2] + a1[3])) + ((a1[4] + a1[5]) + (a1[6] + a1[7]));
        const int64_t ro = p.row_c[m];
        if (ok0) {
            T v = alpha * s0;
            if (beta != (T)0) v += beta * C[ro + c0];
            C[ro + c0] = v;
        }
        if (ok1) {
            T v = alpha * s1;
            if (beta != (T)0) v += beta * C[ro + c1];
            C[ro + c1] = v;
        }
    }
}

// K-tile depth: 16 by default; the large f32 tile takes 32 (the same bytes per tile as f64 at 16:
// an f32 MFMA holds the pipe half as long, so a barrier per 16 k came twice as often in time).
template <typename T, int BM> constexpr int tile_bk()
{
#ifdef CTC_GEMM_BK_F32
    return (sizeof(T) == 4 && BM == 128) ? CTC_GEMM_BK_F32 : CTC_GEMM_BK;
#else
    return (sizeof(T) == 4 && BM == 128) ? 32 : CTC_GEMM_BK;
#endif
}

// workgroup slots of the current device for one kernel instantiation (cached per device).  Tag
// names the instantiation: every gemm_kernel has the same function TYPE, so a cache keyed by the
// type of `kernel` alone would be shared by all of them.
template <typename Tag, typename K> int resident_slots(K kernel, int threads)
{
    static std::atomic<int> cache[64];  // 0: not queried yet; host threads of several devices may race here
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cache[dev].load() == 0) {
        int cus = 0, per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, threads, 0) != hipSuccess) return 0;
        cache[dev].store(cus * per_cu > 0 ? cus * per_cu : -1);
    }
    const int slots = cache[dev].load();
    return slots > 0 ? slots : 0;
}
template <typename T, int LA, int LB, int BM, int BN> struct KernelTag {};

inline int device_cus()
{
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cache[dev].load() == 0) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        cache[dev].store(cus > 0 ? cus : -1);
    }
    const int cus = cache[dev].load();
    return cus > 0 ? cus : 0;
}

// Splits tiles along K.  The unit of balance is the CU, not the workgroup slot: the kernels are bound by a CU's
// matrix pipes, so a launch takes (fixed part) + (workgroups on the busiest CU + d) x (a workgroup's K-tiles),
// whether those workgroups are resident together or one after another - measured, the time of an uncut launch
// steps up every 256 tiles (64 x 64 tile, fp64, K = 4096: 0.193 / 0.298 / 0.422 / 0.547 ms for up to
// 256 / 512 / 768 / 1024 tiles) although 1024 of its workgroups are resident at once (tools/tail_cost_probe.py).
// d: while a CU holds a single workgroup of the small tile, that workgroup is latency-bound and runs at about two
// thirds of the CU's rate (the first step above is 1.5 - 1.85 times the later ones); it stays with the longest
// items of a launch.  The large tile shows less of it (0.25 fp64, 0.5 fp32).
// Three schedules, costs in K-tile times (nk per whole tile, nk / f + overhead per chunk, where the overhead -
// prologue, raw accumulators to the workspace, the chunk's share of the fix-up pass - is `ovh` K-tiles, and a cut
// at all - the second launch - `pen` more), tiles = w * cus + tail:
//   whole tiles only:                  (ceil(tiles / cus) + d) * nk
//   the tail cut into f chunks each:   (w + d) * nk + ceil(tail * f / cus) * chunk + pen      (w >= 1, tail > 0)
//   every tile cut (tiles <= 4 cus):   (ceil(tiles * f / cus) + d) * chunk + pen
// Without the overhead term long splits look free: 2304^3 cut 11 ways ran no faster than uncut (45 TFLOP/s), cut
// 3 ways it runs at 60.  (Counting in rounds of slots instead cut 256 tiles of fp32 K = 1024 seven ways: 44.7
// TFLOP/s against 54.7 uncut; 512 tiles of K = 576 three ways: 49.9 against 66.4.)
constexpr int kMinChunkTiles = 8;

struct SplitChoice {
    int f = 1;          // chunks per cut tile (1: nothing is cut)
    int64_t cut = 0;    // tiles that are cut: the last `cut` ones
};

inline SplitChoice best_split(int64_t tiles, int cus, int nk, int ovh, int pen, double d)
{
    int fmax = nk / ovh;
    if (fmax > 1024) fmax = 1024;  // one tile with a very long K may use every CU several times over
    SplitChoice best;
    double best_cost = ((double)((tiles + cus - 1) / cus) + d) * (double)nk;
    const int64_t w = tiles / cus, tail = tiles % cus;
    for (int f = 2; f <= fmax; ++f) {
        const double chunk = (double)nk / f + ovh;
        if (w >= 1 && tail > 0) {
            const double cost = ((double)w + d) * nk + (double)((tail * f + cus - 1) / cus) * chunk + pen;
            if (cost < best_cost * 0.98) {
                best_cost = cost;
                best.f = f;
                best.cut = tail;
            }
        }
        if (tiles <= 4 * (int64_t)cus) {
            const double cost = ((double)((tiles * f + cus - 1) / cus) + d) * chunk + pen;
            if (cost < best_cost * 0.98) {
                best_cost = cost;
                best.f = f;
                best.cut = tiles;
            }
        }
    }
    return best;
}

// the schedule of before (CTC_TC_SPLIT_BY_SLOTS=1, A/B runs): the last, partly filled round of `slots` workgroups in
// f chunks per tile, rounds of chunks counted whole
inline SplitChoice best_split_by_slots(int64_t tiles, int slots, int nk)
{
    SplitChoice best;
    const int64_t tail = tiles % slots;
    if (tail == 0) return best;
    int fmax = nk / kMinChunkTiles;
    if (fmax > 1024) fmax = 1024;
    double best_cost = (double)nk;
    for (int f = 2; f <= fmax; ++f) {
        const double cost = (double)((tail * f + slots - 1) / slots) * ((double)nk / f + kMinChunkTiles);
        if (cost < best_cost * 0.98) {
            best_cost = cost;
            best.f = f;
            best.cut = tail;
        }
    }
    return best;
}

template <typename T, int LA, int LB, int BM, int BN>
hipError_t schedule_one(const GemmParams &p, GemmSchedule *out)
{
    constexpr int BK = tile_bk<T, BM>(), WM = 2, WN = 2;
    static_assert(3 * BK <= kKTablePad, "the prologue reads the k tables three K-tiles ahead");
    GemmSchedule s;
    s.nbm = (int)((p.M + BM - 1) / BM);
    s.nbn = (int)((p.N + BN - 1) / BN);
    int64_t tiles = (int64_t)s.nbm * s.nbn;
    if (tiles > 0x7FFFFFFF) return hipErrorInvalidValue;
    // a batched launch: one list of tiles over all entries whenever its length fits a grid (2^24 workgroups of
    // 256 threads), so that it is cut like a single product (CTC_TC_NO_FLAT_BATCH: A/B runs)
    static const bool no_flat = getenv("CTC_TC_NO_FLAT_BATCH") != nullptr;
    if (p.batch > 1 && !no_flat && tiles * p.batch <= 0x7FFFFF) {
        s.flat_batch = p.batch;
        tiles *= p.batch;
    }
    s.dp_tiles = (int)tiles;
    s.split = 1;
    s.tail_tiles = 0;
    s.ws_bytes = 0;
    static const bool no_split = getenv("CTC_TC_NO_SPLITK") != nullptr;
    const int slots = resident_slots<KernelTag<T, LA, LB, BM, BN>>(gemm_kernel<T, LA, LB, BM, BN, BK, WM, WN>, 64 * WM * WN);
    const int nk = (int)((p.K + BK - 1) / BK);
    // (the small tile's chunks measure cheaper: 4 - 7 K-tiles of overhead fit its cut launches; a second launch
    // is some 5 us, 10 of its fp64 K-tiles and 20 of its fp32 ones - the fp32 cuts that lose to whole tiles at
    // K = 400 ... 800 lose by what 30 make up; a quarter of that in K-tiles of the large tile)
    constexpr int kChunkOverhead = BM <= 64 ? 6 : kMinChunkTiles;
    constexpr int kCutPenalty = BM <= 64 ? (sizeof(T) == 4 ? 32 : 10) : (sizeof(T) == 4 ? 8 : 3);
    // (large tile, operands through registers: 256 tiles of fp64 K = 4097 gain 3.5 % cut in two, 192 of fp32 17 % cut in five)
    constexpr double kLoneWorkgroup = BM <= 64 ? (sizeof(T) == 4 ? 0.75 : 0.5) : (sizeof(T) == 4 ? 0.5 : 0.25);
    static const bool by_slots = getenv("CTC_TC_SPLIT_BY_SLOTS") != nullptr;
    const int cus = device_cus();
    if (!no_split && slots > 0 && cus > 0 && (p.batch == 1 || s.flat_batch > 0)) {
        // The large tile with more tiles than slots keeps the rounds of slots: its late workgroups do not land
        // one per CU (fp64, element loads of A, K = 4097: 640, 768 and 1024 tiles all take the time of two full
        // rounds; 768 tiles cut by rounds of slots run at 67.2 TFLOP/s, left whole as the balance per CU
        // suggests at 52.4), and this kernel has no form in which workgroups walk their items.
        const bool slots_rule = by_slots || (BM > 64 && tiles > slots);
        const SplitChoice c = slots_rule ? best_split_by_slots(tiles, slots, nk)
                                         : best_split(tiles, cus, nk, kChunkOverhead, kCutPenalty, kLoneWorkgroup);
        if (c.f >= 2 && c.cut > 0) {
            s.tail_tiles = (int)c.cut;
            s.dp_tiles = (int)(tiles - c.cut);
            s.split = c.f;
            s.ws_bytes = (size_t)c.cut * c.f * BM * BN * sizeof(T);
        }
    }
    *out = s;
    return hipSuccess;
}

template <typename T, int LA, int LB, int BM, int BN>
hipError_t launch_one(const GemmParams &p, const GemmSchedule &sch, void *ws, hipStream_t stream)
{
    constexpr int BK = tile_bk<T, BM>(), WM = 2, WN = 2;
    GemmParams q = p;
    q.nbm = sch.nbm;
    q.nbn = sch.nbn;
    q.dp_tiles = sch.dp_tiles;
    q.split = sch.split;
    q.flat_batch = sch.flat_batch;
    q.plain_order = (sch.flat_batch > 0 && sch.nbm * sch.nbn == 1) ? 1 : 0;
    q.ws = ws;
    if (q.group < 1) q.group = 1;
    if (sch.tail_tiles > 0 && ws == nullptr) return hipErrorInvalidValue;
    dim3 grid((unsigned)(sch.dp_tiles + sch.tail_tiles * sch.split), sch.flat_batch > 0 ? 1u : (unsigned)p.batch, 1);
    dim3 block(64 * WM * WN, 1, 1);
    // diagnostic knob: extra dynamic LDS per block lowers the number of co-resident blocks per CU
    static const size_t extra_lds = getenv("CTC_TC_DEBUG_EXTRA_LDS") ? (size_t)atol(getenv("CTC_TC_DEBUG_EXTRA_LDS")) : 0;
    hipLaunchKernelGGL((gemm_kernel<T, LA, LB, BM, BN, BK, WM, WN>), grid, block, extra_lds, stream, q);
    if (sch.tail_tiles > 0)
        hipLaunchKernelGGL((splitk_fixup_kernel<T, BM, BN>), dim3((unsigned)(sch.tail_tiles * (BM / 16))), dim3(256), 0,
                           stream, q);
#ifdef CTC_STAMPS
    {
        (void)hipStreamSynchronize(stream);
        static std::vector<unsigned long long> h(8 * 8192);
        (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_stamp_sums), h.size() * 8);
        const int nb = std::min<int>(8192, q.nbm * q.nbn);
        std::vector<double> seg[4];
        for (int b = 0; b < nb; ++b)
            for (int i = 0; i < 4; ++i) seg[i].push_back((double)h[8 * b + i] / (double)std::max<unsigned long long>(1, h[8 * b + 4]));
        double med[4];
        for (int i = 0; i < 4; ++i) { std::sort(seg[i].begin(), seg[i].end()); med[i] = seg[i][nb / 2]; }
        std::vector<double> ph[3];
        for (int b = 0; b < nb; ++b)
            for (int i = 0; i < 3; ++i) ph[i].push_back((double)h[8 * b + 5 + i]);
        for (int i = 0; i < 3; ++i) std::sort(ph[i].begin(), ph[i].end());
        fprintf(stderr, "[stamps] per block ticks (median/max): prologue %.0f/%.0f | loop %.0f/%.0f | epilogue %.0f/%.0f\n",
                ph[0][nb / 2], ph[0][nb - 1], ph[1][nb / 2], ph[1][nb - 1], ph[2][nb / 2], ph[2][nb - 1]);
        fprintf(stderr, "[stamps] per K-tile ticks (median over %d blocks): ksteps0..n-3 %.0f | commit kstep %.0f | barrier %.0f | realtime %.0f\n",
                nb, med[0], med[1], med[2], med[3]);
    }
#endif
    return hipGetLastError();
}

template <typename T, int BM, int BN>
hipError_t dispatch_layouts(int la, int lb, const GemmParams &p, GemmSchedule *sch_out, const GemmSchedule *sch_in,
                            void *ws, hipStream_t stream)
{
#define CTC_CASE(A_, B_)                                                                                      \
    case A_ * 3 + B_:                                                                                          \
        return sch_out ? schedule_one<T, A_, B_, BM, BN>(p, sch_out) : launch_one<T, A_, B_, BM, BN>(p, *sch_in, ws, stream);
    switch (la * 3 + lb) {
#ifdef CTC_EXP_ONE  // diagnostic builds of a single instantiation, e.g. -DCTC_EXP_ONE="CTC_CASE(1,1)" (tools/build_variant.sh)
        CTC_EXP_ONE
#else
        CTC_CASE(0, 0) CTC_CASE(0, 1) CTC_CASE(0, 2) CTC_CASE(1, 0) CTC_CASE(1, 1) CTC_CASE(1, 2)
        CTC_CASE(2, 0) CTC_CASE(2, 1) CTC_CASE(2, 2)
#endif
    default: return hipErrorInvalidValue;
    }
#undef CTC_CASE
}

template <typename T>
hipError_t dispatch_tiles(int la, int lb, int bm, int bn, const GemmParams &p, GemmSchedule *sch_out,
                          const GemmSchedule *sch_in, void *ws, hipStream_t stream)
{
    if (bm == 128 && bn == 128) return dispatch_layouts<T, CTC_GEMM_BIG_BM, CTC_GEMM_BIG_BN>(la, lb, p, sch_out, sch_in, ws, stream);
#ifndef CTC_EXP_ONE
    if (bm == 64 && bn == 64) return dispatch_layouts<T, 64, 64>(la, lb, p, sch_out, sch_in, ws, stream);
#endif
    return hipErrorInvalidValue;
}

hipError_t dispatch(int dtype, int la, int lb, int bm, int bn, const GemmParams &p, GemmSchedule *sch_out,
                    const GemmSchedule *sch_in, void *ws, hipStream_t stream)
{
    if (la < 0 || la > 2 || lb < 0 || lb > 2) return hipErrorInvalidValue;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.batch <= 0 || p.batch > 65535) return hipErrorInvalidValue;
#if !defined(CTC_EXP_ONE) || !defined(CTC_EXP_F32)
    if (dtype == 0) return dispatch_tiles<double>(la, lb, bm, bn, p, sch_out, sch_in, ws, stream);
#endif
#if !defined(CTC_EXP_ONE) || defined(CTC_EXP_F32)
    if (dtype == 1) return dispatch_tiles<float>(la, lb, bm, bn, p, sch_out, sch_in, ws, stream);
#endif
    return hipErrorInvalidValue;
}

}  // namespace

hipError_t launch_splitk_fixup_128(int dtype, const GemmParams &p, int tail_tiles, hipStream_t stream)
{
    const dim3 grid((unsigned)(tail_tiles * (128 / 16)));
    if (dtype == 0) hipLaunchKernelGGL((splitk_fixup_kernel<double, 128, 128>), grid, dim3(256), 0, stream, p);
    else if (dtype == 1) hipLaunchKernelGGL((splitk_fixup_kernel<float, 128, 128>), grid, dim3(256), 0, stream, p);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t schedule_gemm(int dtype, int layout_a, int layout_b, int bm, int bn, const GemmParams &p, GemmSchedule *out)
{
    if (dtype == 2) return schedule_gemm_bf16(layout_a, layout_b, bm, p, out);
    if (dma_gemm_supported(dtype, layout_a, layout_b, bm, bn, p)) return schedule_gemm_dma(dtype, layout_a, layout_b, p, out);
    return dispatch(dtype, layout_a, layout_b, bm, bn, p, out, nullptr, nullptr, nullptr);
}

hipError_t launch_gemm(int dtype, int layout_a, int layout_b, int bm, int bn, const GemmParams &p,
                       const GemmSchedule &schedule, void *workspace, hipStream_t stream)
{
    if (dtype == 2) return launch_gemm_bf16(layout_a, layout_b, bm, p, schedule, workspace, stream);
    if (dma_gemm_supported(dtype, layout_a, layout_b, bm, bn, p))
        return launch_gemm_dma(dtype, layout_a, layout_b, p, schedule, workspace, stream);
    return dispatch(dtype, layout_a, layout_b, bm, bn, p, nullptr, &schedule, workspace, stream);
}

}  // namespace ctc
