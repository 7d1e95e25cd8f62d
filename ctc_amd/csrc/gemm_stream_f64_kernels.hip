// Persistent fp64 MFMA GEMM for contractions with a short K under a large C: the stores of a finished
// 128 x 64 item are dealt out behind the MFMAs of the next one (see gemm_stream_kernels.hip for the
// scheme; this is its fp64 form), and every byte that travels global -> LDS goes by LDS-DMA
// (global_load_lds_dwordx4), so the waves hold no staging registers, execute no LDS writes and wait for
// a load exactly once per K-tile, a whole K-tile after it was issued:
//   - three LDS stages of one 128 x BK A tile and one 64 x BK B tile (no padding), in two geometries:
//     K-tile 32 (48 KB a stage, one workgroup per CU, every layout) and K-tile 16 (24 KB, two workgroups
//     per CU, outer-major operands: see Geom); during K-tile g the DMA of K-tile g + 2 is in flight and
//     that of g + 1 is awaited just before the barrier that ends g;
//   - a DMA instruction writes 1 KB of consecutive LDS (16 bytes per lane), so bank conflicts are
//     removed by choosing WHICH 16-byte chunk a lane fetches instead of by padding:
//       K-major operand (unit stride along K):  image [row][32 k], chunk c of row r stored at chunk
//           position c ^ (r & 15): the 16 rows x 2 chunks a fragment read touches cover all 64 banks;
//       outer-major operand (unit stride along the rows): image [k][rows], chunk c (rows 2c, 2c+1) of
//           line k stored at position c ^ ((k & 1) << 3): lines k and k + 1 of a fragment land in
//           different halves of the banks;
//   - the element offsets of an item (A rows, B columns, C rows, C columns) are copied from the plan's
//     tables into an LDS ring by DMA as well, two items ahead.
// The only ordinary vector-memory instructions left are the stores of C, so the in-order memory
// counter never makes a wave wait for a store on the way to its operands.
// Needs vector layouts on both operands, beta == 0, batch == 1, even M and N, K a multiple of the K-tile
// and at least 64, and k offsets that are a multiple of k (one contracted mode, or modes that fuse:
// ka_stride / kb_stride); everything else takes the register-staged form or the classic kernel.
//
// This file is compiled once per kernel instantiation (each takes minutes: -DCTC_SX_PART=1..10 selects one
// of the five layout / K-tile combinations, with or without the alpha multiply) plus once for the
// dispatcher (-DCTC_SX_PART=0), so that the parts build in parallel; without the macro it is one
// translation unit holding everything.
#include "gemm_device.hpp"
#include "kernels.hpp"
#include "lds_dma.hpp"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#ifndef CTC_SX_PART
#define CTC_SX_PART (-1)
#endif

namespace ctc {

namespace stream_f64 {
template <int LA, int LB, int BK, bool SCALE> hipError_t launch_scaled(const GemmParams &p, hipStream_t stream);
}

#if CTC_SX_PART != 0
namespace {

using namespace gemm_detail;
using namespace lds_dma;

constexpr int kBM = 128, kBN = 64, kStages = 3;
// Two geometries.  K-tile 32: one workgroup per CU (144 KB of operand stages), every layout.  K-tile 16:
// TWO workgroups per CU (72 KB of stages each, at most 128 + 128 registers a lane), so that a second wave
// on every SIMD issues MFMAs while the first waits for a barrier, a DMA or an issue slot.
template <int BK> struct Geom {
    static constexpr int kABytes = kBM * BK * 8, kBBytes = kBN * BK * 8, kStageBytes = kABytes + kBBytes;
    static constexpr bool kTwoPerCu = (BK == 16);
    // Offset ring, in table entries.  An item's A rows and B columns are read once, when the loader moves
    // on to the item; its C rows and C columns while the NEXT item is multiplied.
    //   K-tile 32: four slots of [A rows | B columns | C rows | C columns], item it in slot it & 3,
    //              fetched during the first K-tile of item it - 2;
    //   K-tile 16: ONE slot of [A rows | B columns] and four of [C rows | C columns], fetched during the
    //              last K-tile of item it - 2 (behind its barrier: every wave has moved its loader on).
    static constexpr int kAbEntries = kBM + kBN, kCEntries = kBM + kBN;
    static constexpr int kRingEntries = kTwoPerCu ? kAbEntries + 4 * kCEntries : 4 * (kAbEntries + kCEntries);
    // entry index of the [A rows | B columns] / [C rows | C columns] run of item it
    __device__ static __forceinline__ int ab_entry(int it) { return kTwoPerCu ? 0 : (it & 3) * (kAbEntries + kCEntries); }
    __device__ static __forceinline__ int c_entry(int it)
    {
        return kTwoPerCu ? kAbEntries + (it & 3) * kCEntries : (it & 3) * (kAbEntries + kCEntries) + kAbEntries;
    }
    // K-tiles of an item that carry the previous item's stores (K >= 64: at least that many K-tiles)
    static constexpr int kStoreTiles = 64 / BK;
};

// DMA staging of one BR x 32 fp64 operand tile by 4 waves; NI instructions per wave per tile.
// Address arithmetic is kept off the vector ALU (a wave that feeds the matrix pipe has about sixteen
// issue slots per MFMA): per item a lane keeps byte POINTERS to what it fetches at k = 0, and a DMA adds
// a wave-uniform byte offset for the K-tile (and, outer-major, for the line) to one of them.
template <int LAY, int BR, int BK> struct DmaStager {
    static constexpr bool KMAJ = (LAY != kLayOuterVec);
    static_assert(!KMAJ || BK == 32, "the K-major image is laid out for rows of 32 contracted elements");
    static constexpr int NI = BR * BK * 8 / 1024 / 4;
    static constexpr int NROW = KMAJ ? NI : (BR == 128 ? 2 : 1);  // distinct rows among the NI fetches
    const char *ptr[NI];   // what instruction i fetches at k0 = 0
    int64_t rowoff[NROW];  // as read from the ring; turned into ptr[] by finish_rows
    int64_t kofs8[NI];     // byte offset of instruction i's k (lane part and line part) from the row at k = 0
    int lane, wave;

    // index into the item's row table of row j
    __device__ __forceinline__ int row_index(int j) const
    {
        if (KMAJ) return 4 * (wave * NI + j) + (lane >> 4);
        if (BR == 128) return 2 * (lane ^ (j << 3));
        return 2 * ((lane & 31) ^ ((lane >> 5) << 3));
    }
    // contracted index, within a K-tile, of what instruction i fetches
    __device__ __forceinline__ int k_of(int i) const
    {
        if (KMAJ) return 2 * ((lane & 15) ^ ((4 * i + (lane >> 4)) & 15));
        if (BR == 128) return wave * NI + i;
        return 2 * (wave * NI + i) + (lane >> 5);
    }
    __device__ __forceinline__ void init(int lane_, int wave_, int64_t kstride)
    {
        lane = lane_;
        wave = wave_;
#pragma unroll
        for (int i = 0; i < NI; ++i) kofs8[i] = 8 * (int64_t)k_of(i) * kstride;
    }
    __device__ __forceinline__ void load_rows(const int64_t *tile_rows)
    {
#pragma unroll
        for (int j = 0; j < NROW; ++j) rowoff[j] = tile_rows[row_index(j)];
    }
    // the same from inside the main loop: issue, then finish_rows() waits
    __device__ __forceinline__ void load_rows_async(unsigned tile_rows_lds)
    {
#pragma unroll
        for (int j = 0; j < NROW; ++j) lds_read64<0>(rowoff[j], tile_rows_lds + 8 * row_index(j));
    }
    __device__ __forceinline__ void finish_rows(const double *__restrict__ base, bool wait)
    {
        if (wait) {
#pragma unroll
            for (int j = 0; j < NROW; ++j) lds_wait(rowoff[j]);
        }
#pragma unroll
        for (int i = 0; i < NI; ++i)
            ptr[i] = reinterpret_cast<const char *>(base + rowoff[KMAJ ? i : (BR == 128 ? (i & 1) : 0)]) + kofs8[i];
    }
    // instruction i of the K-tile whose first contracted index is k0 (kofs = 8 * k0 * kstride, wave-uniform);
    // image = the operand's part of a stage.  K is a multiple of the K-tile here, so nothing is masked.
    template <int i> __device__ __forceinline__ void issue(int64_t kofs, unsigned char *image) const
    {
        dma16(ptr[i] + kofs, image + (wave * NI + i) * 1024);
    }
};

// Byte address, inside the operand's image, of the MFMA fragment element of lane (frow, fk) for row block
// b (16 rows from w0 + 16 b) and k-step kk (k = 4 kk + fk).  lane_* are per-lane constants from frag_init.
template <int LAY, int BR> struct FragAddr {
    static constexpr bool KMAJ = (LAY != kLayOuterVec);
    int c0, c1;  // KMAJ: c0 = frow * 256 + (fk & 1) * 8 + w0 * 256, c1 = (frow ^ (fk >> 1)) << 4
                 // outer: c0 / c1 = lane base for even / odd row blocks
    __device__ __forceinline__ void init(int frow, int fk, int w0)
    {
        if (KMAJ) {
            c0 = (w0 + frow) * 256 + (fk & 1) * 8;
            c1 = (frow ^ (fk >> 1)) << 4;
        } else {
            const int base = fk * BR * 8 + (w0 + frow) * 8, p = fk & 1;
            c0 = base + 128 * p;
            c1 = base - 128 * p;
        }
    }
    template <int b, int kk> __device__ __forceinline__ int at() const
    {
        if (KMAJ) return c0 + b * 16 * 256 + (c1 ^ (kk << 5));
        return ((b & 1) ? c1 : c0) + 128 * b + kk * 4 * BR * 8;
    }
    // at<b, kk>() split into a register part (to which the caller adds the stage's LDS address) and a
    // compile-time part that fits the instruction's offset field
    template <int b, int kk> __device__ __forceinline__ int var() const
    {
        if (KMAJ) return c0 + (c1 ^ (kk << 5));
        return (b & 1) ? c1 : c0;
    }
    template <int b, int kk> static constexpr int imm() { return KMAJ ? b * 16 * 256 : 128 * b + kk * 4 * BR * 8; }
};

// SCALE: alpha != 1, applied to each element on its way out (one fp64 multiply per store; the
// instantiations for alpha == 1 carry none)
template <int LA, int LB, int BK, bool SCALE>
__global__ void __launch_bounds__(256, Geom<BK>::kTwoPerCu ? 2 : 1) gemm_stream_f64_kernel(const GemmParams p)
{
    typedef Geom<BK> G_;
    constexpr int BM = kBM, BN = kBN, WN = 2;
    constexpr int kABytes = G_::kABytes, kStageBytes = G_::kStageBytes;
    constexpr bool TWO = G_::kTwoPerCu;
    constexpr int TM = 4, TN = 2, NMFMA = TM * TN, KSTEPS = BK / 4, NIDX = NMFMA * KSTEPS;
    typedef Traits<double> Tr;
    typedef Tr::acc_t acc_t;
    typedef DmaStager<LA, BM, BK> SA;
    typedef DmaStager<LB, BN, BK> SB;
    constexpr int NDMA = SA::NI + SB::NI;            // operand DMAs per wave per K-tile
    constexpr int STORE_TILES = G_::kStoreTiles;     // K-tiles of an item that carry the previous item's stores
    // the DMAs of a K-tile are all issued before the barrier that ends it
    constexpr int DMA_FIRST = 3, DMA_STEP = 4;
    static_assert(DMA_FIRST + DMA_STEP * (NDMA - 1) < (KSTEPS - 1) * NMFMA, "DMA schedule");

    __shared__ __attribute__((aligned(1024))) unsigned char smem[kStages * kStageBytes];
    __shared__ __attribute__((aligned(16))) int64_t s_ring[G_::kRingEntries];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm0 = (wave / WN) * (TM * 16);
    const int wn0 = (wave % WN) * (TN * 16);
    const int frow = lane & 15;
    const int fk = lane >> 4;

    const int G = (int)gridDim.x;
    const int bid = (int)blockIdx.x;
#ifdef CTC_SX_CLOCK  // diagnostic build: shader clock during the kernel = core-clock ticks / 100 MHz ticks
    const unsigned long long ck_core0 = __builtin_readcyclecounter(), ck_real0 = __builtin_amdgcn_s_memrealtime();
#endif
    const int wslot = (G & 7) == 0 ? (bid & 7) * (G >> 3) + (bid >> 3) : bid;
    const int64_t tiles = (int64_t)p.nbm * p.nbn;
    if (wslot >= tiles) return;
    const int n_items = (int)((tiles - wslot + G - 1) / G);
    // Items are visited in order, G tile ids apart.  The tile id is kept as (group of tile rows, position
    // inside the group) and advanced by additions: integer divisions, which the wave would do through
    // the vector ALU, are left to the last, partly filled group of tile rows.
    const int width = p.group * p.nbn;
    const int step_grp = G / width, step_rem = G % width;
    const int group_shift = (p.group & (p.group - 1)) == 0 ? __builtin_ctz((unsigned)p.group) : -1;
    int fetch_grp = wslot / width, fetch_rem = wslot % width;
    auto next_origin = [&](int64_t &m0, int64_t &n0) {
        const int first_m = fetch_grp * p.group;
        const int gsz = min(p.nbm - first_m, p.group);
        int tm, tn;
        if (gsz == p.group && group_shift >= 0) {
            tm = first_m + (fetch_rem & (p.group - 1));
            tn = fetch_rem >> group_shift;
        } else {
            tm = first_m + fetch_rem % gsz;
            tn = fetch_rem / gsz;
        }
        m0 = (int64_t)tm * BM;
        n0 = (int64_t)tn * BN;
        fetch_grp += step_grp;
        fetch_rem += step_rem;
        if (fetch_rem >= width) {
            fetch_rem -= width;
            fetch_grp += 1;
        }
    };

    const double *A = static_cast<const double *>(p.A) + p.bat_a[0];
    const double *B = static_cast<const double *>(p.B) + p.bat_b[0];
    double *C = static_cast<double *>(p.C) + p.bat_c[0];
    const int nk = (int)(p.K / BK);

    // Ring slot of item `it`: the four runs of table entries by DMA, one instruction per wave (16 bytes =
    // two entries per lane).  Entries past the operand's extent repeat its last two rows (finite data whose
    // products are never stored); stores are bounded by the item's extents, not by sentinel offsets.
    int m_lim_new = 0, n_lim_new = 0;  // out of ring_fetch: extents of the item it fetched
    auto ring_fetch = [&](int it) {  // items must be fetched in order: 0, 1, 2, ...
        int64_t m0, n0;
        next_origin(m0, n0);
        const bool rows = (wave & 1) == 0;                 // waves 0, 2: row tables; 1, 3: column tables
        const int64_t *tab = wave == 0 ? p.row_a : (wave == 1 ? p.col_b : (wave == 2 ? p.row_c : p.col_c));
        const int64_t first = rows ? m0 : n0, lim = rows ? p.M : p.N;
        // a column run is half an instruction long: only its lower 32 lanes take part
        int64_t g = first + 2 * (rows ? lane : (lane & 31));
        if (g > lim - 2) g = lim - 2;  // M and N are even here, so a pair never straddles the end
        const int dst = (wave < 2 ? G_::ab_entry(it) : G_::c_entry(it)) + (rows ? 0 : BM);
        if (rows || lane < 32) dma16(tab + g, reinterpret_cast<unsigned char *>(&s_ring[dst]));
        m_lim_new = (int)(p.M - m0 < BM ? p.M - m0 : BM);
        n_lim_new = (int)(p.N - n0 < BN ? p.N - n0 : BN);
    };

    SA sa;
    SB sb;
    sa.init(lane, wave, p.ka_stride);
    sb.init(lane, wave, p.kb_stride);
    FragAddr<LA, BM> fa_addr;
    FragAddr<LB, BN> fb_addr;
    fa_addr.init(frow, fk, wm0);
    fb_addr.init(frow, fk, wn0);

    acc_t acc[2][TM][TN];

    // ---- prologue
    ring_fetch(0);
    int m_lim_0 = m_lim_new, n_lim_0 = n_lim_new, m_lim_1 = 0, n_lim_1 = 0;
    if (!TWO && n_items > 1) {
        ring_fetch(1);
        m_lim_1 = m_lim_new;
        n_lim_1 = n_lim_new;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    sa.load_rows(&s_ring[G_::ab_entry(0)]);
    sb.load_rows(&s_ring[G_::ab_entry(0) + BM]);
    sa.finish_rows(A, false);
    sb.finish_rows(B, false);
    if (TWO) {
        // the one [A rows | B columns] slot has been read by every wave: item 1 may take it
        __syncthreads();
        if (n_items > 1) {
            ring_fetch(1);
            m_lim_1 = m_lim_new;
            n_lim_1 = n_lim_new;
        }
    }
    auto issue_tile = [&](int kt, unsigned char *stage) {
        const int k0 = kt * BK;
        static_for<0, SA::NI>([&](auto i_t) {
            sa.template issue<decltype(i_t)::value>((int64_t)k0 * 8 * p.ka_stride, stage);
        });
        static_for<0, SB::NI>([&](auto i_t) {
            sb.template issue<decltype(i_t)::value>((int64_t)k0 * 8 * p.kb_stride, stage + kABytes);
        });
    };
    issue_tile(0, smem);
    issue_tile(1, smem + kStageBytes);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");  // K-tile 0 has landed
    __syncthreads();

    double fa[2][TM], fb[2][TN];
    const unsigned smem_lds = lds_addr(smem), ring_lds = lds_addr(&s_ring[0]);
    // fragments of k-step KK of the stage at LDS address `stage` into slot S; asynchronous (frags_wait)
    auto read_frags = [&](auto slot_t, auto kk_t, unsigned stage) {
        constexpr int S = decltype(slot_t)::value, KK = decltype(kk_t)::value;
        static_for<0, TM>([&](auto b_t) {
            constexpr int b = decltype(b_t)::value;
            lds_read64<FragAddr<LA, BM>::template imm<b, KK>()>(fa[S][b], stage + fa_addr.template var<b, KK>());
        });
        static_for<0, TN>([&](auto b_t) {
            constexpr int b = decltype(b_t)::value;
            lds_read64<kABytes + FragAddr<LB, BN>::template imm<b, KK>()>(fb[S][b], stage + fb_addr.template var<b, KK>());
        });
    };
    auto frags_wait = [&](auto slot_t) {
        constexpr int S = decltype(slot_t)::value;
        static_assert(TM == 4 && TN == 2, "operand list of lds_wait");
        lds_wait(fa[S][0], fa[S][1], fa[S][2], fa[S][3], fb[S][0], fb[S][1]);
    };
    read_frags(std::integral_constant<int, 0>(), std::integral_constant<int, 0>(), smem_lds);

    int stage_cur = 0;  // LDS stage of the K-tile being multiplied

    // Stores of the finished item: element (8-byte) stores straight from the accumulator registers, a wave
    // instruction writing 128 contiguous bytes of four rows.  Per store the vector ALU computes one 64-bit
    // add (row offset, read from the ring a store ahead, onto the lane's column pointer); the 16-byte
    // form would cost four register moves, two lane exchanges and four selects per store, and a wave that
    // feeds the matrix pipe has about sixteen issue slots per MFMA.  The stores are always non-temporal: the
    // launcher sends only results far larger than the Infinity Cache here.
    const double alpha = p.alpha;
    constexpr int NSTORE = TM * 4 * TN;  // per wave and item
    constexpr int SPT = NSTORE / STORE_TILES, SPACING = NIDX / SPT;
    static_assert(NSTORE % STORE_TILES == 0 && NIDX % SPT == 0 && SPACING >= 2, "store schedule");
    // stores of a store tile that are issued before the tile's wait for its operands: in a tile whose stores
    // are unconditional (FULL) they are known to sit in the in-order memory counter behind the DMAs the wait
    // is for, so the wait may leave them outstanding as well instead of draining this tile's first DMAs
#ifdef CTC_SX_STORES_AHEAD
    constexpr int STORES_AHEAD = CTC_SX_STORES_AHEAD;
#else
    constexpr int STORES_AHEAD = ((KSTEPS - 1) * NMFMA - (SPACING > 2 ? 2 : 1) + SPACING - 1) / SPACING;
#endif
    char *cptr[TN];                      // C + column offset of this lane's column in block j, in bytes
    int64_t cvoff[TN];
    int64_t rvoff[2] = {0, 0};           // row offsets: the row being stored and the next one (in flight)
    unsigned row_lds = 0;                // LDS address of the previous item's C-row offset of this lane's first row
    // extents inside C (capped at the tile's) of items X-1 (the one being stored; 0: nothing to store), X,
    // X+1 and X+2: computed where the item's ring slot is requested and shifted along once per item
    int m_lim_prev = 0, n_lim_prev = 0, m_lim_q[3] = {0, 0, 0}, n_lim_q[3] = {0, 0, 0};
    // row R = i * 4 + r of the wave's 64: lane (lane >> 4) + 4 r + 16 i, so its ring entry sits at a
    // compile-time distance from the lane's first row
    auto row_read = [&](auto r_t) {
        constexpr int R = decltype(r_t)::value;
        lds_read64<8 * (16 * (R / 4) + 4 * (R % 4))>(rvoff[R & 1], row_lds);
    };
    // slot: LDS address of the previous item's [C rows | C columns] run
    auto load_cols = [&](unsigned slot) {
        static_assert(TN == 2, "two column offsets per lane");
        lds_read64<8 * BM>(cvoff[0], slot + 8 * (wn0 + frow));
        lds_read64<8 * (BM + 16)>(cvoff[1], slot + 8 * (wn0 + frow));
        row_lds = slot + 8 * (wm0 + (lane >> 4));
        row_read(std::integral_constant<int, 0>());
    };
    auto finish_cols = [&]() {
        lds_wait(cvoff[0], cvoff[1]);
        cptr[0] = reinterpret_cast<char *>(C + cvoff[0]);
        cptr[1] = reinterpret_cast<char *>(C + cvoff[1]);
    };
    // FULL: the item lies wholly inside C, no lane is masked
    auto store_one = [&](auto set_t, auto s_t, auto full_t) {
        constexpr int SET = decltype(set_t)::value, S = decltype(s_t)::value;
        constexpr bool FULL = decltype(full_t)::value;
        constexpr int R = S / TN, j = S % TN, i = R / 4, r = R % 4;
        if (j == 0) {
            // this row's offset was requested at the previous row's first store, one k-step ago; the only
            // LDS reads issued since are the TM + TN fragment reads that open this k-step (when the loader
            // switched items in between, its reads were waited for in full)
            if (R == 0) lds_wait(rvoff[0]);
            else lds_wait_but<TM + TN>(rvoff[R & 1]);
            if (R + 1 < TM * 4) row_read(std::integral_constant<int, R + 1>());
        }
        // one vector-ALU instruction per store: (row offset << 3) + column pointer
        // (typed as a global-memory pointer: a generic one would make the store a FLAT instruction, which
        // the compiler orders behind every LDS-DMA in flight)
        typedef __attribute__((address_space(1))) double *gptr_t;
        gptr_t dst;
        asm("v_lshl_add_u64 %0, %1, 3, %2" : "=v"(dst) : "v"(rvoff[R & 1]), "v"(cptr[j]));
        bool ok = true;
        if (!FULL) ok = (wm0 + 16 * i + Tr::acc_row(lane, r) < m_lim_prev) && (wn0 + 16 * j + frow < n_lim_prev);
#if defined(CTC_SX_NOSTORE)
        ok = ok && acc[SET][i][j][r] == 1.2345e300;
#endif
        double v = acc[SET][i][j][r];
        if (SCALE) v *= alpha;
#ifdef CTC_SX_PLAIN_STORES  // diagnostic build: without the streaming hint
        if (ok) *dst = v;
#else
        if (ok) __builtin_nontemporal_store(v, dst);
#endif
    };

#ifdef CTC_SX_STAMPS  // diagnostic build: shader-clock ticks per K-tile kind, split at the wait + barrier
    unsigned long long st_body[STORE_TILES + 1] = {}, st_sync[STORE_TILES + 1] = {}, st_n[STORE_TILES + 1] = {}, st_t0 = 0, st_t1 = 0, st_t2 = 0;
#define CTC_SX_STAMP(var)                                                        \
    do {                                                                         \
        __builtin_amdgcn_sched_barrier(0);                                       \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                       \
    } while (0)
#else
#define CTC_SX_STAMP(var) do { } while (0)
#endif
    auto tile_step = [&](auto set_t, auto var_t, auto full_t, int X, int kt) {
        constexpr int SET = decltype(set_t)::value, VAR = decltype(var_t)::value;
        constexpr bool FULL_T = decltype(full_t)::value;
        CTC_SX_STAMP(st_t0);
        const unsigned cur = smem_lds + stage_cur * kStageBytes;
        const int stage_next = stage_cur == kStages - 1 ? 0 : stage_cur + 1;
        const int stage_load = stage_next == kStages - 1 ? 0 : stage_next + 1;
        const unsigned nxt = smem_lds + stage_next * kStageBytes;
        unsigned char *ld = smem + stage_load * kStageBytes;
        stage_cur = stage_next;
        const bool has_next = X + 1 < n_items, has_next2 = X + 2 < n_items;
        int kt2 = kt + 2;
        if (kt2 >= nk) kt2 -= nk;
        const int k0_load = kt2 * BK;
        const int64_t kofs_a = (int64_t)k0_load * 8 * p.ka_stride, kofs_b = (int64_t)k0_load * 8 * p.kb_stride;
        const bool loader_switch = (kt == nk - 2) && has_next;
        const unsigned slot_prev = ring_lds + 8 * G_::c_entry(X + 3);   // C offsets of item X - 1
        const unsigned slot_next = ring_lds + 8 * G_::ab_entry(X + 1);  // A / B offsets of item X + 1
        const bool ring_tile = TWO ? (kt == nk - 1) : (VAR == 0);        // this tile requests item X + 2's offsets
        constexpr int RING_IDX = TWO ? NIDX - 4 : NIDX - NMFMA - 2;      // K-tile 16: behind the barrier
        static_for<0, KSTEPS>([&](auto kk_t) {
            constexpr int kk = decltype(kk_t)::value;
            constexpr int s = kk & 1;
            constexpr bool last = (kk == KSTEPS - 1);
            if (last) {
                // K-tile +1 has landed: the memory counter retires in order and this wave has issued at
                // least its NDMA operand DMAs since (stores and the ring DMA are issued only where they
                // apply, so they cannot be counted on: when they were issued, the oldest of them are waited
                // for as well, some 40 MFMAs after their issue).  Then the barrier: every wave's part of
                // K-tile +1 is in LDS, and every wave is done reading the stage the next DMAs overwrite.
                // A bare s_barrier: __syncthreads() carries a fence, and with LDS-DMA in flight the compiler
                // turns that fence into a full drain of the memory counter.  What the fence would order is
                // ordered by hand here: this wave's DMAs of K-tile +1 have landed (vmcnt), its LDS reads of
                // the stage about to be overwritten have returned (lgkmcnt).
                CTC_SX_STAMP(st_t1);
#ifdef CTC_SX_NOWAITV  // diagnostic builds (wrong results): tools/build_variant.sh
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#elif defined(CTC_SX_A_ONCE)
                // diagnostic build (wrong results): the A tile is fetched for a workgroup's first item only, as
                // if it stayed resident in LDS - what the kernel gains when two thirds of its L2 -> LDS bytes go
                if (X == 0) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NDMA + (FULL_T && VAR < STORE_TILES ? STORES_AHEAD : 0)) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(SB::NI + (FULL_T && VAR < STORE_TILES ? STORES_AHEAD : 0)) : "memory");
#else
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NDMA + (FULL_T && VAR < STORE_TILES ? STORES_AHEAD : 0)) : "memory");
#endif
#ifndef CTC_SX_NOBARRIER
                __builtin_amdgcn_s_barrier();
#endif
                CTC_SX_STAMP(st_t2);
            }
            frags_wait(std::integral_constant<int, s>());
            static_for<0, NMFMA>([&](auto q_t) {
                constexpr int q = decltype(q_t)::value;
                constexpr int idx = kk * NMFMA + q;
                if (VAR == 0 && kk == 0)
                    acc[SET][q / TN][q % TN] = Tr::mfma(fa[s][q / TN], fb[s][q % TN], zero_of<acc_t>());
                else
                    acc[SET][q / TN][q % TN] = Tr::mfma(fa[s][q / TN], fb[s][q % TN], acc[SET][q / TN][q % TN]);
                // fragments of the next k-step (of the next K-tile in the last k-step)
                if (q == 0) {
                    if (last) read_frags(std::integral_constant<int, s ^ 1>(), std::integral_constant<int, 0>(), nxt);
                    else read_frags(std::integral_constant<int, s ^ 1>(), std::integral_constant<int, (kk + 1) % KSTEPS>(), cur);
                }
                // the loader moves on to the next item's rows before this tile's first DMA
                if (idx == 1 && loader_switch) {
                    sa.load_rows_async(slot_next);
                    sb.load_rows_async(slot_next + 8 * BM);
                }
                if (idx == 2 && loader_switch) {
                    sa.finish_rows(A, true);
                    sb.finish_rows(B, true);
                }
                // operand DMAs of stream tile +2
#ifndef CTC_SX_NODMA
                static_for<0, NDMA>([&](auto u_t) {
                    constexpr int u = decltype(u_t)::value;
                    if (idx == DMA_FIRST + DMA_STEP * u) {
#ifdef CTC_SX_A_ONCE
                        if constexpr (u < SA::NI) { if (X == 0) sa.template issue<u>(kofs_a, ld); }
#else
                        if constexpr (u < SA::NI) sa.template issue<u>(kofs_a, ld);
#endif
                        else sb.template issue<u - SA::NI>(kofs_b, ld + kABytes);
                    }
                });
#endif
                // item bookkeeping
                if (VAR == 0 && idx == 0) load_cols(slot_prev);
                if (VAR == 0 && idx == 1) finish_cols();
                if (idx == RING_IDX && ring_tile && has_next2) ring_fetch(X + 2);
                // stores of the previous item, one every SPACING MFMAs
                if (VAR < STORE_TILES) {
                    static_for<0, SPT>([&](auto u_t) {
                        constexpr int u = decltype(u_t)::value;
                        constexpr int S = (VAR < STORE_TILES ? VAR : 0) * SPT + u;
                        if (idx == u * SPACING + (SPACING > 2 ? 2 : 1))
                            store_one(std::integral_constant<int, SET ^ 1>(), std::integral_constant<int, S>(), full_t);
                    });
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });
#ifdef CTC_SX_STAMPS
        {
            unsigned long long st_t3;
            CTC_SX_STAMP(st_t3);
            st_body[VAR] += (st_t1 - st_t0) + (st_t3 - st_t2);
            st_sync[VAR] += st_t2 - st_t1;
            st_n[VAR] += 1;
        }
#endif
    };

    m_lim_q[0] = m_lim_0; n_lim_q[0] = n_lim_0;
    m_lim_q[1] = m_lim_1; n_lim_q[1] = n_lim_1;
    auto run_item = [&](auto set_t, int X) {
        m_lim_new = n_lim_new = 0;  // stays 0 when there is no item X+2
        // the previous item lies wholly inside C (all but the last tile row / column and "item -1"): its
        // stores carry no per-lane bounds and no branches
        const bool prev_full = m_lim_prev == BM && n_lim_prev == BN;
        if (prev_full) {
            static_for<0, STORE_TILES>([&](auto v_t) { tile_step(set_t, v_t, std::true_type(), X, decltype(v_t)::value); });
        } else {
            static_for<0, STORE_TILES>([&](auto v_t) { tile_step(set_t, v_t, std::false_type(), X, decltype(v_t)::value); });
        }
        for (int kt = STORE_TILES; kt < nk; ++kt)
            tile_step(set_t, std::integral_constant<int, STORE_TILES>(), std::true_type(), X, kt);
        m_lim_q[2] = m_lim_new;  // of item X + 2, whose offsets one of these tiles requested
        n_lim_q[2] = n_lim_new;
        // this item is the "previous item" of the next one
        m_lim_prev = m_lim_q[0]; n_lim_prev = n_lim_q[0];
        m_lim_q[0] = m_lim_q[1]; n_lim_q[0] = n_lim_q[1];
        m_lim_q[1] = m_lim_q[2]; n_lim_q[1] = n_lim_q[2];
    };

    for (int X = 0; X < n_items; X += 2) {
        run_item(std::integral_constant<int, 0>(), X);
        if (X + 1 < n_items) run_item(std::integral_constant<int, 1>(), X + 1);
    }

    // the last item's stores: nothing left to hide them behind
    auto flush = [&](auto set_t, const int64_t *slot) {  // slot: the item's [C rows | C columns] run
        constexpr int SET = decltype(set_t)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm0 + 16 * i + Tr::acc_row(lane, r);
                const int64_t ro = slot[row];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const int col = wn0 + 16 * j + frow;
                    if (row < m_lim_prev && col < n_lim_prev) {
#ifdef CTC_SX_PLAIN_STORES
                        C[ro + slot[BM + col]] = SCALE ? acc[SET][i][j][r] * alpha : acc[SET][i][j][r];
#else
                        __builtin_nontemporal_store(SCALE ? acc[SET][i][j][r] * alpha : acc[SET][i][j][r], C + ro + slot[BM + col]);
#endif
                    }
                }
            }
    };
#ifdef CTC_SX_STAMPS
    if (t == 0 && (bid == 0 || bid == 77))
        printf("[stamps] block %d items %d | per K-tile ticks body/sync: first %llu/%llu second %llu/%llu later %llu/%llu (n %llu %llu %llu)\n", bid,
               n_items, st_body[0] / (st_n[0] ? st_n[0] : 1), st_sync[0] / (st_n[0] ? st_n[0] : 1), st_body[1] / (st_n[1] ? st_n[1] : 1),
               st_sync[1] / (st_n[1] ? st_n[1] : 1), st_body[STORE_TILES] / (st_n[STORE_TILES] ? st_n[STORE_TILES] : 1),
               st_sync[STORE_TILES] / (st_n[STORE_TILES] ? st_n[STORE_TILES] : 1), st_n[0], st_n[1], st_n[STORE_TILES]);
#endif
#ifdef CTC_SX_CLOCK
    if (t == 0 && (bid == 0 || bid == 77)) {
        const unsigned long long dc = __builtin_readcyclecounter() - ck_core0, dr = __builtin_amdgcn_s_memrealtime() - ck_real0;
        printf("[clock] block %d: %llu core ticks in %llu x 10 ns -> %.3f GHz\n", bid, dc, dr, (double)dc / (10.0 * (double)dr));
    }
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // DMAs past the end of the stream have landed before the LDS goes away
    const int64_t *slot_last = &s_ring[G_::c_entry(n_items - 1)];
    if ((n_items - 1) & 1) flush(std::integral_constant<int, 1>(), slot_last);
    else flush(std::integral_constant<int, 0>(), slot_last);
}

// workgroups of the current device that are resident at once, per kernel instantiation
template <int LA, int LB, int BK, bool SCALE> hipError_t resident_workgroups(int *out)
{
    static std::atomic<int> cache[64];  // 0: not queried yet
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64) return hipErrorInvalidDevice;
    if (cache[dev].load() == 0) {
        int cus = 0, per_cu = 0;
        e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess) return e;
        e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemm_stream_f64_kernel<LA, LB, BK, SCALE>, 256, 0);
        if (e != hipSuccess) return e;
        if (per_cu < 1) per_cu = 1;
        if (per_cu > 2) per_cu = 2;
        cache[dev].store(cus * per_cu);
    }
    *out = cache[dev].load();
    return hipSuccess;
}

}  // namespace

template <int LA, int LB, int BK, bool SCALE> hipError_t stream_f64::launch_scaled(const GemmParams &p, hipStream_t stream)
{
    int slots = 0;
    hipError_t e = resident_workgroups<LA, LB, BK, SCALE>(&slots);
    if (e != hipSuccess) return e;
    GemmParams q = p;
    q.nbm = (int)((p.M + kBM - 1) / kBM);
    q.nbn = (int)((p.N + kBN - 1) / kBN);
    if (q.group < 1) q.group = 1;
    const int64_t tiles = (int64_t)q.nbm * q.nbn;
    if (tiles > 0x7FFFFFFF) return hipErrorInvalidValue;
    const int grid = (int)(tiles < slots ? tiles : slots);
    hipLaunchKernelGGL((gemm_stream_f64_kernel<LA, LB, BK, SCALE>), dim3((unsigned)grid), dim3(256), 0, stream, q);
    return hipGetLastError();
}

// one explicit instantiation per part (all of them when the file is compiled as one translation unit)
#define CTC_SX_INSTANTIATE(LA_, LB_, BK_, SC_) \
    template hipError_t stream_f64::launch_scaled<LA_, LB_, BK_, SC_>(const GemmParams &, hipStream_t);
#if CTC_SX_PART < 0 || CTC_SX_PART == 1
CTC_SX_INSTANTIATE(1, 1, 16, false)
#endif
#if CTC_SX_PART < 0 || CTC_SX_PART == 2
CTC_SX_INSTANTIATE(1, 1, 16, true)
#endif
#ifndef CTC_SX_ONE  // diagnostic builds of the outer/outer, K-tile 16 instantiation only (tools/build_variant.sh)
#if CTC_SX_PART < 0 || CTC_SX_PART == 3
CTC_SX_INSTANTIATE(1, 1, 32, false)
#endif
#if CTC_SX_PART < 0 || CTC_SX_PART == 4
CTC_SX_INSTANTIATE(1, 1, 32, true)
#endif
#if CTC_SX_PART < 0 || CTC_SX_PART == 5
CTC_SX_INSTANTIATE(0, 0, 32, false)
#endif
#if CTC_SX_PART < 0 || CTC_SX_PART == 6
CTC_SX_INSTANTIATE(0, 0, 32, true)
#endif
#if CTC_SX_PART < 0 || CTC_SX_PART == 7
CTC_SX_INSTANTIATE(0, 1, 32, false)
#endif
#if CTC_SX_PART < 0 || CTC_SX_PART == 8
CTC_SX_INSTANTIATE(0, 1, 32, true)
#endif
#if CTC_SX_PART < 0 || CTC_SX_PART == 9
CTC_SX_INSTANTIATE(1, 0, 32, false)
#endif
#if CTC_SX_PART < 0 || CTC_SX_PART == 10
CTC_SX_INSTANTIATE(1, 0, 32, true)
#endif
#endif
#undef CTC_SX_INSTANTIATE
#endif  // CTC_SX_PART != 0

#if CTC_SX_PART <= 0
// diagnostic switch: CTC_TC_STREAM_BK=32 keeps the one-workgroup-per-CU geometry for every layout
static int stream_bk()
{
    static const int bk = getenv("CTC_TC_STREAM_BK") ? atoi(getenv("CTC_TC_STREAM_BK")) : 16;
    return bk == 32 ? 32 : 16;
}

bool stream_gemm_f64_dma_supported(int layout_a, int layout_b, const GemmParams &p)
{
    // K: whole K-tiles, and at least the 64 contracted elements whose K-tiles carry the previous item's stores
    const int bk = (layout_a == 1 && layout_b == 1) ? stream_bk() : 32;
    return (layout_a == 0 || layout_a == 1) && (layout_b == 0 || layout_b == 1) && p.batch == 1 && p.beta == 0.0 &&
           p.K >= 64 && p.K % bk == 0 && p.M >= 2 && p.N >= 2 && p.M % 2 == 0 && p.N % 2 == 0 &&
           p.ka_stride > 0 && p.kb_stride > 0;
}

namespace {
template <int LA, int LB, int BK> hipError_t launch_one(const GemmParams &p, hipStream_t stream)
{
    return p.alpha == 1.0 ? stream_f64::launch_scaled<LA, LB, BK, false>(p, stream)
                          : stream_f64::launch_scaled<LA, LB, BK, true>(p, stream);
}
}  // namespace

hipError_t launch_gemm_stream_f64_dma(int layout_a, int layout_b, const GemmParams &p, hipStream_t stream)
{
    if (!stream_gemm_f64_dma_supported(layout_a, layout_b, p)) return hipErrorInvalidValue;
    switch (layout_a * 2 + layout_b) {
#ifndef CTC_SX_ONE
    case 0: return launch_one<0, 0, 32>(p, stream);
    case 1: return launch_one<0, 1, 32>(p, stream);
    case 2: return launch_one<1, 0, 32>(p, stream);
    case 3: return stream_bk() == 32 ? launch_one<1, 1, 32>(p, stream) : launch_one<1, 1, 16>(p, stream);
#else
    case 3: return launch_one<1, 1, 16>(p, stream);
#endif
    default: return hipErrorInvalidValue;
    }
}
#endif  // CTC_SX_PART <= 0

}  // namespace ctc
