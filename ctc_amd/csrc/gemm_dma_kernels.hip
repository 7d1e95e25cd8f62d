// fp64 / fp32 MFMA GEMM, one 128 x 128 tile of C per workgroup as in gemm_kernels.hip, with the operands
// staged global -> LDS by LDS-DMA (global_load_lds_dwordx4) instead of through registers:
//   - the waves hold no staging registers and execute no LDS writes; per K-tile a wave issues four DMA
//     instructions whose addresses are one 32-bit lane offset (fixed for the whole tile of C) on top of a
//     wave-uniform 64-bit base, so staging costs the vector ALU nothing.  On MI355X the fp64 MFMA shares the
//     vector ALU, and the register-staged kernel spends about three vector instructions per MFMA there;
//   - K-tiles of 64 bytes a row (8 fp64 / 16 fp32 contracted elements: two / four k-steps of MFMAs) in a
//     ring of four 16 KB stages (64 KB: two workgroups per CU): during K-tile t the DMA of
//     K-tile t + 3 is issued, the one of t + 1 is awaited (counted s_waitcnt: the memory counter retires in
//     order) just before the single barrier of the tile;
//   - a DMA instruction writes 1 KB of consecutive LDS (16 bytes per lane), so bank conflicts are removed by
//     choosing WHICH 16-byte chunk a lane fetches instead of by padding:
//       K-major operand (unit stride along K):   image [row][64 bytes of k], chunk c of row r stored at
//           chunk position c ^ ((r >> 2) & 3);
//       outer-major operand (unit stride along the rows): image [k][128 rows]; fp64: chunk c (rows 2c, 2c + 1)
//           of line k stored at position c ^ ((k & 1) << 3); fp32: chunk c (rows 4c .. 4c + 3) at c ^ ((k & 3) << 2);
//   - fragment reads are inline-assembly ds_read_b64 / ds_read_b32 with the stage, the row block and the
//     k-step in the instruction's offset field (the K-tile loop is unrolled over the four stages), two (fp64)
//     or four (fp32) address registers per operand.
// Needs: both operands in a 16-byte-vector layout, K a multiple of the K-tile, k offsets that are affine
// inside every aligned group of that many (GemmParams::k8_affine / k16_affine; e.g. every contracted mode's
// extent a multiple of it, or one contracted mode), row / column offsets below 2^28 elements.  Runs of affine
// k offsets that are no multiple of the K-tile (GemmParams::k_run) are served by ragged K-tiles: the last
// tile of a run is the run's last K-tile-many elements, and the elements it shares with its predecessor are
// zeroed in the fragments.  Everything else takes gemm_kernels.hip.
// Tile order, split of the last round along K, epilogue: as there.  A launch may carry fewer workgroups
// than items (fp64 with many rounds: one per slot); a workgroup then walks items b, b + gridDim.x, ...
#include "gemm_device.hpp"
#include "kernels.hpp"
#include "lds_dma.hpp"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace ctc {

namespace {

using namespace gemm_detail;
using namespace lds_dma;

constexpr int kBM = 128, kBN = 128, kStages = 4;
// RB: bytes of k per row of a K-tile.  64 (8 fp64 / 16 fp32 contracted elements) is the geometry described
// above; 32 (images of 4 fp64 / 8 fp32 elements: one / two k-steps, one DMA instruction per wave and image)
// is for contractions whose k tables are affine in groups of that many only, with one or two images per
// operand and stage (NSUB of the kernel).
template <typename T, int RB> constexpr int tile_k() { return RB / (int)sizeof(T); }
template <int RB> constexpr int op_bytes() { return 128 * RB; }         // one operand's tile image
template <int RB> constexpr int stage_bytes() { return 2 * op_bytes<RB>(); }  // A image, then B image

template <int OFF> __device__ __forceinline__ void lds_read(double &d, unsigned addr) { lds_read64<OFF>(d, addr); }
template <int OFF> __device__ __forceinline__ void lds_read(float &d, unsigned addr)
{
    static_assert(OFF >= 0 && OFF < 65536, "ds_read offset field");
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
template <typename T> __device__ __forceinline__ void lds_wait8(T &a0, T &a1, T &a2, T &a3, T &b0, T &b1, T &b2, T &b3)
{
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));
}

// Stream-K shares (GemmParams::sk_units, sk_group): the launch's last tiles are taken g = sk_group at a time
// (consecutive tile ids: neighbours that share operand panels); the W = groups x nk K-tile iterations of these
// groups of tiles, in order, are cut into units / g runs, and the g workgroups of run u work run u off together,
// workgroup j on tile j of the group the run is passing through: run u is [sk_unit_start(u), sk_unit_start(u + 1)).  A cut that would
// fall within kSkSnap K-tiles of a tile's first or last iteration is moved onto the tile boundary (no pieces
// of a few K-tiles, each of which would cost a prologue, a slab and a share of the fix-up pass).  Kernel,
// fix-up kernel and host call this one function.
constexpr int kSkSnap = 8;
__host__ __device__ inline int64_t sk_unit_start(int64_t u, int64_t W, int64_t units, int64_t nk)
{
    int64_t s = u * W / units;
    const int64_t r = s % nk;
    if (r < kSkSnap) s -= r;
    else if (nk - r < kSkSnap) s += nk - r;
    return s;
}

// What a lane fetches for its NI DMA instructions per K-tile of one operand, and where a lane's MFMA
// fragments sit in the operand's image.
template <typename T, int LAY, int RB> struct Operand {
    static constexpr bool KMAJ = (LAY != kLayOuterVec);
    static constexpr int ES = (int)sizeof(T), V = 16 / ES;  // element size, elements per 16-byte chunk
    static constexpr int BK = RB / ES, KSTEPS = BK / 4;
    static constexpr int NI = op_bytes<RB>() / 1024 / 4;    // DMA instructions per wave and K-tile
    static constexpr int CPR = RB / 16;                     // K-major: chunks per row
    static constexpr int RPI = 1024 / RB;                   // K-major: rows per instruction
    static constexpr int NFRAG = (ES == 8) ? 2 : 4;
    unsigned off[NI];      // byte offset of the lane's 16 bytes from the wave-uniform base, per instruction
    unsigned frag[NFRAG];  // LDS byte addresses (stage 0): K-major: of k-step kk, row block 0; outer-major: of
                           // row block b (fp64: b & 1 suffices), k-step 0

    // tab: the operand's row (column) offsets, r0: first row of the tile, R: the operand's extent, kin: its
    // step along k inside a K-tile; instruction j of wave `wave` fills bytes [(NI wave + j) KB, + 1 KB) of the image
    __device__ __forceinline__ void init_dma(const int64_t *__restrict__ tab, int64_t r0, int64_t R, int64_t kin, int lane, int wave)
    {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int inst = NI * wave + j;
            if (KMAJ) {
                // RPI rows per instruction, CPR chunks per row; chunk c of row r sits at position
                // c ^ ((r / rows per 256 bytes) & (CPR - 1))
                const int row = RPI * inst + lane / CPR;
                const int c = (lane % CPR) ^ ((lane >> 4) & (CPR - 1));
                int64_t g = r0 + row;
                if (g > R - 1) g = R - 1;  // rows past the extent repeat its last row: products never stored
                off[j] = (unsigned)(tab[g] * ES + c * 16);
            } else if (ES == 8) {
                // one line per instruction (k = inst), 64 chunks of two rows
                const int c = lane ^ ((inst & 1) << 3);
                int64_t g = r0 + 2 * c;
                if (g > R - 2) g = R - 2;  // R is even: a pair never straddles the end
                off[j] = (unsigned)(tab[g] * 8);
            } else {
                // two lines per instruction (k = 2 inst + (lane >> 5)), 32 chunks of four rows each
                const int kq = (2 * inst + (lane >> 5)) & 3;  // k & 3
                const int c = (lane & 31) ^ (kq << 2);
                int64_t g = r0 + 4 * c;
                if (g > R - 4) g = R - 4;  // R is a multiple of 4
                off[j] = (unsigned)(tab[g] * 4 + (lane >> 5) * kin * 4);
            }
        }
    }
    // w0: first row of the wave's part of the tile; image: LDS address of the operand's image in stage 0
    __device__ __forceinline__ void init_frag(unsigned image, int frow, int fk, int w0)
    {
        if (KMAJ && ES == 8 && RB == 64) {
            const int r2 = (frow >> 2) & 3;
            const int phys0 = ((fk >> 1) ^ (r2 & 1)) | ((r2 >> 1) << 1);
            const unsigned base = image + (w0 + frow) * 64 + phys0 * 16 + (fk & 1) * 8;
            frag[0] = base;
            frag[1] = base ^ 32;
        } else if (KMAJ && ES == 8) {
            // 32-byte rows: chunk fk >> 1 of the row, at position (fk >> 1) ^ ((row >> 3) & 1); one k-step
            const int sw = (frow >> 3) & 1;
            frag[0] = image + (w0 + frow) * 32 + (((fk >> 1) ^ sw) << 4) + (fk & 1) * 8;
            frag[1] = frag[0];
        } else if (KMAJ) {
            // chunk kk of the row holds k = 4 kk .. 4 kk + 3
            const int sw = (frow / (256 / RB)) & (CPR - 1);
            const unsigned base = image + (w0 + frow) * RB + sw * 16 + fk * 4;
#pragma unroll
            for (int kk = 0; kk < KSTEPS; ++kk) frag[kk] = base ^ (kk << 4);
        } else if (ES == 8) {
            const int p = fk & 1;
            const unsigned base = image + fk * 1024 + ((w0 + frow) >> 1) * 16 + (frow & 1) * 8;
            frag[0] = base + 128 * p;
            frag[1] = base - 128 * p;
        } else {
            const unsigned base = image + fk * 512 + ((w0 >> 2) + (frow >> 2)) * 16 + (frow & 3) * 4;
#pragma unroll
            for (int b = 0; b < 4; ++b) frag[b] = base + 64 * (b ^ fk);
        }
    }
    template <int b, int kk> __device__ __forceinline__ unsigned frag_reg() const
    {
        return frag[KMAJ ? kk : (ES == 8 ? (b & 1) : b)];
    }
    template <int b, int kk> static constexpr int frag_imm()
    {
        return KMAJ ? b * 16 * RB : (ES == 8 ? 128 * b + kk * 4096 : kk * 2048);
    }
    // wave-uniform byte offset of instruction j of the K-tile whose first k offset (elements) is kofs
    __device__ __forceinline__ static int64_t uniform(int64_t kofs, int64_t kin, int wave, int j)
    {
        return KMAJ ? kofs * ES : (kofs + (int64_t)((NI * wave + j) * (ES == 8 ? 1 : 2)) * kin) * ES;
    }
};

// NSUB: images per operand and stage.  2 (with RB = 32) puts two K-tiles of 32-byte rows, each with its own k
// offset, into one stage: the K loop then has the shape of the 64-byte geometry (one barrier per 32 MFMAs per
// wave) for k tables that are affine in groups of half a K-tile only.
#ifdef CTC_DMA_ROUND_SYNC
__device__ unsigned g_round_sync[64][8];
#endif

template <typename T, int LA, int LB, int RB, int NSUB, bool RAGGED>
__global__ void __launch_bounds__(256, 2) gemm_dma_kernel(const GemmParams p)
{
    static_assert(NSUB == 1 || (NSUB == 2 && RB == 32), "two sub-tiles only with 32-byte rows");
    static_assert(!RAGGED || (NSUB == 1 && RB == 64), "ragged K-tiles exist for the 64-byte geometry");
    constexpr int BM = kBM, BN = kBN, BKS = tile_k<T, RB>(), BK = NSUB * BKS, WN = 2;
    constexpr int kOpBytes = op_bytes<RB>(), kOperandBytes = NSUB * kOpBytes, kStageBytes = 2 * kOperandBytes;
    constexpr int TM = 4, TN = 4, NMFMA = TM * TN, KSTEPS = BK / 4, KS1 = BKS / 4;  // k-steps per stage / per image
    static_assert(KSTEPS == 1 || KSTEPS % 2 == 0, "fragment slots alternate per k-step (per K-tile when it has one k-step)");
    typedef Traits<T> Tr;
    typedef typename Tr::acc_t acc_t;
    typedef Operand<T, LA, RB> OA;
    typedef Operand<T, LB, RB> OB;
    constexpr int NI = OA::NI, NDMA = 2 * NSUB * NI;  // DMA instructions per wave per stage

    __shared__ __attribute__((aligned(1024))) unsigned char smem[kStages * kStageBytes];
    __shared__ int64_t s_coff[BM + BN];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm0 = (wave / WN) * (TM * 16);
    const int wn0 = (wave % WN) * (TN * 16);
    const int frow = lane & 15;
    const int fk = lane >> 4;
#ifdef CTC_CLOCK  // diagnostic build: shader clock while this workgroup runs = core-clock ticks / 100 MHz ticks
    const unsigned long long ck_core0 = __builtin_readcyclecounter(), ck_real0 = __builtin_amdgcn_s_memrealtime();
#endif

    // RAGGED: the k tables are affine inside runs of p.k_run >= BK contracted elements only, and a run is no
    // multiple of the K-tile.  A run is then cut into ceil(run / BK) tiles whose last one is the LAST BK elements
    // of the run: it overlaps its predecessor by `skip_tail` elements, which are zeroed in the fragments of both
    // operands (in registers, behind the wait that delivers them; k-steps that are overlap entirely are not
    // multiplied at all), so every tile is one affine window of BK elements inside the operands.
    const int run = RAGGED ? (int)p.k_run : BK;
    const int tiles_per_run = (run + BK - 1) / BK;
    const int skip_tail = tiles_per_run * BK - run;
    const int nk_total = RAGGED ? (int)(p.K / run) * tiles_per_run : (int)(p.K / BK);

    // A launch may have fewer workgroups than items: workgroup b then takes items b, b + gridDim.x, ...
    // (gridDim.x a multiple of 8, so an item's XCD is the workgroup's).  Items: the tiles that run whole, then
    // either the K chunks of the remaining tiles (p.split of equal length per tile) or p.sk_units shares of
    // the remaining tiles' K-tile iterations (a share is worked off piece by piece: the rest of a tile,
    // whole tiles, the beginning of a tile).
    const int tiles_per_entry = p.nbm * p.nbn;
    const int tail_tiles = tiles_per_entry * (p.flat_batch > 0 ? p.flat_batch : 1) - p.dp_tiles;
    const int n_items = p.dp_tiles + (p.sk_units > 0 ? p.sk_units : (p.split > 1 ? tail_tiles * p.split : 0));
    const int sk_runs = p.sk_units / p.sk_group;  // runs of K-tile iterations, each worked off by sk_group workgroups
    const int64_t sk_total = (int64_t)((tail_tiles + p.sk_group - 1) / p.sk_group) * nk_total;
    int64_t sk_it = 0, sk_end = 0;  // what is left of the share being worked off
    int sk_unit = 0;
    for (int bid = blockIdx.x;;) {
    // item -> tile, as in gemm_kernel: XCD-contiguous runs of tile ids, groups of tile rows
    int sid, kt_first = 0, kt_last = nk_total;
    int64_t slab = 0;
    bool partial = false;  // raw sums to slab `slab` of the workspace instead of C
    {
        auto xcd_run = [](int item, int nblk) {  // XCD-contiguous renumbering of nblk items
            const int xcd = item & 7, q = nblk >> 3, r = nblk & 7;
            return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (item >> 3);
        };
        bool from_share = sk_it < sk_end;
        if (!from_share) {
            if (bid >= n_items) break;
            const int item = bid;
            bid += gridDim.x;
            if (item < p.dp_tiles) {
                sid = p.plain_order ? item : xcd_run(item, p.dp_tiles);
            } else if (p.sk_units > 0) {
                sk_unit = xcd_run(item - p.dp_tiles, p.sk_units);
                sk_it = sk_unit_start(sk_unit / p.sk_group, sk_total, sk_runs, nk_total);
                sk_end = sk_unit_start(sk_unit / p.sk_group + 1, sk_total, sk_runs, nk_total);
                if (sk_it >= sk_end) continue;
                from_share = true;
            } else {
                const int chunk_item = item - p.dp_tiles, chunk = chunk_item % p.split;
                sid = p.dp_tiles + chunk_item / p.split;
                kt_first = (int)((int64_t)chunk * nk_total / p.split);
                kt_last = (int)((int64_t)(chunk + 1) * nk_total / p.split);
                slab = chunk_item;
                partial = true;
            }
        }
        if (from_share) {
            // the next piece: from sk_it to the end of its tile or of the share.  A piece that starts a tile
            // goes to C like a whole tile (alpha, beta); one that starts inside a tile - only a share's first
            // piece can - leaves its raw sums in the share's slab, and the fix-up pass adds them to C.
            const int tile_rel = (int)(sk_it / nk_total) * p.sk_group + sk_unit % p.sk_group;
            sid = p.dp_tiles + tile_rel;
            kt_first = (int)(sk_it % nk_total);
            kt_last = (int)(sk_end - sk_it < (int64_t)(nk_total - kt_first) ? kt_first + (sk_end - sk_it) : nk_total);
            partial = kt_first != 0;
            slab = sk_unit;
            sk_it += kt_last - kt_first;
            if (tile_rel >= tail_tiles) continue;  // the last group of tiles is not full: nothing for this workgroup there
        }
    }
    // a launch without a batch dimension in its grid: the tile id counts through the batch entries
    int bz = blockIdx.y;
    if (p.flat_batch > 0) {
        bz = sid / tiles_per_entry;
        sid -= bz * tiles_per_entry;
    }
    const int width = p.group * p.nbn;
    const int first_m = (sid / width) * p.group;
    const int gsz = min(p.nbm - first_m, p.group);
    const int tm = first_m + (sid % width) % gsz;
    const int tn = (sid % width) / gsz;
    const int64_t m0 = (int64_t)tm * BM;
    const int64_t n0 = (int64_t)tn * BN;

    const char *A = reinterpret_cast<const char *>(static_cast<const T *>(p.A) + p.bat_a[bz]);
    const char *B = reinterpret_cast<const char *>(static_cast<const T *>(p.B) + p.bat_b[bz]);
    T *C = static_cast<T *>(p.C) + p.bat_c[bz];

    const int nk = kt_last - kt_first;
    // RAGGED: run and tile inside the run of the chunk's first tile, of the tile requested next, and the tile
    // inside its run of the tile multiplied next
    const int run_first = RAGGED ? kt_first / tiles_per_run : 0, tile_first = RAGGED ? kt_first % tiles_per_run : 0;
    int rq_run = run_first, rq_tile = tile_first, mm_tile = tile_first;

    // everything that goes through ordinary loads and LDS stores comes first: once a DMA is in flight the
    // compiler drains it in front of every LDS access it can see
    if (!partial) {
        for (int idx = t; idx < BM + BN; idx += 256) {
            const bool is_row = idx < BM;
            const int64_t g = is_row ? m0 + idx : n0 + (idx - BM);
            const bool ok = g < (is_row ? p.M : p.N);
            const int64_t off = (is_row ? p.row_c : p.col_c)[ok ? g : 0];
            s_coff[idx] = ok ? off : kNoOffset;
        }
    }
    OA oa;
    OB ob;
    oa.init_dma(p.row_a, m0, p.M, p.ka_in, lane, wave);
    ob.init_dma(p.col_b, n0, p.N, p.kb_in, lane, wave);
    const unsigned smem_lds = lds_addr(smem);
    oa.init_frag(smem_lds, frow, fk, wm0);
    ob.init_frag(smem_lds + kOperandBytes, frow, fk, wn0);

    acc_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = zero_of<acc_t>();

    // K-tiles are requested in order, three ahead; past the chunk's end the requests wrap to its first
    // tiles (finite data nobody multiplies) so that every wave issues the same number of DMAs per tile
    // Its k offsets come by SCALAR loads issued from assembly (the compiler would take vector loads here,
    // whose first use drains every DMA in flight); asynchronous: k_wait() before the first use.
    int tl = 0;  // tile the next request is for, relative to kt_first
    int64_t ka_next[NSUB], kb_next[NSUB];  // one k offset per image
    auto k_fetch = [&]() {
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) {
            const int64_t k0 = RAGGED ? (int64_t)rq_run * run + min(rq_tile * BK, run - BK) : (int64_t)(kt_first + tl) * BK + sub * BKS;
            const int64_t *pa = p.k_a + k0, *pb = p.k_b + k0;
            asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=&s"(ka_next[sub]) : "s"(pa));
            asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=&s"(kb_next[sub]) : "s"(pb));
        }
    };
    auto k_wait = [&]() {
#pragma unroll
        for (int sub = 0; sub < NSUB; ++sub) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ka_next[sub]), "+s"(kb_next[sub]));
    };
    auto advance = [&]() {
        tl = (tl + 1 == nk) ? 0 : tl + 1;
        if (RAGGED) {
            if (tl == 0) {
                rq_run = run_first;
                rq_tile = tile_first;
            } else if (++rq_tile == tiles_per_run) {
                rq_tile = 0;
                ++rq_run;
            }
        }
        k_fetch();
    };
    k_fetch();
    k_wait();
    // instruction u (A's images one after the other, then B's) of the requested tile into stage `stage`
    auto issue = [&](int u, unsigned stage, const int64_t (&ka)[NSUB], const int64_t (&kb)[NSUB]) {  // stage: LDS address
        const int opnd = u / (NSUB * NI), sub = (u % (NSUB * NI)) / NI, j = u % NI;
        const unsigned dst = stage + opnd * kOperandBytes + sub * kOpBytes + (NI * wave + j) * 1024;
        if (opnd == 0) dma16_split(A + OA::uniform(ka[sub], p.ka_in, wave, j), oa.off[j], dst);
        else dma16_split(B + OB::uniform(kb[sub], p.kb_in, wave, j), ob.off[j], dst);
    };
    __syncthreads();  // s_coff is written; no DMA is in flight yet
#pragma unroll
    for (int s = 0; s < kStages - 1; ++s) {
#pragma unroll
        for (int u = 0; u < NDMA; ++u) issue(u, smem_lds + s * kStageBytes, ka_next, kb_next);
        advance();
        k_wait();
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * NDMA) : "memory");  // K-tile 0 has landed
    __builtin_amdgcn_s_barrier();

    T fa[2][TM], fb[2][TN];
    // fragments of k-step KK of stage S into slot SLOT; asynchronous (frags_wait)
    auto read_frags = [&](auto slot_t, auto kk_t, auto stage_t) {
        constexpr int SLOT = decltype(slot_t)::value, KK = decltype(kk_t)::value, S = decltype(stage_t)::value;
        constexpr int SUB = KK / KS1, K1 = KK % KS1;  // image, k-step inside it
        static_for<0, TM>([&](auto b_t) {
            constexpr int b = decltype(b_t)::value;
            lds_read<S * kStageBytes + SUB * kOpBytes + OA::template frag_imm<b, K1>()>(fa[SLOT][b], oa.template frag_reg<b, K1>());
        });
        static_for<0, TN>([&](auto b_t) {
            constexpr int b = decltype(b_t)::value;
            lds_read<S * kStageBytes + SUB * kOpBytes + OB::template frag_imm<b, K1>()>(fb[SLOT][b], ob.template frag_reg<b, K1>());
        });
    };
    auto frags_wait = [&](auto slot_t) {
        constexpr int SLOT = decltype(slot_t)::value;
        static_assert(TM == 4 && TN == 4, "operand list of lds_wait");
        lds_wait8(fa[SLOT][0], fa[SLOT][1], fa[SLOT][2], fa[SLOT][3], fb[SLOT][0], fb[SLOT][1], fb[SLOT][2], fb[SLOT][3]);
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    read_frags(I0(), I0(), I0());

    // One K-tile out of stage S.  Every instruction that is not an MFMA is dealt out behind one, in an order
    // pinned with sched_barrier:
    //   every k-step: the fragment reads of the next one (of K-tile +1's first in the last);
    //   k-step 0: the four DMAs of K-tile +3 into stage S + 3;
    //   before the last k-step: wait (K-tile +1 has landed: the eight DMAs issued since may still be in
    //   flight), barrier (every wave's part of it is there, and every wave is done reading stage S, which
    //   the next tile's DMAs overwrite);
    //   last k-step: the k offsets of the tile requested next.
    auto tile_step = [&](auto stage_t) {
        constexpr int S = decltype(stage_t)::value, SN = (S + 1) % kStages, SL = (S + kStages - 1) % kStages;
        const unsigned ld = smem_lds + SL * kStageBytes;
        int64_t ka[NSUB], kb[NSUB];
        // RAGGED: leading elements of this tile that its predecessor has multiplied already
        const int skip = (RAGGED && mm_tile == tiles_per_run - 1) ? skip_tail : 0;
        if (RAGGED) mm_tile = (mm_tile + 1 == tiles_per_run) ? 0 : mm_tile + 1;
        static_for<0, KSTEPS>([&](auto kk_t) {
            // with one k-step per K-tile the fragment slots alternate per tile (the stages' parity)
            constexpr int kk = decltype(kk_t)::value, slot = (KSTEPS == 1 ? S : kk) & 1;
            constexpr bool last = (kk == KSTEPS - 1);
            if (last) {
                // a bare s_barrier: __syncthreads() carries a fence, which with LDS-DMA in flight becomes a
                // full drain of the memory counter.  What the fence would order is ordered by hand: this
                // wave's DMAs of K-tile +1 have landed (vmcnt: the DMAs issued since may be in flight - those
                // of this tile and the one before, or only the latter when the wait opens the tile), its
                // reads of stage S have returned (lgkmcnt).
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(KSTEPS == 1 ? NDMA : 2 * NDMA) : "memory");
                __builtin_amdgcn_s_barrier();
            }
            frags_wait(std::integral_constant<int, slot>());
            if (kk == 0) {
                k_wait();  // requested a k-step ago, and the fragment wait has covered it
#pragma unroll
                for (int sub = 0; sub < NSUB; ++sub) {
                    ka[sub] = ka_next[sub];
                    kb[sub] = kb_next[sub];
                }
            }
            // RAGGED: lane group fk holds element 4 kk + fk of the tile.  A k-step whose elements are overlap
            // entirely (never the last one of a tile) is not multiplied: its MFMAs sit behind wave-uniform
            // branches, everything that is dealt out between them (each asynchronous read and DMA exists once
            // in the code) runs as always.  A k-step that is overlap in part has those elements zeroed.
            // (Only with more than two k-steps per tile: the launch rule below sends no run whose last tile is
            // overlap by half or more to this kernel, so with two k-steps none is ever idle, and the branches
            // would cost the fp64 kernel 3 %.)
            const bool idle = RAGGED && KSTEPS > 2 && !last && skip >= 4 * kk + 4;
            if (RAGGED && skip > 4 * kk && !idle) {
                const bool dead = 4 * kk + fk < skip;
#pragma unroll
                for (int b = 0; b < TM; ++b) fa[slot][b] = dead ? T(0) : fa[slot][b];
#pragma unroll
                for (int b = 0; b < TN; ++b) fb[slot][b] = dead ? T(0) : fb[slot][b];
            }
            static_for<0, NMFMA>([&](auto q_t) {
                constexpr int q = decltype(q_t)::value;
                if (!idle) acc[q / TN][q % TN] = Tr::mfma(fa[slot][q / TN], fb[slot][q % TN], acc[q / TN][q % TN]);
                if (q == 0) {
                    if (last) read_frags(std::integral_constant<int, slot ^ 1>(), I0(), std::integral_constant<int, SN>());
                    else read_frags(std::integral_constant<int, slot ^ 1>(), std::integral_constant<int, (kk + 1) % KSTEPS>(), stage_t);
                }
                if (kk == 0 && q >= 2 && q < 2 + 3 * NDMA && (q - 2) % 3 == 0) issue((q - 2) / 3, ld, ka, kb);
                if (last && q == (KSTEPS == 1 ? 2 + 3 * NDMA : 1)) advance();
                __builtin_amdgcn_sched_barrier(0);
            });
        });
    };

    {
        int kt = 0;
        for (; kt + kStages <= nk; kt += kStages) {
            tile_step(std::integral_constant<int, 0>());
            tile_step(std::integral_constant<int, 1>());
            tile_step(std::integral_constant<int, 2>());
            tile_step(std::integral_constant<int, 3>());
        }
        if (kt < nk) tile_step(std::integral_constant<int, 0>());
        if (kt + 1 < nk) tile_step(std::integral_constant<int, 1>());
        if (kt + 2 < nk) tile_step(std::integral_constant<int, 2>());
    }
    // the requests past the end have landed before the LDS goes away, and the last fragment reads (of a
    // tile nobody multiplies) have returned
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    frags_wait(I0());
    frags_wait(I1());

    store_tile_slab<T, BM, BN, TM, TN>(p, acc, s_coff, C, wm0, wn0, lane, partial, slab);
    __syncthreads();  // s_coff and the stages are the next item's
#ifdef CTC_DMA_ROUND_SYNC
    // Diagnostic build: the workgroups of an XCD that walk their items meet after every round of whole tiles
    // (bounded wait: a workgroup gives up after about 100 us), so that the eight workgroups sharing a panel
    // start the next tile together.  p.ka_stride carries the launch's generation number (the kernel does not
    // use it otherwise): the counters only ever grow.
    if (gridDim.x < (unsigned)n_items && !partial && p.flat_batch == 0 && sid < p.dp_tiles && sk_it >= sk_end) {
        if (t == 0) {
            const int round = (bid - (int)gridDim.x) / (int)gridDim.x;
            unsigned *ctr = &g_round_sync[round & 63][blockIdx.x & 7];
            const unsigned target = (unsigned)p.ka_stride * (gridDim.x >> 3);
            __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int tries = 0; tries < 2000 && __hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target; ++tries)
                __builtin_amdgcn_s_sleep(8);
        }
        __syncthreads();
    }
#endif
    }
#ifdef CTC_CLOCK
    if (t == 0 && blockIdx.y == 0 && (blockIdx.x == 700 || blockIdx.x == 2500 || blockIdx.x == 4500)) {
        const unsigned long long dc = __builtin_readcyclecounter() - ck_core0, dr = __builtin_amdgcn_s_memrealtime() - ck_real0;
        printf("[clock] block %d: %llu core ticks in %llu x 10 ns -> %.3f GHz\n", (int)blockIdx.x, dc, dr, (double)dc / (10.0 * (double)dr));
    }
#endif
}

// Fix-up pass of a launch with stream-K shares: every tile of the shared-out part that a share boundary cuts
// has its first piece in C already (alpha and beta applied) and the raw sums of its other pieces in the slabs
// of the shares that begin inside it; they are added in share order: C += alpha * (slab + slab + ...).
// One workgroup per 16 rows of a tile; tiles no boundary cuts have nothing to add.
template <typename T> __global__ void __launch_bounds__(256) sk_fixup_kernel(const GemmParams p, int nk)
{
    constexpr int BM = kBM, BN = kBN, ROWS = 16, VPR = BN / 2, RPP = 256 / VPR;
    const int tile_rel = blockIdx.x / (BM / ROWS);
    const int rblk = blockIdx.x % (BM / ROWS);
    const int tiles_per_entry = p.nbm * p.nbn;
    const int tail_tiles = tiles_per_entry * (p.flat_batch > 0 ? p.flat_batch : 1) - p.dp_tiles;
    const int g = p.sk_group, units = p.sk_units / g;  // runs
    const int64_t W = (int64_t)((tail_tiles + g - 1) / g) * nk, lo = (int64_t)(tile_rel / g) * nk, hi = lo + nk;
    // runs that begin strictly inside the tile's group of tiles: [u0, u1); the tile's slabs are those of workgroup
    // tile_rel % g of each
    int u0 = (int)(lo * units / W);
    u0 = u0 > 0 ? u0 - 1 : 0;
    while (u0 < units && sk_unit_start(u0, W, units, nk) <= lo) ++u0;
    int u1 = u0;
    while (u1 < units && sk_unit_start(u1, W, units, nk) < hi) ++u1;
    if (u0 == u1) return;
    int sid = p.dp_tiles + tile_rel, bz = 0;
    if (p.flat_batch > 0) {
        bz = sid / tiles_per_entry;
        sid -= bz * tiles_per_entry;
    }
    const int width = p.group * p.nbn;
    const int first_m = (sid / width) * p.group;
    const int gsz = min(p.nbm - first_m, p.group);
    const int tm = first_m + (sid % width) % gsz;
    const int tn = (sid % width) / gsz;
    const T *ws = static_cast<const T *>(p.ws);
    T *C = static_cast<T *>(p.C) + p.bat_c[bz];
    const T alpha = (T)p.alpha;
    const int cv = threadIdx.x % VPR;
    const int64_t n = (int64_t)tn * BN + 2 * cv;
    const bool ok0 = n < p.N, ok1 = n + 1 < p.N;
    const int64_t c0 = p.col_c[ok0 ? n : 0], c1 = p.col_c[ok1 ? n + 1 : 0];
    for (int rr = threadIdx.x / VPR; rr < ROWS; rr += RPP) {
        const int row = rblk * ROWS + rr;
        const int64_t m = (int64_t)tm * BM + row;
        if (m >= p.M) continue;
        T s0 = 0, s1 = 0;
        for (int u = u0; u < u1; ++u) {
            const T *src = ws + ((int64_t)u * g + tile_rel % g) * (BM * BN) + row * BN + 2 * cv;
            s0 += src[0];
            s1 += src[1];
        }
        const int64_t ro = p.row_c[m];
        if (ok0) C[ro + c0] += alpha * s0;
        if (ok1) C[ro + c1] += alpha * s1;
    }
}

template <typename T, int LA, int LB, int RB, int NSUB, bool RAGGED> int resident_slots()
{
    static std::atomic<int> cache[64];  // 0: not queried yet; host threads of several devices may race here
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cache[dev].load() == 0) {
        int cus = 0, per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemm_dma_kernel<T, LA, LB, RB, NSUB, RAGGED>, 256, 0) != hipSuccess) return 0;
        cache[dev].store(cus * per_cu > 0 ? cus * per_cu : -1);
    }
    const int slots = cache[dev].load();
    return slots > 0 ? slots : 0;
}

std::atomic<int> g_sk_group{getenv("CTC_TC_STREAMK_GROUP") ? atoi(getenv("CTC_TC_STREAMK_GROUP")) : 0};  // tiles per group of shares; 0: chosen per launch
std::atomic<int> g_stream_k{getenv("CTC_TC_STREAMK") ? atoi(getenv("CTC_TC_STREAMK")) : 0};  // 0 by cost, 1 never, 2 whenever possible

// Chunk overhead of the split along K, in contracted elements: prologue, raw sums to the workspace, the
// chunk's share of the fix-up pass; the same 128 as in gemm_kernels.hip.
constexpr int kMinChunkK = 128;

// A round of w workgroups on `slots` slots, in units of a full round's time.  With at most one workgroup per CU
// (w <= slots / 2) a workgroup has the CU's matrix pipes to itself and runs about twice as fast: measured, a tail
// of 64, 128 or 256 tiles behind full rounds costs 0.44 - 0.49 of a round in both element types, one of 320 a
// whole round (tools/tail_cost_probe.py).  `light_ok`: whether the launch may count on that (see schedule_one).
inline double round_cost(int64_t w, int slots, bool light_ok) { return (light_ok && 2 * w <= slots) ? 0.5 : 1.0; }

// The tail tiles in f chunks of equal length each: cost of their rounds in K-tile times.
inline double split_cost(int64_t tail, int f, int slots, int nk, int ovh, bool light_ok)
{
    const int64_t chunks = tail * f, full = chunks / slots, rest = chunks % slots;
    return ((double)full + (rest ? round_cost(rest, slots, light_ok) : 0.0)) * ((double)nk / f + ovh);
}

// f_cap: largest f allowed
int best_split(int64_t tail, int slots, int nk, int min_chunk_tiles, int f_cap, bool light_ok)
{
    int fmax = nk / min_chunk_tiles;
    if (fmax > 1024) fmax = 1024;
    if (fmax > f_cap) fmax = f_cap;
    int best_f = 1;
    double best = round_cost(tail, slots, light_ok) * (double)nk;
    for (int f = 2; f <= fmax; ++f) {
        const double cost = split_cost(tail, f, slots, nk, min_chunk_tiles, light_ok);
        if (cost < best * 0.98) {
            best = cost;
            best_f = f;
        }
    }
    return best_f;
}

template <typename T, int LA, int LB, int RB, int NSUB, bool RAGGED> hipError_t schedule_one(const GemmParams &p, GemmSchedule *out)
{
    constexpr int BK = NSUB * tile_k<T, RB>();
    GemmSchedule s;
    s.nbm = (int)((p.M + kBM - 1) / kBM);
    s.nbn = (int)((p.N + kBN - 1) / kBN);
    int64_t tiles = (int64_t)s.nbm * s.nbn;
    if (tiles > 0x7FFFFFFF) return hipErrorInvalidValue;
    // a batched launch: one list of tiles over all entries whenever its length fits (CTC_TC_NO_FLAT_BATCH: A/B runs)
    static const bool no_flat = getenv("CTC_TC_NO_FLAT_BATCH") != nullptr;
    // (a grid holds fewer than 2^32 threads: 2^24 workgroups of 256)
    if (p.batch > 1 && !no_flat && tiles * p.batch <= 0x7FFFFF) {
        s.flat_batch = p.batch;
        tiles *= p.batch;
    }
    s.dp_tiles = (int)tiles;
    s.split = 1;
    s.tail_tiles = 0;
    s.ws_bytes = 0;
    static const bool no_split = getenv("CTC_TC_NO_SPLITK") != nullptr;
    const int slots = resident_slots<T, LA, LB, RB, NSUB, RAGGED>();
    const int nk = RAGGED ? (int)((p.K / p.k_run) * ((p.k_run + BK - 1) / BK)) : (int)(p.K / BK);
    if (!no_split && slots > 0 && (p.batch == 1 || s.flat_batch > 0)) {
        const int tail = (int)(tiles % slots);
        const int64_t rounds = tiles / slots;
        if (tail > 0) {
            const int ovh = kMinChunkK / BK;  // cost of a piece beside its K-tiles, in K-tile times
            // A last round of at most one workgroup per CU is counted as half a round (round_cost).  Left to the
            // dispatcher, the late workgroups of an fp64 launch do not reliably land one per CU once tiles are
            // long (2 x 512 + 256 tiles of K = 4096 run at 63.6 TFLOP/s against 75.0 cut in two, 1 x 512 + 256 of
            // K = 8192 at 74.1 or 57.9 from run to run; fp32 shows no such spread), so an fp64 launch that leaves
            // such a round uncut behind full rounds takes the form in which one workgroup per slot walks the
            // items (launch_one): the last items then belong to the first workgroups of the grid, one per CU
            // (the same two shapes: 76.1 and 76.2).
            static const bool light_ok = getenv("CTC_TC_NO_LIGHT_ROUNDS") == nullptr;  // A/B runs
            // (b) stream-K over the last round and the full round before it (so that a share is at least one
            // tile long and no tile is cut more than once): every workgroup ends at the same time whatever the
            // tile count is, at the price of up to two more pieces per workgroup, a slab per share and the
            // fix-up pass.  With fewer tiles than slots the shares are shorter than a tile.
            const int sk_mode = g_stream_k.load() - 1;  // -1 by cost, 0 never, 1 whenever possible
            double cost_sk = 1e300;
            int sk_tiles = 0, sk_units = 0;
            // groups of 16 tiles (8 tile rows x 2 tile columns) unless groups of 8 leave a share at least 1 %
            // shorter (the last group of tiles is padded to a whole one; 648 tiles: 41 groups of 16 on 32 runs
            // against 81 of 8 on 64): the larger group shares more operand panels in the L2, measured worth about
            // that much.  An explicit "stream_k_group" option fixes the size.
            int sk_group = 16;
            if (sk_mode != 0 && nk >= 8 * kSkSnap) {
                sk_tiles = rounds >= 1 ? tail + slots : tail;
                const int forced = g_sk_group.load();
                double best_share = 1e300;
                for (int g : {16, 8}) {
                    if (forced > 0 && g != 16) break;
                    if (forced > 0) g = forced;
                    if (slots % g != 0) continue;
                    const int64_t W = (int64_t)((sk_tiles + g - 1) / g) * nk;  // iterations of groups of tiles
                    int runs = slots / g;
                    if (W / runs < 8 * kSkSnap) runs = (int)(W / (8 * kSkSnap));
                    if (runs < 2) continue;
                    const double share = (double)((W + runs - 1) / runs);
                    if (share < best_share * (g == 8 ? 0.99 : 1.0)) {
                        best_share = share;
                        sk_group = g;
                        sk_units = runs * g;
                    }
                }
                if (best_share < 1e300) cost_sk = (double)(rounds >= 1 ? rounds - 1 : 0) * nk + best_share + 1.5 * ovh;
                // fp32: shares measure 2 - 5 % behind equal chunks of the same predicted cost (704 tiles of
                // K = 8192: 142.1 against 146.8 TFLOP/s; 1216 of K = 4096: 143.3 against 144.6)
                if (sizeof(T) == 4) cost_sk *= 1.04;
            }
            // (a) the tiles of the last round in f chunks of equal length each (tail x f slabs of workspace; (b)
            // never needs more than one slab per workgroup slot)
            const int f = best_split(tail, slots, nk, ovh, 1024, light_ok);
            const double cost_split = (double)rounds * nk + (f >= 2 ? split_cost(tail, f, slots, nk, ovh, light_ok)
                                                                    : round_cost(tail, slots, light_ok) * (double)nk);
            if (cost_sk < cost_split * (sk_mode == 1 ? 1e9 : 0.995)) {
                s.tail_tiles = sk_tiles;
                s.dp_tiles = (int)tiles - sk_tiles;
                s.split = 1;
                s.sk_units = sk_units;
                s.sk_group = sk_group;
                s.ws_bytes = (size_t)sk_units * kBM * kBN * sizeof(T);
            } else if (f >= 2) {
                s.tail_tiles = tail;
                s.dp_tiles = (int)tiles - tail;
                s.split = f;
                s.ws_bytes = (size_t)tail * f * kBM * kBN * sizeof(T);
            }
            // fp64: more items than slots and a last round (of whole tiles or of chunks) of at most one per CU
            if (sizeof(T) == 8 && s.sk_units == 0) {
                const int64_t items = (int64_t)s.dp_tiles + (int64_t)s.tail_tiles * s.split, last = items % slots;
                if (items > slots && last > 0 && 2 * last <= slots) s.walk = 1;
            }
        }
    }
    *out = s;
    return hipSuccess;
}

template <typename T, int LA, int LB, int RB, int NSUB, bool RAGGED>
hipError_t launch_one(const GemmParams &p, const GemmSchedule &sch, void *ws, hipStream_t stream)
{
    GemmParams q = p;
    q.nbm = sch.nbm;
    q.nbn = sch.nbn;
    q.dp_tiles = sch.dp_tiles;
    q.split = sch.split;
    q.sk_units = sch.sk_units;
    q.sk_group = sch.sk_group > 0 ? sch.sk_group : 1;
    q.flat_batch = sch.flat_batch;
    q.plain_order = (sch.flat_batch > 0 && sch.nbm * sch.nbn == 1) ? 1 : 0;
    q.ws = ws;
    if (q.group < 1) q.group = 1;
    if (sch.tail_tiles > 0 && ws == nullptr) return hipErrorInvalidValue;
    unsigned items = (unsigned)(sch.dp_tiles + (sch.sk_units > 0 ? sch.sk_units : sch.tail_tiles * sch.split));
    // fp64, four rounds of workgroups or more: one workgroup per slot, each walking its items, instead of
    // one workgroup per item (extent 96, 10 rounds: +1 to +3 %; 80 and 88: +0.4 to +0.9 %; two or three
    // rounds: nothing).  Not fp32: there it loses (extent 96: 152.6 -> 141.8 TFLOP/s; the workgroups that
    // share panels drift further apart and the L2 misses, which the fp64 kernel hides, show).
    // CTC_TC_DMA_PERSIST=0 / 1: never / whenever there are more items than slots.
    static const int persist = getenv("CTC_TC_DMA_PERSIST") ? atoi(getenv("CTC_TC_DMA_PERSIST")) : -1;
    if (persist != 0) {
        const unsigned slots = (unsigned)resident_slots<T, LA, LB, RB, NSUB, RAGGED>();
        if (slots >= 8 && slots % 8 == 0 && items > slots && (persist == 1 || sch.walk || (sizeof(T) == 8 && items >= 4 * slots))) items = slots;
    }
    dim3 grid(items, sch.flat_batch > 0 ? 1u : (unsigned)p.batch, 1);
#ifdef CTC_DMA_ROUND_SYNC
    static std::atomic<int64_t> generation{0};
    q.ka_stride = generation.fetch_add(1) + 1;
#endif
    hipLaunchKernelGGL((gemm_dma_kernel<T, LA, LB, RB, NSUB, RAGGED>), grid, dim3(256), 0, stream, q);
    if (sch.sk_units > 0) {
        constexpr int BK = NSUB * tile_k<T, RB>();
        const int nk = RAGGED ? (int)((p.K / p.k_run) * ((p.k_run + BK - 1) / BK)) : (int)(p.K / BK);
        hipLaunchKernelGGL((sk_fixup_kernel<T>), dim3((unsigned)(sch.tail_tiles * (kBM / 16))), dim3(256), 0, stream, q, nk);
    } else if (sch.tail_tiles > 0) {
        hipError_t e = launch_splitk_fixup_128(sizeof(T) == 8 ? 0 : 1, q, sch.tail_tiles, stream);
        if (e != hipSuccess) return e;
    }
    return hipGetLastError();
}

template <typename T, int RB, int NSUB, bool RAGGED>
hipError_t dispatch(int layout_a, int layout_b, const GemmParams &p, GemmSchedule *sch_out, const GemmSchedule *sch_in,
                    void *ws, hipStream_t stream)
{
#define CTC_CASE(A_, B_) \
    case A_ * 2 + B_: return sch_out ? schedule_one<T, A_, B_, RB, NSUB, RAGGED>(p, sch_out) : launch_one<T, A_, B_, RB, NSUB, RAGGED>(p, *sch_in, ws, stream);
    switch (layout_a * 2 + layout_b) {
        CTC_CASE(0, 0) CTC_CASE(0, 1) CTC_CASE(1, 0) CTC_CASE(1, 1)
    default: return hipErrorInvalidValue;
    }
#undef CTC_CASE
}

// The geometry the launch takes: 0 none (it does not qualify), 1: 64-byte rows, 2: two images of 32-byte rows
// per stage (k tables affine in groups of half a K-tile), 3: one image of 32-byte rows (K a multiple of half
// a K-tile only), 4: 64-byte rows with ragged K-tiles (k tables affine in runs that are no multiple of a K-tile)
int dma_gemm_geometry(int dtype, int layout_a, int layout_b, int bm, int bn, const GemmParams &p)
{
    static const bool off = getenv("CTC_TC_NO_DMA_GEMM") != nullptr;
    if (off || bm != kBM || bn != kBN || (dtype != 0 && dtype != 1)) return 0;
    if ((layout_a != 0 && layout_a != 1) || (layout_b != 0 && layout_b != 1)) return 0;
    const int vec = dtype == 0 ? 2 : 4;
    if (p.batch < 1 || p.batch > 65535 || p.M < vec || p.N < vec) return 0;
    if (layout_a == 1 && p.M % vec != 0) return 0;
    if (layout_b == 1 && p.N % vec != 0) return 0;
    if (layout_a == 0 && p.ka_in != 1) return 0;
    if (layout_b == 0 && p.kb_in != 1) return 0;
    // an fp32 outer-major lane offset carries one step along k
    if (dtype == 1 && (p.ka_in >= (int64_t(1) << 28) || p.kb_in >= (int64_t(1) << 28))) return 0;
    const int bk = dtype == 0 ? 8 : 16;  // K-tile of the 64-byte geometry
    const bool full = dtype == 0 ? p.k8_affine != 0 : p.k16_affine != 0;
    const bool half = dtype == 0 ? p.k4_affine != 0 : p.k8_affine != 0;
    static const bool force_ragged = getenv("CTC_TC_DMA_FORCE_RAGGED") != nullptr;  // A/B runs of the ragged bookkeeping
    if (full && p.K >= bk && p.K % bk == 0 && !(force_ragged && p.k_run >= bk)) return 1;
    // Ragged K-tiles of the 64-byte geometry: runs of k_run affine elements cut into ceil(k_run / bk) tiles,
    // the last one overlapping its predecessor.  Taken when no geometry with exact tiles applies.
    static const bool no_ragged = getenv("CTC_TC_NO_DMA_RAGGED") != nullptr;
    bool ragged = false;
    if (!no_ragged && p.k_run >= bk && p.k_run <= 0x7FFFFFFF && p.K % p.k_run == 0 && p.K / p.k_run * ((p.k_run + bk - 1) / bk) <= 0x3FFFFFFF) {
        // the last tile of a run must be live by more than half: measured against the register-staged kernel,
        // fp64 extents 62 .. 94 with 6 live elements of 8 gain 2.5 - 4.3 %, with 2 of 8 they change by -2.4 .. +1.8 %;
        // fp32 with 12 of 16 gains 4 - 6 %, with 4 of 16 0 - 3 %
        const int64_t live = p.k_run - (p.k_run - 1) / bk * bk;
        ragged = 2 * live > bk;
    }
    if (ragged && force_ragged) return 4;
    if (half && p.K >= bk && p.K % bk == 0) return 2;
    // One image of 32-byte rows has a barrier per sixteen MFMAs per wave: it beats the register-staged kernel
    // on problems of two rounds of tiles and more and loses below (fp32, 625 tiles mostly cut along K: 118 -> 110).
    const int64_t tiles = ((p.M + kBM - 1) / kBM) * ((p.N + kBN - 1) / kBN);
    if (half && p.K >= bk / 2 && p.K % (bk / 2) == 0 && tiles * p.batch >= 1024) return 3;
    if (ragged) return 4;
    return 0;
}

template <typename T>
hipError_t dispatch_geometry(int g, int la, int lb, const GemmParams &p, GemmSchedule *sch_out, const GemmSchedule *sch_in, void *ws,
                             hipStream_t stream)
{
    switch (g) {
    case 1: return dispatch<T, 64, 1, false>(la, lb, p, sch_out, sch_in, ws, stream);
    case 2: return dispatch<T, 32, 2, false>(la, lb, p, sch_out, sch_in, ws, stream);
    case 3: return dispatch<T, 32, 1, false>(la, lb, p, sch_out, sch_in, ws, stream);
    case 4: return dispatch<T, 64, 1, true>(la, lb, p, sch_out, sch_in, ws, stream);
    default: return hipErrorInvalidValue;
    }
}

}  // namespace

bool dma_gemm_supported(int dtype, int layout_a, int layout_b, int bm, int bn, const GemmParams &p)
{
    return dma_gemm_geometry(dtype, layout_a, layout_b, bm, bn, p) != 0;
}

void dma_gemm_set_stream_k(int mode) { g_stream_k.store(mode < 0 ? 0 : (mode > 2 ? 2 : mode)); }
int dma_gemm_get_stream_k() { return g_stream_k.load(); }
void dma_gemm_set_stream_k_group(int g) { g_sk_group.store(g < 0 ? 0 : (g > 64 ? 64 : g)); }
int dma_gemm_get_stream_k_group() { return g_sk_group.load(); }

int dma_gemm_kind(int dtype, int layout_a, int layout_b, int bm, int bn, const GemmParams &p)
{
    return dma_gemm_geometry(dtype, layout_a, layout_b, bm, bn, p);
}

// bm / bn are the 128 x 128 tile here: the callers have asked dma_gemm_supported
hipError_t schedule_gemm_dma(int dtype, int layout_a, int layout_b, const GemmParams &p, GemmSchedule *out)
{
    const int g = dma_gemm_geometry(dtype, layout_a, layout_b, kBM, kBN, p);
    return dtype == 0 ? dispatch_geometry<double>(g, layout_a, layout_b, p, out, nullptr, nullptr, nullptr)
                      : dispatch_geometry<float>(g, layout_a, layout_b, p, out, nullptr, nullptr, nullptr);
}

hipError_t launch_gemm_dma(int dtype, int layout_a, int layout_b, const GemmParams &p, const GemmSchedule &schedule,
                           void *workspace, hipStream_t stream)
{
    const int g = dma_gemm_geometry(dtype, layout_a, layout_b, kBM, kBN, p);
    return dtype == 0 ? dispatch_geometry<double>(g, layout_a, layout_b, p, nullptr, &schedule, workspace, stream)
                      : dispatch_geometry<float>(g, layout_a, layout_b, p, nullptr, &schedule, workspace, stream);
}

}  // namespace ctc
