// bf16 MFMA GEMM for tensor contractions on gfx950 (bf16 operands, f32 accumulation,
// bf16 result rounded to nearest even once), with the mode permutation fused into the
// operand loads through the same element-offset tables as the f64/f32 kernel.
//
// v_mfma_f32_16x16x32_bf16 wants, in lane l, eight consecutive k of one row (A) or one
// column (B): A[row l&15][k = 8(l>>4) + j], B[k = 8(l>>4) + j][col l&15], j = 0..7.  A
// workgroup owns a BT x BT tile of C (BT = 128: 2 x 2 waves of 64 x 64, two workgroups per
// CU; BT = 256: 2 x 4 waves of 128 x 64, one workgroup per CU, half the operand bytes per
// flop on the L2 -> LDS path, which is what bounds the 128 tile on large problems) and walks
// K in tiles of 64; each operand tile lives in LDS in one of two images, chosen by which
// GEMM dimension is unit-stride in memory:
//   K-major image  [row][64 k], 128-byte rows.  The 16-byte chunk c (8 k) of row r sits at
//                  128 r + 16 (c ^ ((r >> 1) & 7)).  Filled with 16-byte vectors along K
//                  (or element by element when no vectors are legal); a fragment is one
//                  ds_read_b128.  The XOR makes the sixteen lanes ds_read_b128 services
//                  together ({0-3, 12-15, 20-27}, ...: eight rows at chunk c and eight at
//                  chunk c ^ 1) fall on sixteen different 16-byte slots of the 256-byte
//                  bank row.
//   outer image    [k][BT rows], 2 BT-byte lines, chunk c (8 rows) of line k at
//                  2 BT k + 16 (c ^ (((k & 3) << 2) | ((k >> 2) & 3))).  Filled with 16-byte
//                  vectors along the rows; a fragment is two ds_read_b64_tr_b16 (the
//                  hardware transposes a block of 4 k x 16 rows per 16 lanes), conflict-free
//                  because the two blocks a 32-lane half reads are 8 lines apart.
// Staging uses two LDS buffers: the loads of tile t+1 are in flight while tile t feeds the
// MFMAs, one barrier per tile; two workgroups share a CU.  Tiles inside K go global -> LDS by
// DMA (global_load_lds_dwordx4), the K tail and element-load operands through registers.
#include "gemm_device.hpp"
#include "kernels.hpp"
#include "lds_dma.hpp"

#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

namespace ctc {

namespace {

using gemm_detail::kLayOuterVec;
using gemm_detail::kLayScalar;

typedef __bf16 bf8_t __attribute__((ext_vector_type(8)));
typedef short s4_t __attribute__((ext_vector_type(4)));
typedef short s8_t __attribute__((ext_vector_type(8)));
typedef float f4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u2_t __attribute__((ext_vector_type(2)));

constexpr int BK = 64;
// BT: edge of the square C tile (128 or 256).  2 BT threads; either image of a BT-row operand
// tile takes 128 BT bytes (16 or 32 KiB).
template <int BT> constexpr int image_bytes() { return BT * BK * 2; }

__device__ __forceinline__ int kmaj_off(int row, int chunk) { return 128 * row + 16 * (chunk ^ ((row >> 1) & 7)); }
template <int BT> __device__ __forceinline__ int outer_off(int k, int chunk)
{
    return 2 * BT * k + 16 * (chunk ^ (((k & 3) << 2) | ((k >> 2) & 3)));
}

__device__ __forceinline__ unsigned short f32_to_bf16(float f)
{
    unsigned u = __float_as_uint(f);
    if ((u & 0x7FFFFFFFu) > 0x7F800000u) return (unsigned short)((u >> 16) | 0x0040u);  // NaN stays NaN
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// Stages one BT x 64 operand tile with 2 BT threads.  tab_rows / tab_k are the operand's element-offset
// tables (rows: M or N index, k: K index; the device k tables carry kKTablePad extra
// entries, so k0 + 63 is always a valid index).  Rows past the operand's extent are
// clamped (their products land in C elements that are never stored); k past K reads
// as zero.
template <int LAY, int BT> struct StagerBf16;

// LDS-DMA (global_load_lds_dwordx4): lane l of a wave instruction writes 16 bytes at the wave's
// LDS base + 16 l, so a thread must fetch the chunk whose image slot is 16 t + 32 BT i (pass i
// of four): the swizzle is applied to the SOURCE chunk (which is the same for all four of a
// thread's passes, because the XOR term repeats every 16 rows / 16 lines).
__device__ __forceinline__ void dma16(const void *src, unsigned char *lds_wave_base)
{
    typedef __attribute__((address_space(1))) const void *gptr_t;
    typedef __attribute__((address_space(3))) void *lptr_t;
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lds_wave_base, 16, 0, 0);
}

template <int BT> struct StagerBf16<0 /* unit stride along K */, BT> {
    static constexpr int RPP = BT / 4;  // rows per pass
    static constexpr bool OUTER = false;
    static constexpr bool DIRECT = true;
    int64_t fix_off[4];
    u4_t regs[4];
    int chunk, row0;
    __device__ __forceinline__ void init(const int64_t *__restrict__ tab_rows, int64_t r0, int64_t R, int t)
    {
        chunk = (t & 7) ^ ((t >> 4) & 7);  // lands in slot t & 7 of its row
        row0 = t >> 3;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t r = r0 + row0 + RPP * i;
            fix_off[i] = tab_rows[r < R ? r : R - 1];
        }
    }
    __device__ __forceinline__ void issue(const unsigned short *__restrict__ base, const int64_t *__restrict__ tab_k,
                                          int64_t k0, int64_t)
    {
        const int64_t ko = tab_k[k0 + 8 * chunk];
#pragma unroll
        for (int i = 0; i < 4; ++i) regs[i] = *reinterpret_cast<const u4_t *>(base + fix_off[i] + ko);
    }
    // Whole tile inside K: global -> LDS without passing through registers.  The k offsets come
    // from prefetch(), called one tile earlier: an ordinary load issued while a DMA is in flight
    // would make the compiler drain the DMA before the load's first use.
    int64_t koff;
    __device__ __forceinline__ void prefetch(const int64_t *__restrict__ tab_k, int64_t k0) { koff = tab_k[k0 + 8 * chunk]; }
    __device__ __forceinline__ void direct(const unsigned short *__restrict__ base, unsigned char *img, int wave)
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(base + fix_off[i] + koff, img + 32 * BT * i + 1024 * wave);
    }
    __device__ __forceinline__ void commit(unsigned char *lds, int64_t k0, int64_t K)
    {
        const bool live = k0 + 8 * chunk < K;  // K is a multiple of 8 whenever this layout is chosen
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u4_t v = regs[i];
            if (!live) v = u4_t{0, 0, 0, 0};
            *reinterpret_cast<u4_t *>(lds + kmaj_off(row0 + RPP * i, chunk)) = v;
        }
    }
};

template <int BT> struct StagerBf16<kLayOuterVec /* unit stride along the rows */, BT> {
    static constexpr bool OUTER = true;
    static constexpr bool DIRECT = true;
    static constexpr int CPL = BT / 8;  // 16-byte chunks per line
    int64_t fix_off;
    u4_t regs[4];
    int chunk, line0;
    __device__ __forceinline__ void init(const int64_t *__restrict__ tab_rows, int64_t r0, int64_t R, int t)
    {
        line0 = t / CPL;  // 0..15
        chunk = (t % CPL) ^ (((line0 & 3) << 2) | ((line0 >> 2) & 3));  // lands in slot t % CPL of its line
        const int64_t r = r0 + 8 * chunk;
        fix_off = tab_rows[r < R ? r : R - 8];  // R is a multiple of 8 whenever this layout is chosen
    }
    __device__ __forceinline__ void issue(const unsigned short *__restrict__ base, const int64_t *__restrict__ tab_k,
                                          int64_t k0, int64_t)
    {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            regs[i] = *reinterpret_cast<const u4_t *>(base + fix_off + tab_k[k0 + line0 + 16 * i]);
    }
    int64_t koff[4];
    __device__ __forceinline__ void prefetch(const int64_t *__restrict__ tab_k, int64_t k0)
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) koff[i] = tab_k[k0 + line0 + 16 * i];
    }
    __device__ __forceinline__ void direct(const unsigned short *__restrict__ base, unsigned char *img, int wave)
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) dma16(base + fix_off + koff[i], img + 32 * BT * i + 1024 * wave);
    }
    __device__ __forceinline__ void commit(unsigned char *lds, int64_t k0, int64_t K)
    {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            u4_t v = regs[i];
            if (k0 + line0 + 16 * i >= K) v = u4_t{0, 0, 0, 0};
            *reinterpret_cast<u4_t *>(lds + outer_off<BT>(line0 + 16 * i, chunk)) = v;
        }
    }
};

template <int BT> struct StagerBf16<kLayScalar /* element loads, any strides */, BT> {
    // The fallback when no 16-byte vectors are legal (odd extents, odd strides, unaligned
    // base); 128 tile only.  A thread owns 8 rows x 4 k of the tile; sixteen neighbouring
    // lanes run along whichever of the two dimensions is contiguous in memory.  Element pairs
    // share a register while the tile waits for the MFMAs of the previous one.
    static_assert(BT == 128, "element loads are staged for the 128 tile only");
    static constexpr bool OUTER = false;
    static constexpr bool DIRECT = false;
    int64_t fix_off[8];
    unsigned regs[16];
    __device__ __forceinline__ void prefetch(const int64_t *, int64_t) {}
    __device__ __forceinline__ void direct(const unsigned short *, unsigned char *, int) {}
    int row0, kk0;
    __device__ __forceinline__ void init(const int64_t *__restrict__ tab_rows, int64_t r0, int64_t R, int t)
    {
        const bool rows_fast = R > 1 && tab_rows[1] - tab_rows[0] == 1;
        row0 = rows_fast ? (t & 15) : (t >> 4);
        kk0 = rows_fast ? (t >> 4) : (t & 15);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t r = r0 + row0 + 16 * i;
            fix_off[i] = tab_rows[r < R ? r : R - 1];
        }
    }
    __device__ __forceinline__ void issue(const unsigned short *__restrict__ base, const int64_t *__restrict__ tab_k,
                                          int64_t k0, int64_t)
    {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int64_t ko = tab_k[k0 + kk0 + 16 * j];
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                const unsigned lo = base[fix_off[i] + ko], hi = base[fix_off[i + 1] + ko];
                regs[4 * j + i / 2] = lo | (hi << 16);
            }
        }
    }
    __device__ __forceinline__ void commit(unsigned char *lds, int64_t k0, int64_t K)
    {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = kk0 + 16 * j;
            const bool live = k0 + k < K;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const unsigned pair = regs[4 * j + i / 2];
                const unsigned short v = (unsigned short)((i & 1) ? (pair >> 16) : (pair & 0xFFFFu));
                *reinterpret_cast<unsigned short *>(lds + kmaj_off(row0 + 16 * i, k >> 3) + 2 * (k & 7)) =
                    live ? v : (unsigned short)0;
            }
        }
    }
};

// MFMA operand of the 16 rows (columns) starting at r16 for k-step ks (32 k) of a tile image
template <bool OUTER, int BT> __device__ __forceinline__ bf8_t fragment(const unsigned char *img, int r16, int ks, int lane)
{
    if constexpr (!OUTER) {
        const int row = r16 + (lane & 15);
        return *reinterpret_cast<const bf8_t *>(img + kmaj_off(row, 4 * ks + (lane >> 4)));
    } else {
    // lane 4q+p of a 16-lane group names line q, rows 4p..4p+3 of the block; it receives the
    // four lines of row (lane & 15).  Lines k = 32 ks + 8 g + q and k + 4; their XOR terms are
    // 2 Y and 2 Y + 1 with Y = 2 q + (g & 1), so chunk (r16 / 8 + (p >> 1)) of the two lines sits
    // at 32 ((r16 / 16) ^ Y) + 16 (p >> 1) and at the same place with bit 4 flipped: one XOR with
    // the (wave-uniform) row block per fragment, everything else is per-lane constant or immediate.
    const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
    const int lane_off = 2 * BT * (8 * g + q) + 16 * (p >> 1) + 8 * (p & 1);
    const int y32 = 32 * (2 * q + (g & 1));
    const int off = lane_off + ((2 * r16) ^ y32);
    typedef s4_t __attribute__((address_space(3))) * lds_s4_ptr;
    const s4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (lds_s4_ptr)(const_cast<unsigned char *>(img) + off + 64 * BT * ks));
    const s4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (lds_s4_ptr)(const_cast<unsigned char *>(img) + (off ^ 16) + 64 * BT * ks + 8 * BT));
    const s8_t both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf8_t, both);
    }
}

// Epilogue of a BT x BT tile held as TM x TN 16 x 16 f32 accumulators per wave (wave origin wm0, wn0).
// smem: at least 2 BT BT bytes that no DMA is writing any more (the finished tile passes through them
// on its way to 16-byte stores); every thread of the workgroup must call.
template <int BT, int TM, int TN>
__device__ __forceinline__ void store_tile_bf16(const GemmParams &p, f4_t (&acc)[TM][TN], unsigned char *smem, unsigned short *C,
                                                int64_t m0, int64_t n0, int wm0, int wn0, int t, int lane, bool partial, int sid,
                                                int chunk)
{
    constexpr int BM = BT, BN = BT;
    // Epilogue.  Accumulator register r of lane l: row 4 (l >> 4) + r, column l & 15.
    const float alpha = (float)p.alpha, beta = (float)p.beta;
    const int frow = lane & 15;
    if (partial) {
        // raw f32 accumulators of this K chunk: tile-shaped slab [BM][BN] in the workspace
        float *slab = static_cast<float *>(p.ws) + ((int64_t)(sid - p.dp_tiles) * p.split + chunk) * (BM * BN);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm0 + 16 * i + 4 * (lane >> 4) + r;
#pragma unroll
                for (int j = 0; j < TN; ++j) slab[row * BN + wn0 + 16 * j + frow] = acc[i][j][r];
            }
        return;
    }
    if (p.c_vec >= 8 && beta == 0.f) {
        // The tile goes through LDS (the operand stages are free now) and leaves in 16-byte vectors
        // along the rows of C: 32 (16 for the 128 tile) neighbouring lanes write one whole tile row
        // of 2 BT bytes, against 32-byte pieces per row when a quad stores its 4 x 4 block directly
        // (16384 x 16384 x 256: 0.94 TB/s of C).  Image [row][BT] bf16, its 16-byte chunks XOR-ed
        // with the row so that the quads' 8-byte writes of sixteen rows spread over the banks.
        typedef gemm_detail::Traits<float> Tr;
        constexpr int NTH = 2 * BT;         // threads of the workgroup
        constexpr int CPR = BT / 8;         // 16-byte chunks per tile row
        constexpr int VPT = BT * CPR / NTH;  // vectors per thread on the way out
        const int out_chunk = t % CPR, out_row0 = t / CPR;
        // this thread's C offsets, requested before the barrier
        int64_t ro[VPT];
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const int64_t m = m0 + out_row0 + v * (NTH / CPR);
            ro[v] = m < p.M ? p.row_c[m] : INT64_MIN;
        }
        const int64_t nv = n0 + 8 * out_chunk;
        const int64_t co = nv < p.N ? p.col_c[nv] : INT64_MIN;
        __syncthreads();  // every wave has read its last fragments
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int row = wm0 + 16 * i + Tr::cvec_row(lane, 0);
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const f4_t v = Tr::cvec(acc[i][j], lane, 0);  // every lane takes part in the exchange
                const int col = wn0 + 16 * j + (frow & ~3);
                u2_t w;
                w[0] = (unsigned)f32_to_bf16(alpha * v[0]) | ((unsigned)f32_to_bf16(alpha * v[1]) << 16);
                w[1] = (unsigned)f32_to_bf16(alpha * v[2]) | ((unsigned)f32_to_bf16(alpha * v[3]) << 16);
                const int chunk = (col >> 3) ^ (row & (CPR - 1));
                *reinterpret_cast<u2_t *>(smem + row * (2 * BT) + 16 * chunk + 2 * (col & 7)) = w;
            }
        }
        __syncthreads();
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const int row = out_row0 + v * (NTH / CPR);
            const u4_t w = *reinterpret_cast<const u4_t *>(smem + row * (2 * BT) + 16 * (out_chunk ^ (row & (CPR - 1))));
            if (ro[v] != INT64_MIN && co != INT64_MIN) *reinterpret_cast<u4_t *>(C + ro[v] + co) = w;
        }
        return;
    }
    if (p.c_vec >= 4) {
        // 8-byte stores: a 4 x 4 transpose inside each quad leaves lane q with four
        // neighbouring columns of row register q
        typedef gemm_detail::Traits<float> Tr;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int64_t m = m0 + wm0 + 16 * i + Tr::cvec_row(lane, 0);
            const int64_t ro = p.row_c[m < p.M ? m : 0];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int64_t n = n0 + wn0 + 16 * j + (frow & ~3);
                const f4_t v = Tr::cvec(acc[i][j], lane, 0);  // every lane takes part in the exchange
                if (m < p.M && n < p.N) {                      // N is a multiple of the vector length
                    u2_t *dst = reinterpret_cast<u2_t *>(C + ro + p.col_c[n]);
                    float o[4] = {alpha * v[0], alpha * v[1], alpha * v[2], alpha * v[3]};
                    if (beta != 0.f) {
                        const u2_t old = *dst;
                        o[0] += beta * bf16_to_f32((unsigned short)(old[0] & 0xFFFFu));
                        o[1] += beta * bf16_to_f32((unsigned short)(old[0] >> 16));
                        o[2] += beta * bf16_to_f32((unsigned short)(old[1] & 0xFFFFu));
                        o[3] += beta * bf16_to_f32((unsigned short)(old[1] >> 16));
                    }
                    u2_t w;
                    w[0] = (unsigned)f32_to_bf16(o[0]) | ((unsigned)f32_to_bf16(o[1]) << 16);
                    w[1] = (unsigned)f32_to_bf16(o[2]) | ((unsigned)f32_to_bf16(o[3]) << 16);
                    *dst = w;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int64_t n = n0 + wn0 + 16 * j + frow;
        const int64_t co = p.col_c[n < p.N ? n : 0];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t m = m0 + wm0 + 16 * i + 4 * (lane >> 4) + r;
                if (m < p.M && n < p.N) {
                    unsigned short *dst = C + p.row_c[m] + co;
                    float o = alpha * acc[i][j][r];
                    if (beta != 0.f) o += beta * bf16_to_f32(*dst);
                    *dst = f32_to_bf16(o);
                }
            }
        }
    }
}

template <int LA, int LB, int BT>
__global__ void __launch_bounds__(2 * BT, BT == 128 ? 2 : 1) gemm_bf16_kernel(const GemmParams p)
{
    typedef StagerBf16<LA, BT> SA;
    typedef StagerBf16<LB, BT> SB;
    constexpr int BM = BT, BN = BT, kImageBytes = image_bytes<BT>();
    // 16 x 16 accumulators per wave: a 64 x 64 (BT = 128) or 128 x 64 (BT = 256) block of C
    constexpr int TM = BT / 32, TN = 4, WAVES_N = BT / 64;

    // [buffer][A image | B image]
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 2 * kImageBytes];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm0 = (wave / WAVES_N) * (16 * TM), wn0 = (wave % WAVES_N) * 64;

    // Block -> item.  Blocks b and b+8 share an XCD (and its L2), so every XCD gets a contiguous
    // run of tile ids, walked in groups of `group` tile rows to share A and B panels.  The
    // first dp_tiles tiles run whole; the tiles of the last, partly filled round of the grid
    // are cut along K into `split` chunks whose f32 accumulators go to a workspace
    // (splitk_fixup_bf16_kernel sums them in chunk order).
    int sid, chunk = 0;
    bool partial = false;
    {
        const int bid = blockIdx.x;
        if (bid < p.dp_tiles) {
            const int nblk = p.dp_tiles;
            const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
            sid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        } else {
            const int rel = bid - p.dp_tiles;
            sid = p.dp_tiles + rel / p.split;
            chunk = rel % p.split;
            partial = true;
        }
    }
    // a launch without a batch dimension in its grid: the tile id counts through the batch entries (GemmParams::flat_batch)
    const int list_id = sid;  // position in the launch's list of tiles: what a chunk's workspace slab is counted from
    int bz = blockIdx.y;
    if (p.flat_batch > 0) {
        bz = sid / (p.nbm * p.nbn);
        sid -= bz * (p.nbm * p.nbn);
    }
    const int width = p.group * p.nbn;
    const int first_m = (sid / width) * p.group;
    const int gsz = min(p.nbm - first_m, p.group);
    const int tm = first_m + (sid % width) % gsz;
    const int tn = (sid % width) / gsz;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

    const unsigned short *A = static_cast<const unsigned short *>(p.A) + p.bat_a[bz];
    const unsigned short *B = static_cast<const unsigned short *>(p.B) + p.bat_b[bz];
    unsigned short *C = static_cast<unsigned short *>(p.C) + p.bat_c[bz];

    SA sa;
    SB sb;
    sa.init(p.row_a, m0, p.M, t);
    sb.init(p.col_b, n0, p.N, t);

    f4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

    const int nk_total = (int)((p.K + BK - 1) / BK);
    const int kt_first = partial ? (int)((int64_t)chunk * nk_total / p.split) : 0;
    const int nk = partial ? (int)((int64_t)(chunk + 1) * nk_total / p.split) : nk_total;  // one past the last K-tile
    // A tile that lies inside K goes global -> LDS by DMA (vector layouts); the last tile of a
    // K that is not a multiple of 64 needs its tail zeroed and goes through registers, as do
    // operands read element by element.
    auto is_direct = [&](auto &st, int64_t k0) {
        typedef std::remove_reference_t<decltype(st)> S;
        return S::DIRECT && k0 + BK <= p.K;
    };
    // The 256 tile has no registers to hold a tile across the MFMAs: its one register-staged
    // tile (the K tail) is loaded and committed after them, unoverlapped.
    constexpr bool kLateIssue = BT == 256;
    auto stage_begin = [&](auto &st, const unsigned short *base, const int64_t *tab_k, int64_t k0, unsigned char *img) {
        if (is_direct(st, k0)) st.direct(base, img, wave);
        else if (!kLateIssue) st.issue(base, tab_k, k0, p.K);
    };
    auto stage_end = [&](auto &st, const unsigned short *base, const int64_t *tab_k, int64_t k0, unsigned char *img) {
        if (!is_direct(st, k0)) {
            if (kLateIssue) st.issue(base, tab_k, k0, p.K);
            st.commit(img, k0, p.K);
        }
    };
    // k offsets of a DMA tile are fetched one tile ahead (the k tables are padded past K, but
    // only by one tile)
    auto prefetch = [&](int kt) {
        const int64_t k0 = (int64_t)kt * BK;
        if (kt < nk) {
            if (is_direct(sa, k0)) sa.prefetch(p.k_a, k0);
            if (is_direct(sb, k0)) sb.prefetch(p.k_b, k0);
        }
    };
    {
        const int64_t k_first = (int64_t)kt_first * BK;
        prefetch(kt_first);
        stage_begin(sa, A, p.k_a, k_first, smem);
        stage_begin(sb, B, p.k_b, k_first, smem + kImageBytes);
        prefetch(kt_first + 1);
        stage_end(sa, A, p.k_a, k_first, smem);
        stage_end(sb, B, p.k_b, k_first, smem + kImageBytes);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the DMA counts as an outstanding load
    __syncthreads();

    for (int kt = kt_first; kt < nk; ++kt) {
        const int buf = (kt - kt_first) & 1;
        const unsigned char *ia = smem + buf * 2 * kImageBytes;
        const unsigned char *ib = ia + kImageBytes;
        unsigned char *na = smem + (buf ^ 1) * 2 * kImageBytes;
        const bool more = kt + 1 < nk;
        const int64_t k_next = (int64_t)(kt + 1) * BK;
        // Where the next tile's DMAs are issued.  A global_load_lds queued in front of a ds_read_b64_tr_b16 delays
        // it (8192^3 on the 256 tile, A K-major / B outer: 1009 -> 1155 TFLOP/s with the DMAs behind the up-front
        // reads; A outer: 955 -> 1035 with the DMAs behind the rolling reads of A as well), while the K-major
        // reads are indifferent and an earlier DMA has more time to land (both K-major: 1141 at once, 1109
        // behind the up-front reads, 1036 behind the first k-step).
        constexpr int kDmaAt = SA::OUTER ? 2 : (SB::OUTER ? 1 : 0);
        auto begin_next = [&]() {
            if (more) {
                // the other buffer was last read before the barrier that ended the previous tile
                stage_begin(sa, A, p.k_a, k_next, na);
                stage_begin(sb, B, p.k_b, k_next, na + kImageBytes);
                prefetch(kt + 2);
            }
        };
        if (kDmaAt == 0) begin_next();
        // Both k-steps' fragments are requested up front, in k-step order, so that the second
        // half of the reads lands while the first sixteen MFMAs run; the order is pinned, the
        // compiler only adds the counted waits.
        static_assert(BK == 64, "two k-steps per tile");
        if constexpr (BT == 256) {
            // 128 accumulator registers leave no room for both k-steps' fragments (24 x 4
            // registers): the A fragments are a rolling set, row i of the second k-step is
            // requested right behind the four MFMAs that consumed row i of the first.
            bf8_t fa[TM], fb[2][TN];
            // the fragment addresses are rebuilt from the lane id in every tile (a few VALU
            // instructions) instead of living in two dozen registers across the loop
            int lane_it = lane;
            asm volatile("" : "+v"(lane_it));
            fa[0] = fragment<SA::OUTER, BT>(ia, wm0, 0, lane_it);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[0][j] = fragment<SB::OUTER, BT>(ib, wn0 + 16 * j, 0, lane_it);
#pragma unroll
            for (int i = 1; i < TM; ++i) fa[i] = fragment<SA::OUTER, BT>(ia, wm0 + 16 * i, 0, lane_it);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[1][j] = fragment<SB::OUTER, BT>(ib, wn0 + 16 * j, 1, lane_it);
            __builtin_amdgcn_sched_barrier(0);
            if (kDmaAt == 1) begin_next();
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[0][j], acc[i][j], 0, 0, 0);
                fa[i] = fragment<SA::OUTER, BT>(ia, wm0 + 16 * i, 1, lane_it);
                __builtin_amdgcn_sched_barrier(0);
            }
            if (kDmaAt == 2) begin_next();
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        } else {
        bf8_t fa[2][TM], fb[2][TN];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[ks][i] = fragment<SA::OUTER, BT>(ia, wm0 + 16 * i, ks, lane);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[ks][j] = fragment<SB::OUTER, BT>(ib, wn0 + 16 * j, ks, lane);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (kDmaAt != 0) begin_next();  // the 128 tile reads every fragment up front
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[ks][i], fb[ks][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        }
        if (more) {
            stage_end(sa, A, p.k_a, k_next, na);
            stage_end(sb, B, p.k_b, k_next, na + kImageBytes);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    store_tile_bf16<BT, TM, TN>(p, acc, smem, C, m0, n0, wm0, wn0, t, lane, partial, list_id, chunk);
}

// ---- ring kernel ---------------------------------------------------------------------------------------
// The 256 x 256 tile for two 16-byte-vector operands with the staging of gemm_dma_kernels.hip: K-tiles of 32
// (one MFMA k-step) in a ring of four 32 KB stages, the DMAs of K-tile t + 4 issued during K-tile t,
// K-tile t + 1 awaited with a counted s_waitcnt before the one bare barrier of the tile, fragment reads
// (the next tile's while this tile's MFMAs run) and DMAs issued from assembly so that neither
// the compiler's waits nor the vector ALU are involved.  The two-buffer kernel above waits for a DMA one
// K-tile (0.9 us of MFMAs) after issuing it, which is less than the transfer takes under load.
// Images of one operand's K-tile (16 KB either way):
//   K-major  [row][32 k], 64-byte rows, chunk c (8 k) of row r at chunk position c ^ (-(r >> 2) & 3);
//            a fragment is one ds_read_b128.  The sixteen lanes a ds_read_b128 services together hold rows
//            m, 12 + m (chunk kg) and 4 + m, 8 + m (chunk kg ^ 1), m = 0 .. 3; rows with equal m share a
//            64-byte quarter of the 256-byte bank row, and the XOR terms 0, 3, 2, 1 of the four row groups
//            put their chunks on its four different slots (c ^ ((r >> 2) & 3) is 2-way conflicting: measured
//            6 against 3 LDS clocks per read, tools/lds_read_probe.hip).
//   outer    [k][256 rows], 512-byte lines, chunk c (8 rows) of line k at chunk position
//            c ^ (((k & 3) << 2) | (((k >> 3) & 1) << 1)); a fragment is two ds_read_b64_tr_b16 (lines
//            8 g + 0..3 and 8 g + 4..7 of lane group g).  The eight lines a 32-lane half reads together
//            (q = 0..3 of two lane groups) have eight different XOR terms, so their 32-byte pieces of one
//            16-row block fall on eight different bank groups; the term is the same for lines k and k + 4,
//            which makes the second read of a fragment the first one's address + 2048.
constexpr int kRingBK = 32, kRingStages = 4;
constexpr int kRingImage = 256 * kRingBK * 2;  // one operand's tile image
constexpr int kRingStage = 2 * kRingImage;

template <int OFF> __device__ __forceinline__ void lds_read128(u4_t &d, unsigned addr)
{
    static_assert(OFF >= 0 && OFF < 65536 && OFF % 16 == 0, "ds_read offset field");
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF));
}
// The two transposing reads of one outer-image fragment at (addr ^ XOR) + OFF and 2048 bytes behind.  The XOR is
// formed here, next to the reads, so that the compiler does not keep one address register per block alive
// across the whole K loop (the kernel has no registers to spare).
template <int OFF, int XOR> __device__ __forceinline__ void lds_read_tr64x2(u2_t &lo, u2_t &hi, unsigned addr)
{
    static_assert(OFF >= 0 && OFF + 2048 < 65536 && OFF % 8 == 0, "ds_read offset field");
    unsigned tmp;
    asm volatile("v_xor_b32 %2, %4, %3\n\tds_read_b64_tr_b16 %0, %2 offset:%5\n\tds_read_b64_tr_b16 %1, %2 offset:%6"
                 : "=&v"(lo), "=&v"(hi), "=&v"(tmp)
                 : "v"(addr), "n"(XOR), "n"(OFF), "n"(OFF + 2048));
}
// Values read from assembly pass through here after the s_waitcnt that covers them and before their first
// use: the (empty) statement keeps every use behind that wait.
template <typename V> __device__ __forceinline__ void arrived(V &v) { asm volatile("" : "+v"(v)); }

// One operand of the ring kernel as a wave sees it: NB blocks of 16 rows starting at tile row w0.
// init(): tab = the operand's row table, r0 = the tile's first row, R = its extent, k_in = the element step of
// its k table inside a K-tile, image = LDS address of the operand's image in stage 0.
// off[j]: byte offset of this lane's 16 bytes from the K-tile's base, wave instruction j of two (instruction
// 2 wave + j fills bytes [(2 wave + j) KB, + 1 KB) of the image); rows past the extent repeat rows inside it
// (their products are never stored).
// KG: the k table is affine inside aligned groups of KG contracted elements: 32 (the whole K-tile: one table entry
// per tile) or 16 (one entry per half of the tile; waves 0 - 3 fetch the first half of both operands, waves 4 - 7
// the second, so a wave still needs one entry per operand).  A K-major operand then keeps its two halves in two
// images of 32-byte rows [half][row][16 k] (one DMA instruction: 32 rows of one half), which need no swizzle:
// the sixteen lanes a ds_read_b128 services together hold rows m, 12 + m of chunk c and rows 4 + m, 8 + m of
// chunk c ^ 1 of one half, i.e. slots 2 (row & 7) + chunk of the 256-byte bank row, all different.
template <int LAY, int NB, int KG> struct RingOperand;

template <int NB, int KG> struct RingOperand<0, NB, KG> {
    unsigned off[2];
    unsigned addr[2];  // per pair of stages: the offset field of ds_read holds 64 KB
    u4_t frag[2][NB];
    __device__ __forceinline__ void init(const int64_t *__restrict__ tab, int64_t r0, int64_t R, int64_t, unsigned image, int w0,
                                         int wave, int lane)
    {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            // KG 32: 16 rows of 4 chunks per instruction; the row's XOR term: (row >> 2) & 3 = (lane >> 4) & 3
            // KG 16: 32 rows of the 2 chunks of one half (instructions 0 - 7: first half)
            const int row = KG == 32 ? 16 * (2 * wave + j) + (lane >> 2) : 32 * ((2 * wave + j) & 7) + (lane >> 1);
            const int c = KG == 32 ? (lane & 3) ^ ((0 - (lane >> 4)) & 3) : (lane & 1);
            int64_t g = r0 + row;
            if (g > R - 1) g = R - 1;
            off[j] = (unsigned)(tab[g] * 2 + c * 16);
        }
        const int frow = lane & 15, kg = lane >> 4;  // lane (frow, kg) reads chunk kg of its row
#pragma unroll
        for (int h = 0; h < 2; ++h)
            addr[h] = image + 2 * h * kRingStage +
                      (KG == 32 ? (w0 + frow) * 64 + ((kg ^ ((0 - (frow >> 2)) & 3)) << 4)
                                : (kg >> 1) * (kRingImage / 2) + (w0 + frow) * 32 + (kg & 1) * 16);
    }
    template <int SLOT, int S, int B> __device__ __forceinline__ void read()
    {
        lds_read128<(S & 1) * kRingStage + B * (KG == 32 ? 1024 : 512)>(frag[SLOT][B], addr[S >> 1]);
    }
    template <int SLOT> __device__ __forceinline__ void all_arrived()
    {
#pragma unroll
        for (int b = 0; b < NB; ++b) arrived(frag[SLOT][b]);
    }
    template <int SLOT, int B> __device__ __forceinline__ bf8_t get() const { return __builtin_bit_cast(bf8_t, frag[SLOT][B]); }
};

template <int NB, int KG> struct RingOperand<1, NB, KG> {
    unsigned off[2];
    unsigned addr[2];  // block 0, per pair of stages; block b is at addr ^ 32 b (the XOR term of the swizzle mixes with the block index)
    u2_t lo[2][NB], hi[2][NB];
    __device__ __forceinline__ void init(const int64_t *__restrict__ tab, int64_t r0, int64_t R, int64_t k_in, unsigned image,
                                         int w0, int wave, int lane)
    {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int line = 2 * (2 * wave + j) + (lane >> 5);  // 2 lines of 32 chunks per instruction
            const int c = (lane & 31) ^ (((line & 3) << 2) | (((line >> 3) & 1) << 1));
            int64_t g = r0 + 8 * c;
            if (g > R - 8) g = R - 8;  // R is a multiple of 8 with this layout
            off[j] = (unsigned)((tab[g] + (line & (KG - 1)) * k_in) * 2);  // from the table entry of the line's group
        }
        // lane 4 q + pp of lane group g names rows 4 pp .. 4 pp + 3 of line 8 g + q of the block and receives the
        // four lines of row (lane & 15)
        const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
        const unsigned lane_off = 512 * (8 * g + q) + 8 * pp, y32 = 32 * (2 * q + (g & 1));
        // image and stage addresses are multiples of 1 KB, w0 of 16 NB: bits 5 .. 8 of the address hold
        // (w0 / 16 + b) ^ (y32 / 32) and nothing else
#pragma unroll
        for (int h = 0; h < 2; ++h) addr[h] = image + 2 * h * kRingStage + lane_off + ((unsigned)(2 * w0) ^ y32);
    }
    template <int SLOT, int S, int B> __device__ __forceinline__ void read()
    {
        lds_read_tr64x2<(S & 1) * kRingStage, 32 * B>(lo[SLOT][B], hi[SLOT][B], addr[S >> 1]);
    }
    template <int SLOT> __device__ __forceinline__ void all_arrived()
    {
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            arrived(lo[SLOT][b]);
            arrived(hi[SLOT][b]);
        }
    }
    template <int SLOT, int B> __device__ __forceinline__ bf8_t get() const
    {
        const u4_t both = {lo[SLOT][B][0], lo[SLOT][B][1], hi[SLOT][B][0], hi[SLOT][B][1]};
        return __builtin_bit_cast(bf8_t, both);
    }
};

template <int LA, int LB, int KG> __global__ void __launch_bounds__(512, 1) gemm_bf16_ring_kernel(const GemmParams p)
{
    using namespace lds_dma;
    constexpr int BT = 256, BM = BT, BN = BT, TM = 8, TN = 4, NMFMA = TM * TN, NDMA = 4, BK = kRingBK;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[kRingStages * kRingStage];

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm0 = (wave / 4) * (16 * TM), wn0 = (wave % 4) * 64;
#ifdef CTC_CLOCK  // diagnostic build: shader clock while this workgroup runs = core-clock ticks / 100 MHz ticks
    const unsigned long long ck_core0 = __builtin_readcyclecounter(), ck_real0 = __builtin_amdgcn_s_memrealtime();
#endif

    int sid, chunk = 0;
    bool partial = false;
    {
        const int bid = blockIdx.x;
        if (bid < p.dp_tiles) {
            const int nblk = p.dp_tiles;
            const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
            sid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
        } else {
            const int rel = bid - p.dp_tiles;
            sid = p.dp_tiles + rel / p.split;
            chunk = rel % p.split;
            partial = true;
        }
    }
    // a launch without a batch dimension in its grid: the tile id counts through the batch entries (GemmParams::flat_batch)
    const int list_id = sid;  // position in the launch's list of tiles: what a chunk's workspace slab is counted from
    int bz = blockIdx.y;
    if (p.flat_batch > 0) {
        bz = sid / (p.nbm * p.nbn);
        sid -= bz * (p.nbm * p.nbn);
    }
    const int width = p.group * p.nbn;
    const int first_m = (sid / width) * p.group;
    const int gsz = min(p.nbm - first_m, p.group);
    const int tm = first_m + (sid % width) % gsz;
    const int tn = (sid % width) / gsz;
    const int64_t m0 = (int64_t)tm * BM, n0 = (int64_t)tn * BN;

    const char *A = reinterpret_cast<const char *>(static_cast<const unsigned short *>(p.A) + p.bat_a[bz]);
    const char *B = reinterpret_cast<const char *>(static_cast<const unsigned short *>(p.B) + p.bat_b[bz]);
    unsigned short *C = static_cast<unsigned short *>(p.C) + p.bat_c[bz];

    const int nk_total = (int)(p.K / BK);
    const int kt_first = partial ? (int)((int64_t)chunk * nk_total / p.split) : 0;
    const int kt_last = partial ? (int)((int64_t)(chunk + 1) * nk_total / p.split) : nk_total;
    const int nk = kt_last - kt_first;

    const unsigned smem_lds = lds_addr(smem);
    RingOperand<LA, TM, KG> opa;
    RingOperand<LB, TN, KG> opb;
    opa.init(p.row_a, m0, p.M, p.ka_in, smem_lds, wm0, wave, lane);
    opb.init(p.col_b, n0, p.N, p.kb_in, smem_lds + kRingImage, wn0, wave, lane);

    f4_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = f4_t{0.f, 0.f, 0.f, 0.f};

    // K-tiles are requested in order, four ahead; past the chunk's end the requests wrap to its first tiles
    // (finite data nobody multiplies) so that every wave issues the same number of DMAs per tile.  Their k
    // offsets come by scalar loads issued from assembly; asynchronous: k_wait() before the first use.
    int tl = 0;
    int64_t ka_next, kb_next;
    auto k_fetch = [&]() {
        // KG 16: this wave's DMAs all fetch one half of the K-tile (waves 0 - 3 the first): the entry of that half
        const int64_t k0 = (int64_t)(kt_first + tl) * BK + (KG == 16 ? 16 * (wave >> 2) : 0);
        const int64_t *pa = p.k_a + k0, *pb = p.k_b + k0;
        asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=&s"(ka_next) : "s"(pa));
        asm volatile("s_load_dwordx2 %0, %1, 0x0" : "=&s"(kb_next) : "s"(pb));
    };
    auto k_wait = [&]() { asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(ka_next), "+s"(kb_next)); };
    auto advance = [&]() {
        tl = (tl + 1 == nk) ? 0 : tl + 1;
        k_fetch();
    };
    auto issue = [&](int u, unsigned stage, int64_t ka, int64_t kb) {  // stage: LDS address
        const int j = u & 1;
        if (u < 2) dma16_split(A + ka * 2, opa.off[j], stage + (2 * wave + j) * 1024);
        else dma16_split(B + kb * 2, opb.off[j], stage + kRingImage + (2 * wave + j) * 1024);
    };
    k_fetch();
    k_wait();
#pragma unroll
    for (int s = 0; s < kRingStages; ++s) {
#pragma unroll
        for (int u = 0; u < NDMA; ++u) issue(u, smem_lds + s * kRingStage, ka_next, kb_next);
        advance();
        k_wait();
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(3 * NDMA) : "memory");  // K-tile 0 has landed
    __builtin_amdgcn_s_barrier();

    // fragment R of a tile (0 .. 7: A's row blocks, 8 .. 11: B's column blocks) out of stage S into register set
    // SLOT.  A ds_read_b128 moves 1 KB and an MFMA lasts 16 cycles, so the reads of a tile are dealt out behind
    // the MFMAs of the previous one instead of standing in front of its second MFMA.
    auto read_one = [&](auto slot_t, auto stage_t, auto r_t) {
        constexpr int SLOT = decltype(slot_t)::value, S = decltype(stage_t)::value, R = decltype(r_t)::value;
        if constexpr (R < TM) opa.template read<SLOT, S, R>();
        else opb.template read<SLOT, S, R - TM>();
    };
    static_for<0, TM + TN>([&](auto r_t) { read_one(std::integral_constant<int, 0>(), std::integral_constant<int, 0>(), r_t); });

    // One K-tile out of stage S (its fragments were requested during the previous tile): wait (K-tile + 1 has
    // landed: the eight DMAs issued since may be in flight; this wave's fragment reads have returned),
    // barrier (every wave's part of K-tile + 1 is there and nobody reads stage S any more), then 32 MFMAs
    // with the fragment reads of K-tile + 1 and the DMAs of K-tile + 4 (into stage S) dealt out behind them.
    auto tile_step = [&](auto stage_t) {
        constexpr int S = decltype(stage_t)::value, SN = (S + 1) % kRingStages, slot = S & 1;
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NDMA) : "memory");
        __builtin_amdgcn_s_barrier();
        opa.template all_arrived<slot>();
        opb.template all_arrived<slot>();
        k_wait();
        const int64_t ka = ka_next, kb = kb_next;
        const unsigned ld = smem_lds + S * kRingStage;
        static_for<0, NMFMA>([&](auto q_t) {
            constexpr int q = decltype(q_t)::value;
            acc[q / TN][q % TN] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(opa.template get<slot, q / TN>(), opb.template get<slot, q % TN>(),
                                                                         acc[q / TN][q % TN], 0, 0, 0);
#ifndef CTC_BF16_RING_READ_EVERY
#define CTC_BF16_RING_READ_EVERY 2
#endif
#ifndef CTC_BF16_RING_DMA_FIRST
#define CTC_BF16_RING_DMA_FIRST 14  // behind the reads of A: DMAs among transposing reads cost up to half the rate
#endif
            constexpr int EVERY = CTC_BF16_RING_READ_EVERY, DF = CTC_BF16_RING_DMA_FIRST;
            if constexpr (q % EVERY == 0 && q / EVERY < TM + TN)
                read_one(std::integral_constant<int, slot ^ 1>(), std::integral_constant<int, SN>(), std::integral_constant<int, q / EVERY>());
            if (q >= DF && q < DF + 4 * NDMA && (q - DF) % 4 == 0) issue((q - DF) / 4, ld, ka, kb);
            if (q == 29) advance();
            __builtin_amdgcn_sched_barrier(0);
        });
    };
    {
        int kt = 0;
        for (; kt + kRingStages <= nk; kt += kRingStages) {
            tile_step(std::integral_constant<int, 0>());
            tile_step(std::integral_constant<int, 1>());
            tile_step(std::integral_constant<int, 2>());
            tile_step(std::integral_constant<int, 3>());
        }
        if (kt < nk) tile_step(std::integral_constant<int, 0>());
        if (kt + 1 < nk) tile_step(std::integral_constant<int, 1>());
        if (kt + 2 < nk) tile_step(std::integral_constant<int, 2>());
    }
    // the requests past the end have landed before the epilogue takes the stages over, and the last
    // fragment reads (of a tile nobody multiplies) have returned
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    opa.template all_arrived<0>();
    opa.template all_arrived<1>();
    opb.template all_arrived<0>();
    opb.template all_arrived<1>();
    __syncthreads();
#ifdef CTC_CLOCK
    if (t == 0 && blockIdx.y == 0 && (blockIdx.x == 100 || blockIdx.x == 500 || blockIdx.x == 900)) {
        const unsigned long long dc = __builtin_readcyclecounter() - ck_core0, dr = __builtin_amdgcn_s_memrealtime() - ck_real0;
        printf("[clock] block %d: %llu core ticks in %llu x 10 ns -> %.3f GHz (K loop only)\n", (int)blockIdx.x, dc, dr, (double)dc / (10.0 * (double)dr));
    }
#endif
    store_tile_bf16<BT, TM, TN>(p, acc, smem, C, m0, n0, wm0, wn0, t, lane, partial, list_id, chunk);
}

// both operands read in 16-byte vectors, the 256 tile, K a multiple of 32 with k offsets affine inside aligned
// groups of 32 (returns 32) or of 16 (returns 16; not for two outer-major operands), step 1 for a K-major operand,
// lane offsets that fit 32 bits; 0: the two-buffer kernel
int ring_group(int la, int lb, int bt, const GemmParams &p)
{
    static const bool off = getenv("CTC_TC_NO_BF16_RING") != nullptr;
    static const bool no_outer = getenv("CTC_TC_NO_BF16_RING_OUTER") != nullptr;
    static const bool no_k16 = getenv("CTC_TC_NO_BF16_RING_K16") != nullptr;
    if (off || bt != 256 || la == kLayScalar || lb == kLayScalar || (no_outer && (la != 0 || lb != 0))) return 0;
    if (p.K < kRingBK || p.K % kRingBK != 0) return 0;
    const int64_t step_max = int64_t(1) << 22;  // 31 lines of a K-tile stay inside the 32-bit lane offset
    if (la == 0 ? p.ka_in != 1 : (p.ka_in < 0 || p.ka_in > step_max || p.M < 8)) return 0;
    if (lb == 0 ? p.kb_in != 1 : (p.kb_in < 0 || p.kb_in > step_max || p.N < 8)) return 0;
    if (p.k32_affine != 0) return 32;
    // (two outer-major operands with groups of 16: that instantiation does not fit the register file)
    if (p.k16_affine != 0 && !no_k16 && !(la == 1 && lb == 1)) return 16;
    return 0;
}

// Sums the K chunks of the split tiles in chunk order and writes C = alpha * sum + beta * C in bf16.
// One workgroup per (tile, 16 rows); a thread owns two neighbouring columns.
template <int BT> __global__ void __launch_bounds__(256) splitk_fixup_bf16_kernel(const GemmParams p)
{
    constexpr int BM = BT, BN = BT;
    constexpr int ROWS = 16, VPR = BN / 2, RPP = 256 / VPR;  // column pairs per row, rows per pass
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
    const float *slab0 = static_cast<const float *>(p.ws) + (int64_t)tile_rel * p.split * (BM * BN);
    unsigned short *C = static_cast<unsigned short *>(p.C) + p.bat_c[bz];
    const float alpha = (float)p.alpha, beta = (float)p.beta;
    const int cv = threadIdx.x % VPR;
    const int64_t n = (int64_t)tn * BN + 2 * cv;
    const bool ok0 = n < p.N, ok1 = n + 1 < p.N;
    const int64_t c0 = p.col_c[ok0 ? n : 0], c1 = p.col_c[ok1 ? n + 1 : 0];
    for (int rr = threadIdx.x / VPR; rr < ROWS; rr += RPP) {
        const int row = rblk * ROWS + rr;
        const int64_t m = (int64_t)tm * BM + row;
        if (m >= p.M) continue;
        // eight chunks at a time into separate sums (eight loads in flight); fixed order of additions
        float a0[8], a1[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) a0[u] = a1[u] = 0.f;
        int c = 0;
        for (; c + 8 <= p.split; c += 8) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const float *src = slab0 + (int64_t)(c + u) * (BM * BN) + row * BN + 2 * cv;
                a0[u] += src[0];
                a1[u] += src[1];
            }
        }
        for (int u = 0; c < p.split; ++c, ++u) {
            const float *src = slab0 + (int64_t)c * (BM * BN) + row * BN + 2 * cv;
            a0[u] += src[0];
            a1[u] += src[1];
        }
        const float s0 = ((a0[0] + a0[1]) + (a0[2] + a0[3])) + ((a0[4] + a0[5]) + (a0[6] + a0[7]));
        const float s1 = ((a1[0] + a1[1]) + (a1[2] + a1[3])) + ((a1[4] + a1[5]) + (a1[6] + a1[7]));
        const int64_t ro = p.row_c[m];
        if (ok0) {
            float v = alpha * s0;
            if (beta != 0.f) v += beta * bf16_to_f32(C[ro + c0]);
            C[ro + c0] = f32_to_bf16(v);
        }
        if (ok1) {
            float v = alpha * s1;
            if (beta != 0.f) v += beta * bf16_to_f32(C[ro + c1]);
            C[ro + c1] = f32_to_bf16(v);
        }
    }
}

// workgroup slots of the current device: two workgroups per CU with the 128 tile (64 KB of LDS
// each), one with the 256 tile (128 KB)
template <int BT> int bf16_slots()
{
    static std::atomic<int> cache[64];  // 0: not queried yet; host threads of several devices may race here
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 0;
    if (cache[dev].load() == 0) {
        int cus = 0, per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemm_bf16_kernel<0, 0, BT>, 2 * BT, 0) != hipSuccess) return 0;
        cache[dev].store(cus * per_cu > 0 ? cus * per_cu : -1);
    }
    const int slots = cache[dev].load();
    return slots > 0 ? slots : 0;
}

constexpr int kMinChunkTilesBf16 = 16;  // K-tiles (of 64) per chunk at least: shorter chunks cost more in the second launch than they save

// The 256 tile exists for the 16-byte-vector layouts only; a request for it with an operand
// that has to be read element by element (the planner never makes one; an unaligned base can
// demote a layout at execution time) runs on the 128 tile.
int tile_edge(int la, int lb, int bm) { return bm == 256 && la != kLayScalar && lb != kLayScalar ? 256 : 128; }

template <int LA, int LB, int BT>
hipError_t launch_one(const GemmParams &p, const GemmSchedule &sch, void *ws, hipStream_t stream)
{
    GemmParams q = p;
    // diagnostic: CTC_TC_BF16_DIRECT_STORE keeps the quads' direct 8-byte stores (no pass through LDS)
    static const bool direct_store = getenv("CTC_TC_BF16_DIRECT_STORE") != nullptr;
    if (direct_store && q.c_vec > 4) q.c_vec = 4;
    q.nbm = sch.nbm;
    q.nbn = sch.nbn;
    q.dp_tiles = sch.dp_tiles;
    q.split = sch.split;
    q.flat_batch = sch.flat_batch;
    q.ws = ws;
    if (q.group < 1) q.group = 1;
    if (sch.tail_tiles > 0 && ws == nullptr) return hipErrorInvalidValue;
    if (sch.nbm != (int)((p.M + BT - 1) / BT) || sch.nbn != (int)((p.N + BT - 1) / BT)) return hipErrorInvalidValue;
    hipLaunchKernelGGL((gemm_bf16_kernel<LA, LB, BT>), dim3((unsigned)(sch.dp_tiles + sch.tail_tiles * sch.split), sch.flat_batch > 0 ? 1u : (unsigned)p.batch, 1),
                       dim3(2 * BT), 0, stream, q);
    if (sch.tail_tiles > 0)
        hipLaunchKernelGGL(splitk_fixup_bf16_kernel<BT>, dim3((unsigned)(sch.tail_tiles * (BT / 16))), dim3(256), 0, stream, q);
    return hipGetLastError();
}

std::atomic<int64_t> g_ring_launches{0};

template <int LA, int LB, int KG> hipError_t launch_ring(const GemmParams &p, const GemmSchedule &sch, void *ws, hipStream_t stream)
{
    constexpr int BT = 256;
    g_ring_launches.fetch_add(1);
    GemmParams q = p;
    q.nbm = sch.nbm;
    q.nbn = sch.nbn;
    q.dp_tiles = sch.dp_tiles;
    q.split = sch.split;
    q.flat_batch = sch.flat_batch;
    q.ws = ws;
    if (q.group < 1) q.group = 1;
    if (sch.tail_tiles > 0 && ws == nullptr) return hipErrorInvalidValue;
    if (sch.nbm != (int)((p.M + BT - 1) / BT) || sch.nbn != (int)((p.N + BT - 1) / BT)) return hipErrorInvalidValue;
    hipLaunchKernelGGL((gemm_bf16_ring_kernel<LA, LB, KG>), dim3((unsigned)(sch.dp_tiles + sch.tail_tiles * sch.split), sch.flat_batch > 0 ? 1u : (unsigned)p.batch, 1), dim3(2 * BT),
                       0, stream, q);
    if (sch.tail_tiles > 0)
        hipLaunchKernelGGL(splitk_fixup_bf16_kernel<BT>, dim3((unsigned)(sch.tail_tiles * (BT / 16))), dim3(256), 0, stream, q);
    return hipGetLastError();
}

}  // namespace

// Same policy as the f64/f32 kernel: tiles = q * slots + tail; the q full rounds run whole, each
// tail tile is cut along K into floor(slots / tail) chunks.  Problems with fewer tiles than
// slots are split entirely, which is what keeps medium-sized contractions on the whole chip.
hipError_t schedule_gemm_bf16(int la, int lb, int bm, const GemmParams &p, GemmSchedule *out)
{
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.batch <= 0 || p.batch > 65535) return hipErrorInvalidValue;
    const int bt = tile_edge(la, lb, bm);
    GemmSchedule s;
    s.nbm = (int)((p.M + bt - 1) / bt);
    s.nbn = (int)((p.N + bt - 1) / bt);
    int64_t tiles = (int64_t)s.nbm * s.nbn;
    if (tiles > 0x7FFFFFFF) return hipErrorInvalidValue;
    // a batched launch: one list of tiles over all entries whenever its length fits a grid, cut like a single
    // product's (CTC_TC_NO_FLAT_BATCH: A/B runs)
    static const bool no_flat = getenv("CTC_TC_NO_FLAT_BATCH") != nullptr;
    if (p.batch > 1 && !no_flat && tiles * p.batch <= 0x3FFFFF) {
        s.flat_batch = p.batch;
        tiles *= p.batch;
    }
    s.dp_tiles = (int)tiles;
    s.split = 1;
    s.tail_tiles = 0;
    s.ws_bytes = 0;
    static const bool no_split = getenv("CTC_TC_NO_SPLITK") != nullptr;
    const int slots = bt == 256 ? bf16_slots<256>() : bf16_slots<128>();
    const int nk = (int)((p.K + BK - 1) / BK);
    if (!no_split && slots > 0 && (p.batch == 1 || s.flat_batch > 0)) {
        const int tail = (int)(tiles % slots);
        if (tail > 0) {
            // f chunks per tail tile take ceil(tail * f / slots) rounds of (nk / f + overhead) K-tile
            // times (overhead of a chunk: kMinChunkTilesBf16 K-tiles); f minimises that, as in the
            // f64 / f32 scheduler
            int fmax = nk / kMinChunkTilesBf16;
            if (fmax > 1024) fmax = 1024;  // one tile with a very long K may use every slot
            int f = 1;
            double best = (double)nk;
            for (int c = 2; c <= fmax; ++c) {
                const double rounds = (double)(((int64_t)tail * c + slots - 1) / slots);
                const double cost = rounds * ((double)nk / c + kMinChunkTilesBf16);
                if (cost < best * 0.98) {
                    best = cost;
                    f = c;
                }
            }
            if (f >= 2) {
                s.tail_tiles = tail;
                s.dp_tiles = (int)tiles - tail;
                s.split = f;
                s.ws_bytes = (size_t)tail * f * bt * bt * sizeof(float);
            }
        }
    }
    *out = s;
    return hipSuccess;
}

int64_t bf16_ring_launches() { return g_ring_launches.load(); }

hipError_t launch_gemm_bf16(int la, int lb, int bm, const GemmParams &p, const GemmSchedule &sch, void *ws,
                            hipStream_t stream)
{
    if (la < 0 || la > 2 || lb < 0 || lb > 2) return hipErrorInvalidValue;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.batch <= 0 || p.batch > 65535) return hipErrorInvalidValue;
    const int group = ring_group(la, lb, tile_edge(la, lb, bm), p);  // 0 unless both layouts are 0 or 1
    switch (group == 0 ? -1 : (group / 16) * 4 + la * 2 + lb) {     // group 16: 4 .. 7, 32: 8 .. 11
    case 4: return launch_ring<0, 0, 16>(p, sch, ws, stream);
    case 5: return launch_ring<0, 1, 16>(p, sch, ws, stream);
    case 6: return launch_ring<1, 0, 16>(p, sch, ws, stream);
    case 8: return launch_ring<0, 0, 32>(p, sch, ws, stream);
    case 9: return launch_ring<0, 1, 32>(p, sch, ws, stream);
    case 10: return launch_ring<1, 0, 32>(p, sch, ws, stream);
    case 11: return launch_ring<1, 1, 32>(p, sch, ws, stream);
    default: break;
    }
    if (tile_edge(la, lb, bm) == 256) {
        switch (la * 2 + lb) {
        case 0: return launch_one<0, 0, 256>(p, sch, ws, stream);
        case 1: return launch_one<0, 1, 256>(p, sch, ws, stream);
        case 2: return launch_one<1, 0, 256>(p, sch, ws, stream);
        default: return launch_one<1, 1, 256>(p, sch, ws, stream);
        }
    }
#define CTC_CASE(A_, B_) \
    case A_ * 3 + B_: return launch_one<A_, B_, 128>(p, sch, ws, stream);
    switch (la * 3 + lb) {
        CTC_CASE(0, 0) CTC_CASE(0, 1) CTC_CASE(0, 2) CTC_CASE(1, 0) CTC_CASE(1, 1) CTC_CASE(1, 2)
        CTC_CASE(2, 0) CTC_CASE(2, 1) CTC_CASE(2, 2)
    default: return hipErrorInvalidValue;
    }
#undef CTC_CASE
}

}  // namespace ctc
