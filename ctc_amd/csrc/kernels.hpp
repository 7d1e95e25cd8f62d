// Device-side entry points shared between the C ABI (api.cpp) and the kernel files.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

namespace ctc {

// The device copies of k_a and k_b carry this many extra entries (repeating the last
// whole 16-byte vector's entries) so that the kernel's look-ahead reads past K need
// no clamping and stay inside the operand.  The furthest look-ahead is the prologue's third
// K-tile (entries up to 3 * BK - 1 whatever K is), with BK <= 32 for f64/f32 and one tile of 64
// for bf16.
constexpr int kKTablePad = 128;

// One GEMM-shaped contraction.  All tables hold ELEMENT offsets:
//   A(m,k) = A[bat_a[b] + row_a[m] + k_a[k]],  B(k,n) = B[bat_b[b] + k_b[k] + col_b[n]],
//   C(m,n) = C[bat_c[b] + row_c[m] + col_c[n]].
struct GemmParams {
    const void *A = nullptr;
    const void *B = nullptr;
    void *C = nullptr;
    const int64_t *row_a = nullptr, *k_a = nullptr, *k_b = nullptr, *col_b = nullptr;
    const int64_t *row_c = nullptr, *col_c = nullptr;
    const int64_t *bat_a = nullptr, *bat_b = nullptr, *bat_c = nullptr;
    int64_t M = 0, N = 0, K = 0;
    int batch = 1;
    double alpha = 1.0, beta = 0.0;
    int nbm = 0, nbn = 0, group = 8;
    // schedule (filled by launch_gemm): tiles [0, dp_tiles) run whole, every later tile is cut
    // into `split` K chunks whose raw accumulators go to `ws`
    int dp_tiles = 0, split = 1;
    // sk_units > 0 (LDS-DMA kernel only, instead of `split`): the K-tile iterations of the tiles from dp_tiles
    // on, taken as ONE sequence (tile after tile, K-tile after K-tile), are dealt out to sk_units workgroups in
    // equal contiguous shares; a share's pieces of tiles go to C when they start a tile and to slab `unit` of
    // `ws` otherwise (see sk_unit_start in gemm_dma_kernels.hip)
    // sk_group: the shares are formed over GROUPS of that many consecutive tiles; the sk_group workgroups of a
    // group of units walk the same K-tile range of their sk_group tiles together, so that they keep sharing
    // operand panels in the L2 as whole tiles of a round do (sk_units is a multiple of it)
    int sk_units = 0, sk_group = 1;
    // flat_batch > 0 (fp64 / fp32 tiled kernels): the launch's grid has no batch dimension; the tiles of its flat_batch
    // batch entries form ONE list (entry after entry, nbm * nbn tiles each) that dp_tiles, split and sk_units
    // refer to, so that the last round of a batched launch is balanced like that of a single product
    int flat_batch = 0;
    // plain_order != 0 (set by the fp64 / fp32 launchers for a flat batch of one tile per entry): workgroup i takes
    // tile i instead of XCD-contiguous runs of tile ids - such tiles share no operand panel, and neighbouring
    // workgroups on neighbouring entries keep the DRAM pages they stream from in step (4096 x 64^3 fp64: 24.6
    // TFLOP/s against 21.9 with the runs)
    int plain_order = 0;
    void *ws = nullptr;
    int c_vec = 1;  // > 1: C may be stored in 16-byte vectors of that many elements along N
    int c_stream = 0;  // 1: C is larger than the Infinity Cache: store it with the non-temporal hint
    // k_a[k] == k * ka_stride for every k (likewise k_b): the operand's k offsets need no table
    // (0: they do).  Set by the launcher from the plan.
    int64_t ka_stride = 0, kb_stride = 0;
    // k8_affine != 0: K is a multiple of 8 and inside every aligned group of 8 the k offsets are affine,
    // k_a[8 g + i] == k_a[8 g] + i * ka_in (likewise k_b, kb_in): a K-tile of 8 needs one table entry
    // k4_affine, k16_affine, k32_affine: the same for groups of 4 / 16 / 32
    int k4_affine = 0, k8_affine = 0, k16_affine = 0, k32_affine = 0;
    int64_t ka_in = 0, kb_in = 0;
    // k_run > 0: K is a multiple of k_run and the tables are affine (same steps) inside every aligned run of
    // that many k (runs of the fastest contracted mode)
    int64_t k_run = 0;
};

struct GemmSchedule {
    int nbm = 0, nbn = 0;
    int dp_tiles = 0, tail_tiles = 0, split = 1;
    int sk_units = 0, sk_group = 1;  // > 0: the tail_tiles tiles are shared out in equal runs of K-tiles (GemmParams::sk_units, sk_group)
    int flat_batch = 0;   // GemmParams::flat_batch
    int walk = 0;         // 1: one workgroup per slot walks the items (the LDS-DMA kernel's launch decides otherwise by size)
    size_t ws_bytes = 0;  // workspace the launch needs (0: none)
};

// dtype: 0 f64, 1 f32, 2 bf16.  layouts: see ctc_tc_layout.  bm/bn select the tile.
// Returns hipSuccess or the launch error; hipErrorInvalidValue for an
// unsupported combination.
hipError_t schedule_gemm(int dtype, int layout_a, int layout_b, int bm, int bn, const GemmParams &p,
                         GemmSchedule *out);
hipError_t launch_gemm(int dtype, int layout_a, int layout_b, int bm, int bn, const GemmParams &p,
                       const GemmSchedule &schedule, void *workspace, hipStream_t stream);

// Persistent variant for a short K under a large C (gemm_stream_kernels.hip): stores of a finished tile
// are dealt out behind the MFMAs of the next one.  f64 / f32, 128 x 128 tiles, vector layouts,
// 16-byte stores of C, beta == 0, batch == 1; stream_gemm_supported says whether a launch qualifies.
bool stream_gemm_supported(int dtype, int layout_a, int layout_b, int bm, int bn, const GemmParams &p);
hipError_t launch_gemm_stream(int dtype, int layout_a, int layout_b, const GemmParams &p, hipStream_t stream);
// Its fp64 form with every global -> LDS transfer done by LDS-DMA (gemm_stream_f64_kernels.hip); needs, in
// addition, even M and N and k offsets that are multiples of k (GemmParams::ka_stride / kb_stride).
bool stream_gemm_f64_dma_supported(int layout_a, int layout_b, const GemmParams &p);
hipError_t launch_gemm_stream_f64_dma(int layout_a, int layout_b, const GemmParams &p, hipStream_t stream);

// fp64 / fp32, 128 x 128 tiles, operands staged by LDS-DMA (gemm_dma_kernels.hip); schedule_gemm / launch_gemm
// route to it when dma_gemm_supported says the launch qualifies (vector layouts, k tables affine in aligned
// groups of at least half a K-tile: k4_affine / k8_affine / k16_affine).
bool dma_gemm_supported(int dtype, int layout_a, int layout_b, int bm, int bn, const GemmParams &p);
// how its schedules cut the tiles of the last round: 0 by cost (default), 1 never into stream-K shares, 2 into
// stream-K shares whenever the launch allows (process-wide, for measurements and tests)
void dma_gemm_set_stream_k(int mode);
int dma_gemm_get_stream_k();
// tiles per group of stream-K shares (GemmParams::sk_group): 1 .. 64; 0 (default): 16, or 8 where that balances better
void dma_gemm_set_stream_k_group(int g);
int dma_gemm_get_stream_k_group();
// which form of it the launch takes: 0 none (the register-staged kernel runs), 1 K-tiles of 64-byte rows,
// 2 two images of 32-byte rows per stage, 3 one image of 32-byte rows, 4 64-byte rows with ragged K-tiles
int dma_gemm_kind(int dtype, int layout_a, int layout_b, int bm, int bn, const GemmParams &p);
hipError_t schedule_gemm_dma(int dtype, int layout_a, int layout_b, const GemmParams &p, GemmSchedule *out);
hipError_t launch_gemm_dma(int dtype, int layout_a, int layout_b, const GemmParams &p, const GemmSchedule &schedule,
                           void *workspace, hipStream_t stream);
// the split-K fix-up pass of 128 x 128 tiles, dtype 0 / 1 (gemm_kernels.hip); p carries the schedule
hipError_t launch_splitk_fixup_128(int dtype, const GemmParams &p, int tail_tiles, hipStream_t stream);

// bf16 operands and result, f32 accumulation (gemm_bf16_kernels.hip): square tiles, bm = 128 or 256
// (256 only with 16-byte-vector layouts on both operands, otherwise 128 is used), K-tile 64;
// reached through schedule_gemm / launch_gemm with dtype 2.
hipError_t schedule_gemm_bf16(int layout_a, int layout_b, int bm, const GemmParams &p, GemmSchedule *out);
hipError_t launch_gemm_bf16(int layout_a, int layout_b, int bm, const GemmParams &p, const GemmSchedule &schedule,
                            void *workspace, hipStream_t stream);
// launches, since the library was loaded, that took the 256-tile ring kernel (two K-major operands, K a
// multiple of 32, k32_affine) instead of the two-buffer kernel
int64_t bf16_ring_launches();

// Skinny contraction C(m,n) = sum_k A(m,k) B(k,n) for M <= 16: streams B once and C once.
// transpose == 0: n is the fused N index, B(k,n) = B[k_b[k] + nb[n]], C(m,n) = C[row_c[m] + nc[n]]
//                 with nb = col_b, nc = col_c and X = N.
// transpose == 1: n = (x, y, o); B offset = bx[x] + by[y] + bo[o], C offset = cx[x] + cy[y] + co[o].
struct SkinnyParams {
    const void *A = nullptr;
    const void *B = nullptr;
    void *C = nullptr;
    const int64_t *row_a = nullptr, *k_a = nullptr, *k_b = nullptr, *row_c = nullptr;
    const int64_t *bat_a = nullptr, *bat_b = nullptr, *bat_c = nullptr;
    const int64_t *bx = nullptr, *cx = nullptr, *by = nullptr, *cy = nullptr, *bo = nullptr, *co = nullptr;
    int64_t X = 1, Y = 1, O = 1;
    int M = 1, K = 1, batch = 1;
    int transpose = 0;
    int stream_hint = 0;  // 1: the large operand and C exceed the Infinity Cache and C's runs cover whole 64-byte sectors
    double alpha = 1.0, beta = 0.0;
};
hipError_t launch_skinny(int dtype, const SkinnyParams &p, hipStream_t stream);

// One small row-major GEMM of a grouped launch; layout shared with ctc_tc_small_gemm (ctc_tc.h).
struct SmallGemm {
    int32_t m, n, k;
    int32_t lda, ldb, ldc;
    int64_t off_a, off_b, off_c;
};
// a_outer: every A is stored [k][m] (lda: stride of k); b_kmajor: every B is stored [n][k] (ldb: stride of a column)
hipError_t launch_grouped(int dtype, const SmallGemm *problems_device, int count, const void *A, const void *B,
                          void *C, double alpha, double beta, hipStream_t stream, bool a_outer = false, bool b_kmajor = false,
                          bool small_equal = false);  // small_equal: every problem has the first one's m, n <= 16, k and leading dimensions

// FASTR scoring: one work item = one consensus x up to fastr_reads_per_block() reads.
struct FastrWork {
    int64_t ref_off;     // first byte of the consensus in ref
    int32_t ref_len;
    int32_t n_reads;     // reads handled by this workgroup
    int64_t first_read;  // global index of its first read
    int64_t score_off;   // index of its first result
};
struct FastrParams {
    const unsigned char *ref = nullptr, *read_seq = nullptr, *quals = nullptr;
    const int32_t *read_len = nullptr, *original_alignment = nullptr;
    const FastrWork *work = nullptr;
    int32_t *best_score = nullptr, *best_index = nullptr;
    int read_stride = 0;
};
int fastr_max_ref_lds();
int fastr_max_read_lds();
int fastr_reads_per_block();
hipError_t launch_fastr(const FastrParams &p, int n_blocks, bool staged, hipStream_t stream);

// Batched pairwise alignment (sw_kernels.hip): pair q of a launch is global pair first_pair + q.
struct SwParams {
    const unsigned char *ref = nullptr, *reads = nullptr;
    const int64_t *ref_off = nullptr;
    const int32_t *ref_len = nullptr, *read_len = nullptr;
    int read_stride = 0;
    long long first_pair = 0;
    int n_pairs = 0;
    double w_match = 0, w_mismatch = 0, w_open = 0, w_extend = 0;
    int max_ref_len = 0, max_read_len = 0, max_ops = 0;
    short *bt = nullptr;        // back-track store, bt_per_pair int16 per pair of the launch
    size_t bt_per_pair = 0;
    int32_t *offset = nullptr, *n_ops = nullptr, *op_len = nullptr;
    unsigned char *op_type = nullptr;
};
size_t sw_lds_bytes(int max_read_len, int max_ops);
size_t sw_bt_elems_per_pair(int max_ref_len, int max_read_len);
hipError_t launch_sw(const SwParams &p, hipStream_t stream);

// Generic tiled permutation through element-offset tables (see permute_kernels.hip).
struct PermuteParams {
    const void *in = nullptr;
    void *out = nullptr;
    // x: fused index that is contiguous in the INPUT; y: fused index contiguous in the OUTPUT;
    // o: everything else.
    const int64_t *in_x = nullptr, *out_x = nullptr;
    const int64_t *in_y = nullptr, *out_y = nullptr;
    const int64_t *in_o = nullptr, *out_o = nullptr;
    int64_t X = 1, Y = 1, O = 1;
    int same_fast = 0;  // 1: input and output share their fastest mode (row copy)
    int vec_mode = 0;  // for the element size the tables were built for: 2 vectors on both sides, 1 vectors with
                       // element tails (one mode on a ragged side), 0 element accesses (see PermutePlan)
    int o_per_block = 4;        // consecutive o values one workgroup of the vector kernel walks
    int64_t in_run = 1, out_run = 1;  // contiguous elements of the input along x / of the output along y
    int nt_loads = 0, nt_stores = 0;  // streaming hints, set by launch_permute
};
hipError_t launch_permute(int elem_size, const PermuteParams &p, hipStream_t stream);

hipError_t launch_fill(void *dev, int64_t n, uint64_t seed, int64_t first, int dtype,
                       hipStream_t stream);

}  // namespace ctc
