// Contraction planner: host-only, no HIP types.  Turns an einsum-style spec plus
// extents/strides into (1) a mode classification, (2) the canonical GEMM view and
// its transposes, (3) the execution view used on the device: operand roles, mode
// orders, per-dimension element-offset tables, operand layouts and a tile.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace ctc {

constexpr int kMaxModes = 16;

enum Dtype { kF64 = 0, kF32 = 1, kBF16 = 2 };
enum ModeClass { kBatch = 0, kContracted = 1, kFreeM = 2, kFreeN = 3 };
enum Layout { kLayoutKVec = 0, kLayoutOuterVec = 1, kLayoutScalar = 2 };
enum KernelKind { kKernelGemm = 0, kKernelSkinnySame = 1, kKernelSkinnyTranspose = 2 };
constexpr int kSkinnyMaxM = 16;
constexpr int kSkinnyMaxMK = 2048;  // elements of the small operand kept in LDS

inline int dtype_size(int dtype) { return dtype == kF64 ? 8 : (dtype == kF32 ? 4 : 2); }

struct Mode {
    char name = 0;
    int cls = 0;
    int64_t extent = 0;
    // element strides in the ORIGINAL operands (0 when the operand lacks the mode)
    int64_t sa = 0, sb = 0, sc = 0;
    // positions in the original operands (-1 when absent)
    int pa = -1, pb = -1, pc = -1;
};

struct Tile {
    int bm = 128, bn = 128, bk = 16;
    int threads = 256;
    int group = 8;  // tile rows per L2-sharing group in the block order
};

struct Plan {
    std::string spec;
    int dtype = kF64;
    std::vector<Mode> modes;           // distinct letters in order of first appearance
    std::vector<int> a_modes, b_modes, c_modes;  // mode ids per operand position
    std::vector<int64_t> strides_a, strides_b, strides_c;  // element strides per operand position

    // canonical view (slowest first), independent of strides
    std::vector<int> canon_batch, canon_m, canon_n, canon_k;
    std::vector<int> perm_a, perm_b, perm_c;

    // execution view (fastest first).  When `swapped`, GEMM operand "A" is the
    // caller's B and vice versa, so that C's unit-stride mode is an N mode.
    bool swapped = false;
    std::vector<int> ex_batch, ex_m, ex_n, ex_k;
    int64_t M = 1, N = 1, K = 1, batch = 1;
    int layout_a = kLayoutScalar, layout_b = kLayoutScalar;
    int vec_a = 1, vec_b = 1;
    int vec_c = 1;  // elements per 16-byte store of C along its unit-stride N mode (1: element stores)
    // every contiguous run of C a tile stores begins and ends on a 64-byte boundary when C's base does: the
    // unit-stride N mode's extent and every other stride of C are multiples of 64 bytes.  Only then do stores
    // with the streaming (non-temporal) hint pay: pieces of a 64-byte sector that arrive from different
    // workgroups are merged by the L2, which the hint bypasses.
    bool c_sectors_whole = false;
    Tile tile;
    std::vector<int64_t> row_a, k_a, k_b, col_b, row_c, col_c, bat_a, bat_b, bat_c;
    // k_a[k] == k * ka_stride for every k (the contracted modes fuse into one stride); 0 when they do not
    int64_t ka_stride = 0, kb_stride = 0;
    // k8_affine: K is a multiple of 8 and inside every aligned group of 8 contracted indices both k tables
    // are affine with steps ka_in / kb_in (true when the fastest contracted mode's extent is a multiple
    // of 8): a kernel that walks K in tiles of 8 needs one table entry per tile
    bool k4_affine = false, k8_affine = false, k16_affine = false, k32_affine = false;  // groups of 4 / 8 / 16 / 32
    int64_t ka_in = 0, kb_in = 0;
    // length of the runs of k inside which both k tables are affine with those steps (the extent of the
    // fastest contracted mode; K is a multiple of it); 0: none
    int64_t k_run = 0;
    // largest row offset of A / column offset of B (elements)
    int64_t max_row_a = 0, max_col_b = 0;

    // Skinny path (M <= 16, K small, N large): memory-bound, no MFMA.  kSkinnySame:
    // B and C share their unit-stride N mode, vectors run along the fused N index and
    // col_b / col_c are used as they are.  kSkinnyTranspose: they differ; N is split
    // into x (contiguous in B), y (contiguous in C) and o (the rest) with one offset
    // table per operand, exactly as in a mode permutation.
    int kernel = 0;  // KernelKind
    // The fused kernel would read all three tensors element by element at a stride (a batch mode is their
    // unit-stride mode): ctc_tc_execute transposes the operands explicitly instead (packed operands, beta == 0).
    bool prefer_transposes = false;
    // An operand would be read element by element (an odd extent on its unit-stride mode) under a contraction
    // large enough to pay for two passes over A and B: ctc_tc_execute first copies both into K-major rows
    // [batch, M, K] / [batch, N, K] whose length is padded to k_padded (a multiple of 64 bytes, zero-filled),
    // so that the GEMM kernel with LDS-DMA staging runs on them; C is written in place (any alpha, beta).
    bool pad_transposes = false;
    int64_t k_padded = 0;
    int64_t sk_X = 1, sk_Y = 1, sk_O = 1;
    std::vector<int64_t> sk_bx, sk_cx, sk_by, sk_cy, sk_bo, sk_co;

    int64_t size_a = 0, size_b = 0, size_c = 0;  // packed element counts (informational)
};

// All functions return "" on success or an error message.
std::string parse_and_classify(const std::string &spec, Plan &plan);
std::string build_plan(const std::string &spec, int n_ext, const char *mode_names,
                       const int64_t *extents, const int64_t *strides_a, const int64_t *strides_b,
                       const int64_t *strides_c, int dtype, Plan &plan);
// Slab [rank*ext/nranks, (rank+1)*ext/nranks) of `mode`; offsets in elements.
std::string shard_plan(const Plan &full, char mode, int rank, int nranks, Plan &shard,
                       int64_t &off_a, int64_t &off_b, int64_t &off_c);
std::string describe(const Plan &plan);

}  // namespace ctc
