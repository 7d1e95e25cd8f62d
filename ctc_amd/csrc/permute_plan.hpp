// Host-side planning of a mode permutation: fuses modes, picks the x/y/o split
// and builds the element-offset tables consumed by the permute kernels.
#pragma once

#include <cstdint>
#include <string>
#include <vector>

namespace ctc {

struct PermutePlan {
    int64_t X = 1, Y = 1, O = 1;
    bool same_fast = false;
    std::vector<int64_t> in_x, out_x, in_y, out_y, in_o, out_o;
    int64_t total = 0;
    // extents of the (fused) fastest input and output modes
    int64_t fast_in_extent = 1, fast_out_extent = 1;
    // what the element size allows (V = 16 / element size elements per vector):
    //   2  V divides both fastest extents: 16-byte vectors on both sides, aligned whenever the base
    //      addresses are (every other stride is then a multiple of V too)
    //   1  a side's fastest extent is not a multiple of V: that side is ONE mode, moved in vectors at
    //      element-aligned addresses with an element tail per run (4- and 8-byte elements only)
    //   0  element accesses
    int vec_mode = 0;
    // contiguous elements of the input along x and of the output along y: the fused fastest modes of
    // each side for as long as they are adjacent in memory
    int64_t in_run = 1, out_run = 1;
    // fused problem, for reporting: extents (input order) and permutation
    std::vector<int64_t> fused_extents;
    std::vector<int> fused_perm;
};

// out mode p = in mode perm[p]; both packed, last mode fastest.
// out_row_modes > 0: the last out_row_modes output modes form rows that lie out_row_pitch elements apart in the
// output (out_row_pitch >= the rows' packed length, a multiple of a 16-byte vector); the slower output modes count
// rows.  The elements between a row's end and the next row are not written.  `total` stays the number of elements
// moved.
std::string plan_permute(int rank, const int64_t *extents_in, const int *perm, int elem_size, PermutePlan &plan,
                         int out_row_modes = 0, int64_t out_row_pitch = 0);

}  // namespace ctc
