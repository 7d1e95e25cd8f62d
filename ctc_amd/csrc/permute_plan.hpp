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
    // extents of the (fused) fastest input and output modes: 16-byte vectors of V elements are
    // legal on both sides when V divides both (every other stride is then a multiple of V too)
    int64_t fast_in_extent = 1, fast_out_extent = 1;
    // fused problem, for reporting: extents (input order) and permutation
    std::vector<int64_t> fused_extents;
    std::vector<int> fused_perm;
};

// out mode p = in mode perm[p]; both packed, last mode fastest.
std::string plan_permute(int rank, const int64_t *extents_in, const int *perm, PermutePlan &plan);

}  // namespace ctc
