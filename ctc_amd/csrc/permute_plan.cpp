// Host-side planning of a mode permutation.  See permute_plan.hpp.
#include "permute_plan.hpp"

#include <algorithm>

namespace ctc {

namespace {

// offsets of a fused index over `modes` (fastest first) for the given per-mode strides
void offsets(const std::vector<int> &modes, const std::vector<int64_t> &ext,
             const std::vector<int64_t> &stride, std::vector<int64_t> &tab, int64_t &count)
{
    count = 1;
    for (int m : modes) count *= ext[m];
    tab.assign((size_t)count, 0);
    std::vector<int64_t> idx(modes.size(), 0);
    int64_t off = 0;
    for (int64_t i = 0; i < count; ++i) {
        tab[(size_t)i] = off;
        for (size_t d = 0; d < modes.size(); ++d) {
            int m = modes[d];
            off += stride[m];
            if (++idx[d] < ext[m]) break;
            off -= idx[d] * stride[m];
            idx[d] = 0;
        }
    }
}

}  // namespace

std::string plan_permute(int rank, const int64_t *extents_in, const int *perm, PermutePlan &pl)
{
    pl = PermutePlan();
    if (rank <= 0 || rank > 32) return "bad rank";
    std::vector<int> seen(rank, 0);
    int64_t total = 1;
    for (int p = 0; p < rank; ++p) {
        if (perm[p] < 0 || perm[p] >= rank || seen[perm[p]]) return "perm is not a permutation";
        seen[perm[p]] = 1;
        if (extents_in[p] <= 0) return "extents must be positive";
        total *= extents_in[p];
    }
    pl.total = total;

    // 1. drop extent-1 modes
    std::vector<int> keep_id(rank, -1);
    std::vector<int64_t> e1;
    for (int i = 0; i < rank; ++i)
        if (extents_in[i] > 1) {
            keep_id[i] = (int)e1.size();
            e1.push_back(extents_in[i]);
        }
    std::vector<int> q1;
    for (int p = 0; p < rank; ++p)
        if (keep_id[perm[p]] >= 0) q1.push_back(keep_id[perm[p]]);

    // 2. fuse runs of input modes that stay adjacent and in order in the output
    std::vector<std::pair<int, int>> groups;  // [first, last] input mode of each output group
    for (size_t p = 0; p < q1.size();) {
        size_t e = p;
        while (e + 1 < q1.size() && q1[e + 1] == q1[e] + 1) ++e;
        groups.push_back({q1[p], q1[e]});
        p = e + 1;
    }
    std::vector<int> order(groups.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    std::sort(order.begin(), order.end(), [&](int a, int b) { return groups[a].first < groups[b].first; });
    const int r = (int)groups.size();
    std::vector<int64_t> ext(r, 1);
    std::vector<int> in_pos_of_group(r, 0);
    for (int i = 0; i < r; ++i) {
        int g = order[i];
        in_pos_of_group[g] = i;
        for (int m = groups[g].first; m <= groups[g].second; ++m) ext[i] *= e1[m];
    }
    std::vector<int> q(r);
    for (int p = 0; p < r; ++p) q[p] = in_pos_of_group[p];
    pl.fused_extents = ext;
    pl.fused_perm = q;

    std::vector<int64_t> si(r, 1), so(r, 1);
    {
        int64_t s = 1;
        for (int i = r - 1; i >= 0; --i) { si[i] = s; s *= ext[i]; }
        s = 1;
        for (int p = r - 1; p >= 0; --p) { so[q[p]] = s; s *= ext[q[p]]; }
    }

    if (r >= 1) {
        pl.fast_in_extent = ext[r - 1];
        pl.fast_out_extent = ext[q[r - 1]];
    }
    std::vector<int> xm, ym, om;
    if (r <= 1 || q[r - 1] == r - 1) {
        pl.same_fast = true;
        if (r >= 1) xm.push_back(r - 1);
        for (int p = r - 2; p >= 0; --p) om.push_back(q[p]);  // fastest first, output order
    } else {
        const int fi = r - 1;
        std::vector<int> used(r, 0);
        int64_t vol = 1;
        for (int p = r - 1; p >= 0 && vol < 64; --p) {
            if (q[p] == fi) break;
            ym.push_back(q[p]);
            used[q[p]] = 1;
            vol *= ext[q[p]];
        }
        vol = 1;
        for (int i = r - 1; i >= 0 && vol < 64; --i) {
            if (used[i]) break;
            xm.push_back(i);
            used[i] = 1;
            vol *= ext[i];
        }
        // fastest first in output order: consecutive o values (one workgroup walks a few) extend the
        // contiguous runs of the stores; ordering them by the input instead measured 5 % slower
        for (int p = r - 1; p >= 0; --p)
            if (!used[q[p]]) om.push_back(q[p]);
    }
    offsets(xm, ext, si, pl.in_x, pl.X);
    offsets(xm, ext, so, pl.out_x, pl.X);
    offsets(ym, ext, si, pl.in_y, pl.Y);
    offsets(ym, ext, so, pl.out_y, pl.Y);
    offsets(om, ext, si, pl.in_o, pl.O);
    offsets(om, ext, so, pl.out_o, pl.O);
    return "";
}

}  // namespace ctc
