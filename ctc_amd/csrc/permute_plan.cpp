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

std::string plan_permute(int rank, const int64_t *extents_in, const int *perm, int elem_size, PermutePlan &pl,
                         int out_row_modes, int64_t out_row_pitch)
{
    pl = PermutePlan();
    if (rank <= 0 || rank > 32) return "bad rank";
    if (elem_size != 2 && elem_size != 4 && elem_size != 8) return "element size must be 2, 4 or 8 bytes";
    const int64_t V = 16 / elem_size;  // elements per 16-byte vector
    std::vector<int> seen(rank, 0);
    int64_t total = 1;
    for (int p = 0; p < rank; ++p) {
        if (perm[p] < 0 || perm[p] >= rank || seen[perm[p]]) return "perm is not a permutation";
        seen[perm[p]] = 1;
        if (extents_in[p] <= 0) return "extents must be positive";
        total *= extents_in[p];
    }
    pl.total = total;
    // padded output rows: the row part is the last out_row_modes output positions
    if (out_row_modes < 0 || out_row_modes > rank) return "bad out_row_modes";
    int64_t row_len = 1;
    for (int p = rank - out_row_modes; p < rank; ++p) row_len *= extents_in[perm[p]];
    const bool padded = out_row_modes > 0 && out_row_pitch != row_len;
    if (padded && (out_row_pitch < row_len || out_row_pitch % V != 0 || row_len < 2))
        return "out_row_pitch must be at least the row length (of 2 elements or more) and a multiple of a 16-byte vector";

    // 1. drop extent-1 modes
    std::vector<int> keep_id(rank, -1);
    std::vector<int64_t> e1;
    for (int i = 0; i < rank; ++i)
        if (extents_in[i] > 1) {
            keep_id[i] = (int)e1.size();
            e1.push_back(extents_in[i]);
        }
    std::vector<int> q1;
    std::vector<char> in_row1;  // per entry of q1: part of a padded output row
    for (int p = 0; p < rank; ++p)
        if (keep_id[perm[p]] >= 0) {
            q1.push_back(keep_id[perm[p]]);
            in_row1.push_back(padded && p >= rank - out_row_modes);
        }

    // 2. fuse runs of input modes that stay adjacent and in order in the output (never across the end of a
    // padded row: the modes that count rows are not contiguous with the modes inside a row)
    std::vector<std::pair<int, int>> groups;  // [first, last] input mode of each output group
    std::vector<char> group_in_row;
    for (size_t p = 0; p < q1.size();) {
        size_t e = p;
        while (e + 1 < q1.size() && q1[e + 1] == q1[e] + 1 && in_row1[e + 1] == in_row1[e]) ++e;
        groups.push_back({q1[p], q1[e]});
        group_in_row.push_back(in_row1[p]);
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
        // output group p is groups[p]; a group that counts rows steps by whole (padded) rows
        if (padded)
            for (int p = 0; p < r; ++p)
                if (!group_in_row[(size_t)p]) so[q[p]] = so[q[p]] / row_len * out_row_pitch;
    }

    if (r >= 1) {
        pl.fast_in_extent = ext[r - 1];
        pl.fast_out_extent = ext[q[r - 1]];
    }
    std::vector<int> xm, ym, om;
    if (r <= 1 || q[r - 1] == r - 1) {
        pl.same_fast = true;
        pl.vec_mode = (r >= 1 && ext[r - 1] % V == 0) ? 2 : 0;
        if (r >= 1) xm.push_back(r - 1);
        for (int p = r - 2; p >= 0; --p) om.push_back(q[p]);  // fastest first, output order
    } else {
        // Each side starts from its fastest mode.  A side whose fastest extent is a multiple of V may fuse
        // further modes (a vector never straddles two runs of the fastest mode): first up to a volume of
        // 64, then - below - for as long as that trims the part of the last patch that hangs over the
        // edge.  A side whose fastest extent is NOT a multiple of V keeps that one mode when it is long
        // enough (>= 32 elements of 4 or 8 bytes): the kernel then moves whole vectors at element-aligned
        // addresses and finishes each run with single elements.  Shorter odd extents fuse as before and
        // go through the element kernel.
        const int fi = r - 1, fo = q[r - 1];
        const auto side_mode = [&](int64_t e) { return e % V == 0 ? 2 : (elem_size >= 4 && e >= 32) ? 1 : 0; };
        const int mode_x = side_mode(ext[fi]), mode_y = side_mode(ext[fo]);
        pl.vec_mode = std::min(mode_x, mode_y);
        const bool fuse_x = pl.vec_mode == 0 || mode_x == 2, fuse_y = pl.vec_mode == 0 || mode_y == 2;
        const int64_t tile = pl.vec_mode == 0 ? 64 : (elem_size >= 4 ? 32 : 16) * V;  // the larger patch of the vector kernel
        std::vector<int> used(r, 0);
        int64_t vol_y = 1, vol_x = 1;
        for (int p = r - 1; p >= 0 && vol_y < 64; --p) {
            if (q[p] == fi) break;
            ym.push_back(q[p]);
            used[q[p]] = 1;
            vol_y *= ext[q[p]];
            if (!fuse_y) break;
        }
        for (int i = r - 1; i >= 0 && vol_x < 64; --i) {
            if (used[i]) break;
            xm.push_back(i);
            used[i] = 1;
            vol_x *= ext[i];
            if (!fuse_x) break;
        }
        // Patches are tile x tile: an index of 80 fills 62 % of its two patches.  Take in further unused
        // modes (next in the side's own order, so that runs stay as long as they can) while more than 3 %
        // of a side's patches hang over the edge; the tables stay small (at most 64 K entries a side).
        const auto waste = [&](int64_t n) { return (double)((n + tile - 1) / tile * tile) / (double)n; };
        constexpr int64_t kMaxSide = 65536;
        for (bool changed = true; changed;) {
            changed = false;
            if (fuse_x && waste(vol_x) > 1.03)
                for (int i = r - 1; i >= 0; --i)
                    if (!used[i] && vol_x * ext[i] <= kMaxSide) {
                        xm.push_back(i);
                        used[i] = 1;
                        vol_x *= ext[i];
                        changed = true;
                        break;
                    }
            if (fuse_y && waste(vol_y) > 1.03)
                for (int p = r - 1; p >= 0; --p)
                    if (!used[q[p]] && vol_y * ext[q[p]] <= kMaxSide) {
                        ym.push_back(q[p]);
                        used[q[p]] = 1;
                        vol_y *= ext[q[p]];
                        changed = true;
                        break;
                    }
        }
        // fastest first in output order: consecutive o values (one workgroup walks a few) extend the
        // contiguous runs of the stores; ordering them by the input instead measured 5 % slower
        for (int p = r - 1; p >= 0; --p)
            if (!used[q[p]]) om.push_back(q[p]);
    }
    const auto run = [&](const std::vector<int> &modes, const std::vector<int64_t> &stride) {
        int64_t n = 1;
        for (int m : modes) {
            if (stride[m] != n) break;
            n *= ext[m];
        }
        return n;
    };
    pl.in_run = run(xm, si);
    pl.out_run = run(pl.same_fast ? xm : ym, so);
    offsets(xm, ext, si, pl.in_x, pl.X);
    offsets(xm, ext, so, pl.out_x, pl.X);
    offsets(ym, ext, si, pl.in_y, pl.Y);
    offsets(ym, ext, so, pl.out_y, pl.Y);
    offsets(om, ext, si, pl.in_o, pl.O);
    offsets(om, ext, so, pl.out_o, pl.O);
    return "";
}

}  // namespace ctc
