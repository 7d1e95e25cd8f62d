// Batched pairwise alignment of reads against reference windows on gfx950: the aligner FASTR
// uses to propose alternate consensuses (apps/fastr/SWPairwiseAlignment.cpp:128-399, called
// per read at apps/fastr/IndelRealigner.cpp:366-371).
//
// Score recurrence (calculateMatrix): cell (i, j), i over the reference window, j over the read,
//   diag  = S[i-1][j-1] + (a[i-1] == b[j-1] ? match : mismatch)
//   down  = best over k of S[i-k][j] + open + (k-1) extend   (kept per column as a running best)
//   right = best over k of S[i][j-k] + open + (k-1) extend   (kept per row)
//   S[i][j] = down > right ? (down > diag ? down : diag) : (right > diag ? right : diag)
// with row 0 and column 0 zero and no clamping at zero; the back-track entry is the gap length
// (+ vertical, - horizontal, 0 diagonal).  Arithmetic is double, as in the reference.
//
// Mapping: one wave per pair.  The window is cut into blocks of 64 rows; in a block lane l owns
// row 64*rb + l + 1 and walks the columns skewed by its lane number, so that at every step the
// wave covers one anti-diagonal; a lane that finishes its row goes straight on to its row of the
// next block, so the blocks overlap and the skew is paid once per pair.  Everything a cell needs from the row above (score, the column's
// running vertical gap and its length, the read base of the column) is handed from lane l to
// lane l+1 by a whole-wave DPP shift per step; lane 0 takes it from a per-column boundary
// array in LDS that lane 63 of the previous block left there.  Back-track entries go to global
// memory as int16, one 128-byte line per step.  The start cell follows the reference's two
// scans (last column top to bottom with >=, then last row with the closer-to-diagonal rule),
// and the back-track walks the stored entries exactly as calculateCigar does.
//
// Score type: double as in the reference, or int32 when the four weights are integers of
// modest size.  Every score is then an integer that both types represent exactly, every
// comparison has the same outcome, and the int32 cell costs about half the instructions
// (one register per value for the selects and the DPP hand-over).  The "no gap yet" and
// "minus infinity" sentinels only need to stay below every reachable score.
#include "kernels.hpp"

#include <climits>
#include <cstdlib>

namespace ctc {

namespace {

constexpr int kSwWaves = 4;

__device__ __forceinline__ int shr1_i(int old, int x)
{
    return __builtin_amdgcn_update_dpp(old, x, 0x138 /* wave_shr:1 */, 0xF, 0xF, false);
}
__device__ __forceinline__ double shr1_d(double x)
{
    const int hi = __double2hiint(x), lo = __double2loint(x);
    return __hiloint2double(shr1_i(hi, hi), shr1_i(lo, lo));
}
__device__ __forceinline__ int shr1_s(int x) { return shr1_i(x, x); }
__device__ __forceinline__ double shr1_s(double x) { return shr1_d(x); }
__device__ __forceinline__ int xor_s(int x, int off) { return __shfl_xor(x, off); }
__device__ __forceinline__ double xor_s(double x, int off)
{
    return __hiloint2double(__shfl_xor(__double2hiint(x), off), __shfl_xor(__double2loint(x), off));
}

template <typename S> struct SwLimits;
template <> struct SwLimits<double> {
    static __device__ __forceinline__ double cutoff() { return -1e100; }   // MATRIX_MIN_CUTOFF of the reference
    static __device__ __forceinline__ double no_gap() { return -1.0e40; }
    static __device__ __forceinline__ double lowest() { return -1.0e300; }
};
template <> struct SwLimits<int> {
    // weights are at most kSwMaxIntWeight in magnitude and a path has at most 65534 steps, so
    // scores stay within +-6.6e7; no_gap may drift down by one extension per row or column
    static __device__ __forceinline__ int cutoff() { return INT_MIN; }
    static __device__ __forceinline__ int no_gap() { return -(1 << 29); }
    static __device__ __forceinline__ int lowest() { return INT_MIN; }
};

template <typename S> __global__ void __launch_bounds__(64 * kSwWaves) sw_kernel(const SwParams p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sw_lds[];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int pair_local = blockIdx.x * kSwWaves + wave;
    if (pair_local >= p.n_pairs) return;
    const long long pair = (long long)p.first_pair + pair_local;

    // per-wave LDS: boundary score / vertical gap / gap length per column, last-row scores,
    // read bases, collected CIGAR operations
    const int L1 = p.max_read_len + 1;
    const size_t per_wave = (size_t)L1 * (3 * sizeof(S) + sizeof(int)) + (size_t)((p.max_read_len + 7) & ~7) +
                            (size_t)p.max_ops * (sizeof(int) + 1) + 16;
    unsigned char *base = sw_lds + (size_t)wave * ((per_wave + 15) & ~(size_t)15);
    S *s_up = reinterpret_cast<S *>(base);
    S *s_bgv = s_up + L1;
    S *s_last = s_bgv + L1;
    int *s_gsv = reinterpret_cast<int *>(s_last + L1);
    int *s_oplen = s_gsv + L1;
    unsigned char *s_read = reinterpret_cast<unsigned char *>(s_oplen + p.max_ops);
    unsigned char *s_opty = s_read + ((p.max_read_len + 7) & ~7);

    const int n = p.ref_len[pair], m = p.read_len[pair];
    int32_t *out_n_ops = p.n_ops + pair;
    if (n < 0 || m < 0 || n > p.max_ref_len || m > p.max_read_len) {
        if (lane == 0) { *out_n_ops = -1; p.offset[pair] = 0; }
        return;
    }
    const unsigned char *a = p.ref + p.ref_off[pair];
    const unsigned char *b = p.reads + pair * (long long)p.read_stride;
    const S wm = (S)p.w_match, wx = (S)p.w_mismatch, wo = (S)p.w_open, we = (S)p.w_extend;
    const S kCutoff = SwLimits<S>::cutoff(), kNoGap = SwLimits<S>::no_gap();

    for (int j = lane; j <= m; j += 64) {
        s_up[j] = (S)0;
        s_bgv[j] = kNoGap;
        s_gsv[j] = 0;
        s_last[j] = (S)0;
        if (j < m) s_read[j] = b[j];
    }

    const int T = p.max_read_len + 63;  // steps per row block in the back-track store
    short *bt = p.bt + (size_t)pair_local * p.bt_per_pair;

    // One continuous sweep over all row blocks.  Lane l enters row block rb at step rb * P + l + 1
    // and spends P = max(m, 64) steps in it (the first m on columns 1..m), so a lane moves from
    // the end of its row in one block straight into its row of the next block: the skew of 63
    // steps is paid once per pair, not once per block.  P >= 64 makes sure lane 63 of block
    // rb - 1 has left column j in the LDS boundary arrays before lane 0 of block rb reads it.
    const int nrb = (n + 63) / 64;
    const int P = m > 64 ? m : 64;
    const int total = nrb * P + 63;
    int j = 1 - lane;  // this lane's column in its current row; <= 0: not started yet
    int rb = 0;
    S val = (S)0, up_prev = (S)0, bgh = kNoGap;
    int gsh = 0;
    S out_val = (S)0, out_bgv = kNoGap;
    int out_gsv = 0, out_base = 0;
    S best_s = (S)0;  // best of the last column over this lane's rows, later rows win ties
    int best_i = 0;
    int i = lane + 1;
    bool row_ok = i <= n;
    unsigned a_base = row_ok ? a[i - 1] : 0x100u;
    for (int step = 1; step <= total; ++step) {
        S r_up = shr1_s(out_val), r_bgv = shr1_s(out_bgv);
        int r_gsv = shr1_i(out_gsv, out_gsv), r_base = shr1_i(out_base, out_base);
        const bool in_row = j >= 1 && j <= m && rb < nrb;
        if (lane == 0 && in_row) {
            r_up = s_up[j];
            r_bgv = s_bgv[j];
            r_gsv = s_gsv[j];
            r_base = s_read[j - 1];
        }
        if (in_row) {
            if (j == 1) {  // row start: column 0 is all zeros and carries no horizontal gap
                val = (S)0;
                up_prev = (S)0;
                bgh = kNoGap;
                gsh = 0;
            }
            const S step_diag = up_prev + ((unsigned)r_base == a_base ? wm : wx);
            S prev_gap = r_up + wo;
            S bgv = r_bgv + we;
            int gsv;
            if (prev_gap > bgv) { bgv = prev_gap; gsv = 1; }
            else gsv = r_gsv + 1;
            prev_gap = val + wo;
            bgh += we;
            if (prev_gap > bgh) { bgh = prev_gap; gsh = 1; }
            else gsh += 1;
            S v;
            int tr;
            if (bgv > bgh) {
                if (bgv > step_diag) { v = bgv; tr = gsv; }
                else { v = step_diag; tr = 0; }
            } else {
                if (bgh > step_diag) { v = bgh; tr = -gsh; }
                else { v = step_diag; tr = 0; }
            }
            val = v > kCutoff ? v : kCutoff;
            up_prev = r_up;
            out_val = val;
            out_bgv = bgv;
            out_gsv = gsv;
            out_base = r_base;
            if (row_ok) {
                bt[((size_t)rb * T + (size_t)(j + lane - 1)) * 64 + lane] = (short)tr;
                // last column: max score, the largest row among equals (rows come in ascending order)
                if (j == m && val >= best_s) { best_s = val; best_i = i; }
                if (i == n) s_last[j] = val;
            }
            if (lane == 63) {
                s_up[j] = val;
                s_bgv[j] = bgv;
                s_gsv[j] = gsv;
            }
        }
        j += 1;
        if (j > P) {  // on to this lane's row of the next block
            j = 1;
            rb += 1;
            i += 64;
            row_ok = i <= n;
            a_base = row_ok ? a[i - 1] : 0x100u;
        }
    }
    // first scan of the start cell: best of the last column over all rows, larger rows win ties,
    // and only scores >= 0 (the scan starts from score 0 at row 0)
    S col_best = best_s;
    int col_best_i = best_i;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const S os = xor_s(col_best, off);
        const int oi = __shfl_xor(col_best_i, off);
        if (os > col_best || (os == col_best && oi > col_best_i)) { col_best = os; col_best_i = oi; }
    }

    // ---- start cell (calculateCigar, first part); every lane runs the same scalar walk ----
    int p1 = col_best_i, p2 = col_best_i > 0 ? m : 0;
    S maxscore = col_best;
    int segment_length = 0;
    for (int j = 1; j <= m; ++j) {
        const S s = s_last[j];
        const int d1 = n - j, d2 = p1 - p2;
        if ((s > maxscore || s == maxscore) && (d1 < 0 ? -d1 : d1) < (d2 < 0 ? -d2 : d2)) {
            p1 = n;
            p2 = j;
            maxscore = s;
            segment_length = m - j;
        }
    }

    // ---- back-track (calculateCigar, second part) ----
    int cnt = 0;
    auto emit = [&](char ty, int len) {
        if (lane == 0 && cnt < p.max_ops) { s_opty[cnt] = (unsigned char)ty; s_oplen[cnt] = len; }
        cnt += 1;
    };
    if (segment_length > 0) { emit('S', segment_length); segment_length = 0; }
    int state = 0;  // 0 M, 1 I, 2 D
    do {
        int btr = 0;
        if (p1 > 0 && p2 > 0) {
            const int rb = (p1 - 1) >> 6, l = (p1 - 1) & 63;
            btr = bt[((size_t)rb * T + (size_t)(p2 + l - 1)) * 64 + l];
        }
        int new_state, step = 1;
        if (btr > 0) { new_state = 2; step = btr; }
        else if (btr < 0) { new_state = 1; step = -btr; }
        else new_state = 0;
        if (new_state == 0) { p1 -= 1; p2 -= 1; }
        else if (new_state == 1) p2 -= step;
        else p1 -= step;
        if (new_state == state) segment_length += step;
        else {
            emit(state == 0 ? 'M' : (state == 1 ? 'I' : 'D'), segment_length);
            segment_length = step;
            state = new_state;
        }
    } while (p1 > 0 && p2 > 0);
    emit(state == 0 ? 'M' : (state == 1 ? 'I' : 'D'), segment_length);
    if (p2 > 0) emit('S', p2);

    // operations were collected back to front
    const int kept = cnt < p.max_ops ? cnt : p.max_ops;
    unsigned char *o_ty = p.op_type + pair * (long long)p.max_ops;
    int32_t *o_len = p.op_len + pair * (long long)p.max_ops;
    if (cnt <= p.max_ops) {
        for (int k = lane; k < kept; k += 64) {
            o_ty[k] = s_opty[cnt - 1 - k];
            o_len[k] = s_oplen[cnt - 1 - k];
        }
    }
    if (lane == 0) {
        *out_n_ops = cnt;  // > max_ops: the operations did not fit and none were written
        p.offset[pair] = p1;
    }
}

}  // namespace

static size_t sw_lds_bytes_for(int max_read_len, int max_ops, size_t score_bytes)
{
    const size_t L1 = (size_t)max_read_len + 1;
    const size_t per_wave = L1 * (3 * score_bytes + sizeof(int)) + (size_t)((max_read_len + 7) & ~7) +
                            (size_t)max_ops * (sizeof(int) + 1) + 16;
    return (size_t)kSwWaves * ((per_wave + 15) & ~(size_t)15);
}

size_t sw_lds_bytes(int max_read_len, int max_ops) { return sw_lds_bytes_for(max_read_len, max_ops, sizeof(double)); }

// int32 scores are exact when every weight is an integer of modest size
constexpr double kSwMaxIntWeight = 1000.0;
static bool sw_integer_weights(const SwParams &p)
{
    const double w[4] = {p.w_match, p.w_mismatch, p.w_open, p.w_extend};
    for (double x : w)
        if (!(x == (double)(long long)x) || x > kSwMaxIntWeight || x < -kSwMaxIntWeight) return false;
    return true;
}

size_t sw_bt_elems_per_pair(int max_ref_len, int max_read_len)
{
    return (size_t)((max_ref_len + 63) / 64) * (size_t)(max_read_len + 63) * 64;
}

hipError_t launch_sw(const SwParams &p, hipStream_t stream)
{
    if (p.n_pairs <= 0) return hipSuccess;
    static const bool force_double = getenv("CTC_SW_FORCE_DOUBLE") != nullptr;  // A/B runs
    const bool ints = !force_double && sw_integer_weights(p);
    const size_t lds = sw_lds_bytes_for(p.max_read_len, p.max_ops, ints ? sizeof(int) : sizeof(double));
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    const unsigned blocks = (unsigned)((p.n_pairs + kSwWaves - 1) / kSwWaves);
    if (ints)
        hipLaunchKernelGGL(sw_kernel<int>, dim3(blocks), dim3(64 * kSwWaves), lds, stream, p);
    else
        hipLaunchKernelGGL(sw_kernel<double>, dim3(blocks), dim3(64 * kSwWaves), lds, stream, p);
    return hipGetLastError();
}

}  // namespace ctc
