// FASTR read-vs-consensus scoring on gfx950.
//
// For every (consensus, read) pair: argmin over offsets o of
//     sum_r [read[r] != ref[o+r] and both are ACGTacgt] * qual[r]  +  99 per base hanging past the consensus,
// ties resolved in the visiting order of the CPU loop of the reference
// (apps/fastr/IndelRealigner.cpp:502-542): original alignment, then 0..orig-1, then orig+1..refLen-readLen.
//
// Mapping: one workgroup = one consensus x up to 4 reads, one wave per read.  The consensus is
// staged once into LDS with irregular bases replaced by 0x80, the read beside its qualities
// with the quality of an irregular read base set to 0, so a position contributes its quality
// exactly when the consensus byte is not 0x80 and differs from the read byte.
// Phase 1: the wave scores the original alignment with its lanes across read positions; that
// score bounds every other offset.  Phase 2: lanes across offsets, 64 per pass in visiting
// order; a lane stops adding once it exceeds the bound, and the pass ends when every lane has
// (the reference's quitAboveThisValue short cut, which leaves the result unchanged because an
// offset whose sum exceeds the bound can never win).  The winner is the minimum of the 64-bit
// key (score << 32 | visiting rank) over lanes and passes.
//
// Phase 2 works on eight positions at a time: a lane keeps a sliding window of consensus
// dwords (two new ds_read_b32 per step, shifted to its byte offset with v_alignbyte_b32), the
// read and quality bytes of the step are the same for all lanes (one broadcast ds_read_b64
// each), and four positions are scored with packed byte arithmetic: x = consensus ^ read has
// bit 7 set where the consensus byte is the 0x80 sentinel (bases are ASCII letters) and is
// non-zero where the bases differ; the per-byte flag "low seven bits non-zero and bit 7
// clear" is widened to a byte mask, and v_sad_u8 adds the masked qualities.  Passes that hang
// past the end of the consensus (possible only left of an original alignment that itself
// hangs) go position by position; a read with a quality byte above 127 (negative in the
// reference's signed arithmetic) is replayed by one lane exactly as the reference loops.
#include "kernels.hpp"

namespace ctc {

namespace {

constexpr int kWaves = 4;
constexpr int kMaxRefLds = 8192;   // consensus bytes staged per workgroup
constexpr int kMaxReadLds = 1024;  // read bytes staged per wave
constexpr unsigned kMaxQual = 99;  // apps/fastr/IndelRealigner.cpp:170

__device__ __forceinline__ bool regular_base(unsigned c)
{
    const unsigned u = c & 0xDFu;  // fold case
    return (c < 128u) && (u == 'A' || u == 'C' || u == 'G' || u == 'T');
}

__device__ __forceinline__ unsigned long long wave_min_u64(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)v, off), hi = __shfl_xor((unsigned)(v >> 32), off);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        v = o < v ? o : v;
    }
    return v;
}

__device__ __forceinline__ unsigned wave_sum_u32(unsigned v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// The reference loop as it is written (apps/fastr/IndelRealigner.cpp:264-295, 502-542), for one
// pair on one lane.  Needed when a quality byte is above 127, i.e. negative in the reference's
// signed arithmetic: partial sums are then not monotonic, and where the reference abandons an
// offset depends on the position and on the best score found so far.  ref_at(x) returns the
// consensus byte with irregular bases as 0x80; rd / ql are the read's original bytes.
template <typename RefAt>
__device__ void replay_reference(RefAt ref_at, int ref_len, const unsigned char *rd, const unsigned char *ql, int read_len,
                                 int orig, int first_score, int *best_score, int *best_index)
{
    auto quality_sum = [&](int off, long long quit_above) {
        int sum = 0;
        for (int r = 0; r < read_len; ++r) {
            const int x = off + r;
            if (x >= ref_len) {
                sum += (int)kMaxQual;
                if (sum > quit_above) return sum;
            } else {
                const unsigned rc = ref_at(x), qc = rd[r];
                if (rc != 0x80u && regular_base(qc) && rc != qc) {
                    sum += (int)(signed char)ql[r];
                    if (sum > quit_above) return sum;
                }
            }
        }
        return sum;
    };
    long long bst = first_score;
    int index = orig;
    if (bst != 0) {
        for (int i = 0; i < orig && bst != 0; ++i) {
            const int v = quality_sum(i, bst);
            if (v < bst) { bst = v; index = i; }
        }
        for (int i = orig + 1; i <= ref_len - read_len && bst != 0; ++i) {
            const int v = quality_sum(i, bst);
            if (v < bst) { bst = v; index = i; }
        }
    }
    *best_score = (int)bst;
    *best_index = index;
}

__global__ void __launch_bounds__(64 * kWaves) fastr_kernel(const FastrParams p)
{
    // padded so that the eight-position steps may run past the end of the consensus or the read
    __shared__ __attribute__((aligned(8))) unsigned char s_ref[kMaxRefLds + 32];
    __shared__ __attribute__((aligned(8))) unsigned char s_read[kWaves][kMaxReadLds + 8];
    __shared__ __attribute__((aligned(8))) unsigned char s_qual[kWaves][kMaxReadLds + 8];

    const FastrWork w = p.work[blockIdx.x];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const unsigned char *ref = p.ref + w.ref_off;
    const int ref_len = w.ref_len;

    for (int i = threadIdx.x; i < ref_len + 32; i += blockDim.x) {
        const unsigned c = i < ref_len ? ref[i] : 0x80u;
        s_ref[i] = regular_base(c) ? (unsigned char)c : (unsigned char)0x80;
    }
    const bool active = wave < w.n_reads;
    const long long g = (long long)w.first_read + wave;
    int read_len = 0, orig = 0;
    bool signed_quals = false;
    if (active) {
        // device-side arguments cannot be checked by the host: keep them inside the read's slot
        // and the consensus so that bad input gives a wrong score, not a stray access
        read_len = min(max(p.read_len[g], 0), p.read_stride);
        orig = max(p.original_alignment[g], 0);
        const unsigned char *rs = p.read_seq + g * p.read_stride;
        const unsigned char *qs = p.quals + g * p.read_stride;
        for (int i = lane; i < read_len + 8; i += 64) {
            const bool in = i < read_len;
            const unsigned c = in ? rs[i] : 0u;
            const unsigned q = (in && regular_base(c)) ? qs[i] : 0u;
            s_read[wave][i] = (unsigned char)c;
            s_qual[wave][i] = (unsigned char)q;
            signed_quals |= q > 127u;
        }
        signed_quals = __any(signed_quals);
    }
    __syncthreads();
    if (!active) return;
    const unsigned char *rd = s_read[wave];
    const unsigned char *ql = s_qual[wave];

    // phase 1: the original alignment, lanes across read positions
    unsigned part = 0;
    for (int r = lane; r < read_len; r += 64) {
        const int x = orig + r;
        if (x >= ref_len) part += kMaxQual;
        else if (x >= 0) {
            const unsigned rc = s_ref[x], qc = rd[r];
            if (rc != 0x80u && rc != qc) part += (unsigned)(signed char)ql[r];
        }
    }
    const unsigned bound = wave_sum_u32(part);
    unsigned long long best = ((unsigned long long)bound << 32);  // rank 0 = original alignment

    if (signed_quals) {
        // s_qual holds the original quality wherever the read base is regular, which is all
        // the replay looks at
        if (lane == 0)
            replay_reference([&](int x) { return (unsigned)s_ref[x]; }, ref_len, rd, ql, read_len, orig, (int)bound,
                             &p.best_score[w.score_off + wave], &p.best_index[w.score_off + wave]);
        return;
    }

    if (bound != 0) {
        // visiting rank j+1 for j = 0 .. n_other-1: offsets 0..orig-1, then orig+1..max_start
        const int max_start = ref_len - read_len;
        const int left = orig > 0 ? orig : 0;
        const int right = max_start > orig ? max_start - orig : 0;
        const int n_other = left + right;
        const unsigned *ref32 = reinterpret_cast<const unsigned *>(s_ref);
        for (int base = 0; base < n_other; base += 64) {
            // lanes past the last offset score that last offset again and never win
            const int j_last = base + 63 < n_other ? base + 63 : n_other - 1;
            const int j = base + lane < n_other ? base + lane : j_last;
            const bool valid = base + lane < n_other;
            const int off = j < left ? j : (orig + 1 + (j - left));
            const int off_last = j_last < left ? j_last : (orig + 1 + (j_last - left));
            unsigned sum = valid ? 0u : 0xFFFFFFFFu;
            if (off_last + read_len <= ref_len) {
                // packed path: no position of this pass hangs past the consensus
                unsigned acc = 0;
                const int shift = off & 3;
                int idx = off >> 2;
                unsigned d0 = ref32[idx], d1 = ref32[idx + 1];
                for (int r0 = 0; r0 < read_len; r0 += 8) {
                    const unsigned d2 = ref32[idx + 2], d3 = ref32[idx + 3];
                    idx += 2;
                    const uint2 rw = *reinterpret_cast<const uint2 *>(rd + r0);
                    const uint2 qw = *reinterpret_cast<const uint2 *>(ql + r0);
                    const unsigned c_lo = __builtin_amdgcn_alignbyte(d1, d0, shift);
                    const unsigned c_hi = __builtin_amdgcn_alignbyte(d2, d1, shift);
                    d0 = d2;
                    d1 = d3;
                    auto weighted = [](unsigned cons, unsigned read, unsigned qual, unsigned add) {
                        const unsigned x = cons ^ read;
                        // bit 7 of a byte: low seven bits of x non-zero, and bit 7 of x clear
                        const unsigned t = (((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) & ~x) & 0x80808080u;
                        const unsigned mask = (t >> 7) * 0xFFu;
                        return __builtin_amdgcn_sad_u8(qual & mask, 0u, add);
                    };
                    acc = weighted(c_lo, rw.x, qw.x, acc);
                    acc = weighted(c_hi, rw.y, qw.y, acc);
                    if (__all(acc > bound)) break;
                }
                if (valid) sum = acc;
            } else {
                for (int r0 = 0; r0 < read_len; r0 += 8) {
                    if (sum <= bound) {
                        const int r1 = r0 + 8 < read_len ? r0 + 8 : read_len;
                        for (int r = r0; r < r1; ++r) {
                            const int x = off + r;
                            if (x >= ref_len) sum += kMaxQual;
                            else {
                                const unsigned rc = s_ref[x], qc = rd[r];
                                if (rc != 0x80u && rc != qc) sum += (unsigned)(signed char)ql[r];
                            }
                        }
                    }
                    if (__all(sum > bound)) break;
                }
            }
            if (valid && sum <= bound) {
                const unsigned long long key = ((unsigned long long)sum << 32) | (unsigned)(j + 1);
                best = key < best ? key : best;
            }
        }
        best = wave_min_u64(best);
    }
    if (lane == 0) {
        const unsigned rank = (unsigned)best;
        const int left = orig > 0 ? orig : 0;
        int index = orig;
        if (rank != 0) {
            const int j = (int)rank - 1;
            index = j < left ? j : (orig + 1 + (j - left));
        }
        p.best_score[w.score_off + wave] = (int)(best >> 32);
        p.best_index[w.score_off + wave] = index;
    }
}

// Same result without LDS staging limits: consensus or read too long for the staged kernel.
__global__ void __launch_bounds__(64 * kWaves) fastr_kernel_global(const FastrParams p)
{
    const FastrWork w = p.work[blockIdx.x];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    if (wave >= w.n_reads) return;
    const unsigned char *ref = p.ref + w.ref_off;
    const int ref_len = w.ref_len;
    const long long g = (long long)w.first_read + wave;
    const int read_len = min(max(p.read_len[g], 0), p.read_stride), orig = max(p.original_alignment[g], 0);
    const unsigned char *rd = p.read_seq + g * p.read_stride;
    const unsigned char *ql = p.quals + g * p.read_stride;

    auto score_at = [&](int off, int r) -> unsigned {
        const int x = off + r;
        if (x >= ref_len) return kMaxQual;
        if (x < 0) return 0u;
        const unsigned rc = ref[x], qc = rd[r];
        return (regular_base(rc) && regular_base(qc) && rc != qc) ? (unsigned)(signed char)ql[r] : 0u;
    };
    unsigned part = 0;
    bool signed_quals = false;
    for (int r = lane; r < read_len; r += 64) {
        part += score_at(orig, r);
        signed_quals |= ql[r] > 127u && regular_base(rd[r]);
    }
    const unsigned bound = wave_sum_u32(part);
    if (__any(signed_quals)) {
        if (lane == 0)
            replay_reference([&](int x) { const unsigned c = ref[x]; return regular_base(c) ? c : 0x80u; }, ref_len, rd, ql,
                             read_len, orig, (int)bound, &p.best_score[w.score_off + wave],
                             &p.best_index[w.score_off + wave]);
        return;
    }
    unsigned long long best = ((unsigned long long)bound << 32);
    const int max_start = ref_len - read_len;
    const int left = orig > 0 ? orig : 0;
    const int right = max_start > orig ? max_start - orig : 0;
    const int n_other = left + right;
    if (bound != 0) {
        for (int base = 0; base < n_other; base += 64) {
            const int j = base + lane;
            const bool valid = j < n_other;
            const int off = j < left ? j : (orig + 1 + (j - left));
            unsigned sum = valid ? 0u : 0xFFFFFFFFu;
            for (int r = 0; r < read_len; ++r) {
                if (sum <= bound) sum += score_at(off, r);
                if ((r & 7) == 7 && __all(sum > bound)) break;
            }
            if (valid && sum <= bound) {
                const unsigned long long key = ((unsigned long long)sum << 32) | (unsigned)(j + 1);
                best = key < best ? key : best;
            }
        }
        best = wave_min_u64(best);
    }
    if (lane == 0) {
        const unsigned rank = (unsigned)best;
        int index = orig;
        if (rank != 0) {
            const int j = (int)rank - 1;
            index = j < left ? j : (orig + 1 + (j - left));
        }
        p.best_score[w.score_off + wave] = (int)(best >> 32);
        p.best_index[w.score_off + wave] = index;
    }
}

}  // namespace

int fastr_max_ref_lds() { return kMaxRefLds; }
int fastr_max_read_lds() { return kMaxReadLds; }
int fastr_reads_per_block() { return kWaves; }

hipError_t launch_fastr(const FastrParams &p, int n_blocks, bool staged, hipStream_t stream)
{
    if (n_blocks <= 0) return hipSuccess;
    if (staged)
        hipLaunchKernelGGL(fastr_kernel, dim3((unsigned)n_blocks), dim3(64 * kWaves), 0, stream, p);
    else
        hipLaunchKernelGGL(fastr_kernel_global, dim3((unsigned)n_blocks), dim3(64 * kWaves), 0, stream, p);
    return hipGetLastError();
}

}  // namespace ctc
