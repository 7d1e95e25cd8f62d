// FASTR read-vs-consensus scoring on gfx950.
//
// For every (consensus, read) pair: argmin over offsets o of
//     sum_r [read[r] != ref[o+r] and both are ACGTacgt] * qual[r]  +  99 per base hanging past the consensus,
// ties resolved in the visiting order of the CPU loop of the reference
// (apps/fastr/IndelRealigner.cpp:502-542): original alignment, then 0..orig-1, then orig+1..refLen-readLen.
//
// Mapping: one workgroup = one consensus x up to 4 reads, one wave per read.  The consensus is
// staged once into LDS with irregular bases replaced by 0xFF, the read with irregular bases
// replaced by 0xFE and its qualities beside it, so the inner test is two byte compares.
// Phase 1: the wave scores the original alignment with its lanes across read positions; that
// score bounds every other offset.  Phase 2: lanes across offsets, 64 per pass in visiting
// order; a lane stops adding once it exceeds the bound, and the pass ends when every lane has
// (the reference's quitAboveThisValue short cut, which leaves the result unchanged because an
// offset whose sum exceeds the bound can never win).  The winner is the minimum of the 64-bit
// key (score << 32 | visiting rank) over lanes and passes.
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

__global__ void __launch_bounds__(64 * kWaves) fastr_kernel(const FastrParams p)
{
    __shared__ unsigned char s_ref[kMaxRefLds];
    __shared__ unsigned char s_read[kWaves][kMaxReadLds];
    __shared__ unsigned char s_qual[kWaves][kMaxReadLds];

    const FastrWork w = p.work[blockIdx.x];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const unsigned char *ref = p.ref + w.ref_off;
    const int ref_len = w.ref_len;

    for (int i = threadIdx.x; i < ref_len; i += blockDim.x) {
        const unsigned c = ref[i];
        s_ref[i] = regular_base(c) ? (unsigned char)c : (unsigned char)0xFF;
    }
    const bool active = wave < w.n_reads;
    const long long g = (long long)w.first_read + wave;
    int read_len = 0, orig = 0;
    if (active) {
        read_len = p.read_len[g];
        orig = p.original_alignment[g];
        const unsigned char *rs = p.read_seq + g * p.read_stride;
        const unsigned char *qs = p.quals + g * p.read_stride;
        for (int i = lane; i < read_len; i += 64) {
            const unsigned c = rs[i];
            s_read[wave][i] = regular_base(c) ? (unsigned char)c : (unsigned char)0xFE;
            s_qual[wave][i] = qs[i];
        }
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
            if (rc != 0xFFu && qc != 0xFEu && rc != qc) part += (unsigned)(signed char)ql[r];
        }
    }
    const unsigned bound = wave_sum_u32(part);
    unsigned long long best = ((unsigned long long)bound << 32);  // rank 0 = original alignment

    if (bound != 0) {
        // visiting rank j+1 for j = 0 .. n_other-1: offsets 0..orig-1, then orig+1..max_start
        const int max_start = ref_len - read_len;
        const int left = orig > 0 ? orig : 0;
        const int right = max_start > orig ? max_start - orig : 0;
        const int n_other = left + right;
        for (int base = 0; base < n_other; base += 64) {
            const int j = base + lane;
            const bool valid = j < n_other;
            const int off = j < left ? j : (orig + 1 + (j - left));
            unsigned sum = valid ? 0u : 0xFFFFFFFFu;
            for (int r0 = 0; r0 < read_len; r0 += 8) {
                if (sum <= bound) {
                    const int r1 = r0 + 8 < read_len ? r0 + 8 : read_len;
                    for (int r = r0; r < r1; ++r) {
                        const int x = off + r;
                        if (x >= ref_len) sum += kMaxQual;
                        else {
                            const unsigned rc = s_ref[x], qc = rd[r];
                            if (rc != 0xFFu && qc != 0xFEu && rc != qc) sum += (unsigned)(signed char)ql[r];
                        }
                    }
                }
                if (__all(sum > bound)) break;
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
    const int read_len = p.read_len[g], orig = p.original_alignment[g];
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
    for (int r = lane; r < read_len; r += 64) part += score_at(orig, r);
    const unsigned bound = wave_sum_u32(part);
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
