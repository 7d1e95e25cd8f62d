// Grouped small contractions on gfx950: many independent C_p = alpha * A_p * B_p + beta * C_p
// (ab = ak,kb) with extents of a few tens, all in ONE launch.  Each problem is far too
// small to fill a workgroup, so one wave owns one problem at a time and walks a
// grid-stride list of problems.  Operands are not staged through LDS: for a 16x16x4 MFMA
// the B fragment of lane l is B[4kk + (l>>4)][16j + (l&15)], i.e. 16 neighbouring lanes
// read 16 consecutive elements of a row, and a problem's whole footprint (<= 3 * 32 * 32
// elements) lives in L1/L2 after first touch.  The wave covers the output in 32 x 32
// blocks of 2 x 2 MFMA accumulators; extents need not be multiples of anything.
#include "kernels.hpp"

namespace ctc {

namespace {

template <typename T> struct Mf;
template <> struct Mf<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int acc_row(int lane, int j) { return (lane >> 4) + 4 * j; }
};
template <> struct Mf<float> {
    typedef float acc_t __attribute__((ext_vector_type(4)));
    static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
    static __device__ __forceinline__ int acc_row(int lane, int j) { return (lane >> 4) * 4 + j; }
};

// AT: A is stored [k][m] (A(m, k) at k lda + m) instead of [m][k]; BT: B is stored [n][k] (B(k, n) at n ldb + k)
// instead of [k][n].  The public entry point takes the plain forms; batched plans of small products take all four.
template <typename T, bool AT, bool BT>
__global__ void __launch_bounds__(256) grouped_kernel(const SmallGemm *__restrict__ problems, int count, const T *A,
                                                      const T *B, T *C, T alpha, T beta)
{
    typedef typename Mf<T>::acc_t acc_t;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int fr = lane & 15, fk = lane >> 4;

    for (int pi = wave; pi < count; pi += nwaves) {
        const SmallGemm g = problems[pi];
        const T *a = A + g.off_a;
        const T *b = B + g.off_b;
        T *c = C + g.off_c;
        for (int m0 = 0; m0 < g.m; m0 += 32) {
            for (int n0 = 0; n0 < g.n; n0 += 32) {
                acc_t acc[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = acc_t{0, 0, 0, 0};
                const int r0 = m0 + fr, r1 = m0 + 16 + fr;
                const int c0 = n0 + fr, c1 = n0 + 16 + fr;
                const bool r0ok = r0 < g.m, r1ok = r1 < g.m, c0ok = c0 < g.n, c1ok = c1 < g.n;
                // strides of a row / of k in A, of k / of a column in B
                const int64_t a_row = AT ? 1 : g.lda, a_k = AT ? g.lda : 1, b_k = BT ? 1 : g.ldb, b_col = BT ? g.ldb : 1;
                const T *a0 = a + (int64_t)(r0ok ? r0 : 0) * a_row;
                const T *a1 = a + (int64_t)(r1ok ? r1 : 0) * a_row;
                const T *b0 = b + (int64_t)(c0ok ? c0 : 0) * b_col;
                const T *b1 = b + (int64_t)(c1ok ? c1 : 0) * b_col;
                // a block with at most 16 rows (columns) inside the problem needs one row (column) of MFMAs only:
                // uniform per problem, so a batch of 8 x 8 or 16 x 16 products issues a quarter of the MFMAs and half
                // of the loads
                const bool two_m = g.m - m0 > 16, two_n = g.n - n0 > 16;
                for (int k0 = 0; k0 < g.k; k0 += 4) {
                    const int k = k0 + fk;
                    const bool kok = k < g.k;
                    const int kc = kok ? k : 0;
                    T fa0 = a0[kc * a_k], fb0 = b0[kc * b_k], fa1 = 0, fb1 = 0;
                    if (two_m) fa1 = a1[kc * a_k];
                    if (two_n) fb1 = b1[kc * b_k];
                    // zero what lies outside the problem: k tail in both operands, rows/columns past the edge
                    if (!(kok && r0ok)) fa0 = 0;
                    if (!(kok && r1ok)) fa1 = 0;
                    if (!(kok && c0ok)) fb0 = 0;
                    if (!(kok && c1ok)) fb1 = 0;
                    acc[0][0] = Mf<T>::mfma(fa0, fb0, acc[0][0]);
                    if (two_n) acc[0][1] = Mf<T>::mfma(fa0, fb1, acc[0][1]);
                    if (two_m) {
                        acc[1][0] = Mf<T>::mfma(fa1, fb0, acc[1][0]);
                        if (two_n) acc[1][1] = Mf<T>::mfma(fa1, fb1, acc[1][1]);
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = m0 + 16 * i + Mf<T>::acc_row(lane, r);
#pragma unroll
                        for (int j = 0; j < 2; ++j) {
                            const int col = n0 + 16 * j + fr;
                            if (row < g.m && col < g.n) {
                                T *dst = c + (int64_t)row * g.ldc + col;
                                T v = alpha * acc[i][j][r];
                                if (beta != (T)0) v += beta * *dst;
                                *dst = v;
                            }
                        }
                    }
            }
        }
    }
}

}  // namespace

namespace {
// Equal products of at most 16 x 16, four per wave: each of the wave's four accumulators belongs to a product of its
// own, so the descriptor fetch, the address arithmetic and the loop are paid once for four products and eight loads are
// in flight per k-step instead of two (100,000 products of 8 x 8 x 8 are bound by that fixed work per wave, not by
// the 150 MB they move).  m, n, k and the leading dimensions come from the first descriptor; the offsets from each.
template <typename T, bool AT, bool BT>
__global__ void __launch_bounds__(256) grouped_small_equal_kernel(const SmallGemm *__restrict__ problems, int count, const T *A,
                                                                  const T *B, T *C, T alpha, T beta)
{
    typedef typename Mf<T>::acc_t acc_t;
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int nwaves = (gridDim.x * blockDim.x) >> 6;
    const int fr = lane & 15, fk = lane >> 4;
    const SmallGemm g0 = problems[0];
    const int m = g0.m, n = g0.n, kdim = g0.k;
    const int64_t a_row = AT ? 1 : g0.lda, a_k = AT ? g0.lda : 1, b_k = BT ? 1 : g0.ldb, b_col = BT ? g0.ldb : 1;
    const bool rok = fr < m, cok = fr < n;
    const int64_t a_lane = (int64_t)(rok ? fr : 0) * a_row, b_lane = (int64_t)(cok ? fr : 0) * b_col;

    for (int q = 4 * wave; q < count; q += 4 * nwaves) {
        const T *a[4];
        const T *b[4];
        T *c[4];
        bool live[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            live[s] = q + s < count;
            const SmallGemm g = problems[live[s] ? q + s : q];
            a[s] = A + g.off_a + a_lane;
            b[s] = B + g.off_b + b_lane;
            c[s] = C + g.off_c;
        }
        acc_t acc[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[s] = acc_t{0, 0, 0, 0};
        for (int k0 = 0; k0 < kdim; k0 += 4) {
            const int k = k0 + fk;
            const bool kok = k < kdim;
            const int kc = kok ? k : 0;
            T fa[4], fb[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                fa[s] = a[s][kc * a_k];
                fb[s] = b[s][kc * b_k];
            }
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                // zero what lies outside the product: the k tail, rows / columns past the edge
                if (!(kok && rok)) fa[s] = 0;
                if (!(kok && cok)) fb[s] = 0;
                acc[s] = Mf<T>::mfma(fa[s], fb[s], acc[s]);
            }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (!live[s]) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = Mf<T>::acc_row(lane, r);
                if (row < m && cok) {
                    T *dst = c[s] + (int64_t)row * g0.ldc + fr;
                    T v = alpha * acc[s][r];
                    if (beta != (T)0) v += beta * *dst;
                    *dst = v;
                }
            }
        }
    }
}

}  // namespace

namespace {
template <typename T, bool AT, bool BT>
void launch_one(const SmallGemm *problems, int count, const void *A, const void *B, void *C, double alpha, double beta, int blocks,
                hipStream_t stream, bool small_equal)
{
    if (small_equal)
        hipLaunchKernelGGL((grouped_small_equal_kernel<T, AT, BT>), dim3(blocks), dim3(256), 0, stream, problems, count,
                           static_cast<const T *>(A), static_cast<const T *>(B), static_cast<T *>(C), (T)alpha, (T)beta);
    else
        hipLaunchKernelGGL((grouped_kernel<T, AT, BT>), dim3(blocks), dim3(256), 0, stream, problems, count, static_cast<const T *>(A),
                           static_cast<const T *>(B), static_cast<T *>(C), (T)alpha, (T)beta);
}
template <typename T>
void launch_forms(bool at, bool bt, const SmallGemm *problems, int count, const void *A, const void *B, void *C, double alpha,
                  double beta, int blocks, hipStream_t stream, bool small_equal)
{
    if (!at && !bt) launch_one<T, false, false>(problems, count, A, B, C, alpha, beta, blocks, stream, small_equal);
    else if (!at) launch_one<T, false, true>(problems, count, A, B, C, alpha, beta, blocks, stream, small_equal);
    else if (!bt) launch_one<T, true, false>(problems, count, A, B, C, alpha, beta, blocks, stream, small_equal);
    else launch_one<T, true, true>(problems, count, A, B, C, alpha, beta, blocks, stream, small_equal);
}
}  // namespace

hipError_t launch_grouped(int dtype, const SmallGemm *problems, int count, const void *A, const void *B, void *C,
                          double alpha, double beta, hipStream_t stream, bool a_outer, bool b_kmajor, bool small_equal)
{
    if (count <= 0) return hipSuccess;
    // four waves per workgroup; one problem per wave and pass, or four of the equal small ones
    const int per_block = small_equal ? 16 : 4;
    int blocks = (count + per_block - 1) / per_block;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (dtype == 0) launch_forms<double>(a_outer, b_kmajor, problems, count, A, B, C, alpha, beta, blocks, stream, small_equal);
    else if (dtype == 1) launch_forms<float>(a_outer, b_kmajor, problems, count, A, B, C, alpha, beta, blocks, stream, small_equal);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace ctc
