// Skinny contractions on gfx950: C(m,n) = alpha * sum_k A(m,k) B(k,n) + beta * C(m,n) with a
// handful of rows (M <= 16), a short K and a very long N.  Such a contraction is a
// stream over B and C: 2*M*K flops per (K + M) elements moved, far below the machine
// balance, so the kernels are built like the mode-permutation kernels (16-byte accesses,
// 256 contiguous bytes per 16 lanes on both sides) and do the arithmetic on the vector
// ALU while the data passes through registers.  The small operand A sits in LDS.
//
// same kernel:      B and C are contiguous along the same N mode; one 16-byte vector of
//                   n per thread and k, M vector accumulators, direct stores.
// transpose kernel: B is contiguous along x, C along y (different modes).  A workgroup
//                   owns a (16V x 16V) patch of (x, y); thread (vx, vq) accumulates a
//                   V x V micro-tile for every m, then each m goes through the
//                   XOR-swizzled LDS image of the permutation kernel so that stores run
//                   along y.
#include "kernels.hpp"

namespace ctc {

namespace {

constexpr int kThreads = 256;
constexpr int kMaxMK = 2048;

template <typename T> struct VecT;
template <> struct VecT<double> {
    typedef double type __attribute__((ext_vector_type(2)));
    static constexpr int V = 2;
};
template <> struct VecT<float> {
    typedef float type __attribute__((ext_vector_type(4)));
    static constexpr int V = 4;
};

template <typename T>
__device__ __forceinline__ void load_small_operand(T *sA, const SkinnyParams &p, const T *A)
{
    for (int i = threadIdx.x; i < p.M * p.K; i += kThreads) {
        const int m = i / p.K, k = i % p.K;
        sA[i] = A[p.row_a[m] + p.k_a[k]];
    }
    __syncthreads();
}

// (at least four workgroups per CU: without the bound the MMAX = 8 instantiations are compiled to 256
// registers, one wave per SIMD, and stream at 0.6 - 1.1 TB/s where MMAX = 4 and 16 reach 3 - 4)
// NT: streaming hint on the loads of the large operand and the stores of C (SkinnyParams::stream_hint)
template <typename T, int MMAX, bool NT>
__global__ void __launch_bounds__(kThreads, 4) skinny_same_kernel(const SkinnyParams p)
{
    typedef typename VecT<T>::type vec_t;
    constexpr int V = VecT<T>::V;
    __shared__ T sA[kMaxMK];
    const int bz = blockIdx.y;
    const T *A = static_cast<const T *>(p.A) + p.bat_a[bz];
    const T *B = static_cast<const T *>(p.B) + p.bat_b[bz];
    T *C = static_cast<T *>(p.C) + p.bat_c[bz];
    load_small_operand(sA, p, A);

    const int64_t nvec = p.X / V;
    const T alpha = (T)p.alpha, beta = (T)p.beta;
    for (int64_t i = blockIdx.x * (int64_t)kThreads + threadIdx.x; i < nvec; i += (int64_t)gridDim.x * kThreads) {
        const int64_t n = i * V;
        const T *bsrc = B + p.bx[n];
        T *cdst = C + p.cx[n];
        vec_t acc[MMAX];
#pragma unroll
        for (int m = 0; m < MMAX; ++m) acc[m] = vec_t(0);
        if (NT) {
            // operands beyond the Infinity Cache: eight rows of the large operand requested before the first is
            // used (16 M columns of fp32: 3.75 -> 4.3 TB/s with the hint; cache-resident sizes lose 6 % this way)
            for (int k0 = 0; k0 < p.K; k0 += 8) {
                vec_t b[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (k0 + u < p.K) b[u] = __builtin_nontemporal_load(reinterpret_cast<const vec_t *>(bsrc + p.k_b[k0 + u]));
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (k0 + u < p.K) {
#pragma unroll
                        for (int m = 0; m < MMAX; ++m)
                            if (m < p.M) acc[m] += sA[m * p.K + k0 + u] * b[u];
                    }
            }
        } else {
#pragma unroll 4
            for (int k = 0; k < p.K; ++k) {
                const vec_t b = *reinterpret_cast<const vec_t *>(bsrc + p.k_b[k]);
#pragma unroll
                for (int m = 0; m < MMAX; ++m)
                    if (m < p.M) acc[m] += sA[m * p.K + k] * b;
            }
        }
#pragma unroll
        for (int m = 0; m < MMAX; ++m) {
            if (m < p.M) {
                vec_t *dst = reinterpret_cast<vec_t *>(cdst + p.row_c[m]);
                vec_t v = alpha * acc[m];
                if (beta != (T)0) v += beta * *dst;
                if (NT) __builtin_nontemporal_store(v, dst);
                else *dst = v;
            }
        }
    }
}

// NT: loads of the large operand and stores of C carry the streaming hint (operands that do not fit the 256 MiB
// Infinity Cache: 4.19 -> 4.62 TB/s on 2 x 2 GiB, 4.25 -> 4.53 on 2 x 512 MiB; 3 % slower on 2 x 64 MiB)
template <typename T, int MMAX, bool NT>
__global__ void __launch_bounds__(kThreads) skinny_transpose_kernel(const SkinnyParams p, int ntx, int nty)
{
    typedef typename VecT<T>::type vec_t;
    constexpr int V = VecT<T>::V;
    constexpr int TT = 16 * V;  // patch edge; one LDS row = 256 bytes
    __shared__ __attribute__((aligned(16))) T tile[TT * TT];
    __shared__ T sA[kMaxMK];

    const int bz = blockIdx.y;
    const T *A = static_cast<const T *>(p.A) + p.bat_a[bz];
    const T *B = static_cast<const T *>(p.B) + p.bat_b[bz];
    T *C = static_cast<T *>(p.C) + p.bat_c[bz];
    load_small_operand(sA, p, A);

    const int64_t bid = blockIdx.x;
    const int tx = (int)(bid % ntx);
    const int ty = (int)((bid / ntx) % nty);
    const int64_t o = bid / ((int64_t)ntx * nty);
    const int64_t x0 = (int64_t)tx * TT, y0 = (int64_t)ty * TT;
    const int lo = threadIdx.x & 15;
    const int hi = threadIdx.x >> 4;

    // load side: vector along x at x0 + V*lo, rows y0 + V*hi + r
    const int64_t lx = x0 + (int64_t)V * lo;
    const bool lx_ok = lx < p.X;
    const T *bsrc = B + p.bo[o] + p.bx[lx_ok ? lx : 0];
    int64_t by[V];
    bool ly_ok[V];
#pragma unroll
    for (int r = 0; r < V; ++r) {
        const int64_t y = y0 + (int64_t)V * hi + r;
        ly_ok[r] = lx_ok && y < p.Y;
        by[r] = p.by[y < p.Y ? y : 0];
    }
    // store side: vector along y at y0 + V*lo, rows x0 + hi + 16*r
    const int64_t sy = y0 + (int64_t)V * lo;
    const bool sy_ok = sy < p.Y;
    T *cdst = C + p.co[o] + p.cy[sy_ok ? sy : 0];
    int64_t cx[V];
    bool sx_ok[V];
#pragma unroll
    for (int r = 0; r < V; ++r) {
        const int64_t x = x0 + hi + 16 * r;
        sx_ok[r] = sy_ok && x < p.X;
        cx[r] = p.cx[x < p.X ? x : 0];
    }

    vec_t acc[MMAX][V];  // acc[m][r]: V consecutive x of row y = V*hi + r
#pragma unroll
    for (int m = 0; m < MMAX; ++m)
#pragma unroll
        for (int r = 0; r < V; ++r) acc[m][r] = vec_t(0);

#pragma unroll 2
    for (int k = 0; k < p.K; ++k) {
        const T *src = bsrc + p.k_b[k];
        vec_t b[V];
#pragma unroll
        for (int r = 0; r < V; ++r) {
            if (NT) b[r] = ly_ok[r] ? __builtin_nontemporal_load(reinterpret_cast<const vec_t *>(src + by[r])) : vec_t(0);
            else b[r] = ly_ok[r] ? *reinterpret_cast<const vec_t *>(src + by[r]) : vec_t(0);
        }
#pragma unroll
        for (int m = 0; m < MMAX; ++m) {
            if (m < p.M) {
                const T a = sA[m * p.K + k];
#pragma unroll
                for (int r = 0; r < V; ++r) acc[m][r] += a * b[r];
            }
        }
    }

    const T alpha = (T)p.alpha, beta = (T)p.beta;
#pragma unroll
    for (int m = 0; m < MMAX; ++m) {
        if (m < p.M) {  // uniform across the workgroup
            if (m > 0) __syncthreads();
            // micro-tile transpose into the [x][y] image, 16-byte chunks swizzled with the row's vx
#pragma unroll
            for (int c = 0; c < V; ++c) {
                vec_t col;
#pragma unroll
                for (int r = 0; r < V; ++r) col[r] = acc[m][r][c];
                *reinterpret_cast<vec_t *>(&tile[(V * lo + c) * TT + (hi ^ lo) * V]) = col;
            }
            __syncthreads();
            T *dst_m = cdst + p.row_c[m];
#pragma unroll
            for (int r = 0; r < V; ++r) {
                const int xr = hi + 16 * r;
                vec_t v = alpha * *reinterpret_cast<const vec_t *>(&tile[xr * TT + (lo ^ (xr / V)) * V]);
                if (sx_ok[r]) {
                    vec_t *dst = reinterpret_cast<vec_t *>(dst_m + cx[r]);
                    if (beta != (T)0) v += beta * *dst;
                    if (NT) __builtin_nontemporal_store(v, dst);
                    else *dst = v;
                }
            }
        }
    }
}

template <typename T, int MMAX> hipError_t launch_t(const SkinnyParams &p, hipStream_t stream)
{
    constexpr int V = VecT<T>::V;
    if (p.transpose) {
        constexpr int TT = 16 * V;
        const int64_t ntx = (p.X + TT - 1) / TT, nty = (p.Y + TT - 1) / TT;
        const int64_t blocks = ntx * nty * p.O;
        if (blocks <= 0 || blocks > 0x7FFFFFFF || p.batch > 65535) return hipErrorInvalidValue;
        if (p.stream_hint)
            hipLaunchKernelGGL((skinny_transpose_kernel<T, MMAX, true>), dim3((unsigned)blocks, (unsigned)p.batch), dim3(kThreads), 0,
                               stream, p, (int)ntx, (int)nty);
        else
            hipLaunchKernelGGL((skinny_transpose_kernel<T, MMAX, false>), dim3((unsigned)blocks, (unsigned)p.batch), dim3(kThreads), 0,
                               stream, p, (int)ntx, (int)nty);
    } else {
        int64_t blocks = (p.X / V + kThreads - 1) / kThreads;
        if (blocks > 256 * 32) blocks = 256 * 32;
        if (blocks <= 0 || p.batch > 65535) return hipErrorInvalidValue;
        if (p.stream_hint)
            hipLaunchKernelGGL((skinny_same_kernel<T, MMAX, true>), dim3((unsigned)blocks, (unsigned)p.batch), dim3(kThreads), 0, stream, p);
        else
            hipLaunchKernelGGL((skinny_same_kernel<T, MMAX, false>), dim3((unsigned)blocks, (unsigned)p.batch), dim3(kThreads), 0, stream, p);
    }
    return hipGetLastError();
}

template <typename T> hipError_t launch_m(const SkinnyParams &p, hipStream_t stream)
{
    if (p.M <= 4) return launch_t<T, 4>(p, stream);
    if (p.M <= 8) return launch_t<T, 8>(p, stream);
    if (p.M <= 16) return launch_t<T, 16>(p, stream);
    return hipErrorInvalidValue;
}

}  // namespace

hipError_t launch_skinny(int dtype, const SkinnyParams &p, hipStream_t stream)
{
    if (p.M < 1 || p.K < 1 || (int64_t)p.M * p.K > kMaxMK) return hipErrorInvalidValue;
    if (dtype == 0) return launch_m<double>(p, stream);
    if (dtype == 1) return launch_m<float>(p, stream);
    return hipErrorInvalidValue;
}

}  // namespace ctc
