// MFMA GEMM for tensor contractions on gfx950, with the mode permutation fused
// into the operand loads.
//
// The planner hands over element-offset tables (row/k/col per operand), so the
// kernel never sees tensor modes: A(m,k) lives at A[row_a[m] + k_a[k]].  Each
// operand is staged through registers into a double-buffered LDS tile whose
// shape depends on which GEMM dimension is unit-stride in memory:
//   K-major    (unit stride along K):      LDS image [row][BK + pad], loads are
//                                          16-byte vectors along K;
//   outer-major(unit stride along M or N): LDS image [k][rows + 16], loads are
//                                          16-byte vectors along the rows;
//   scalar     (no usable unit stride):    K-major image, element loads.
// Both images are read back one element per lane in the MFMA 16x16x4 operand
// pattern (lane l: row l&15, k l>>4) without bank conflicts for 8-byte elements:
// the K-major row stride of BK+2 doubles walks the 64 dword banks in steps of 36,
// the outer-major stride of rows+16 doubles offsets consecutive k by 32 banks.
//
// One workgroup = WM x WN waves; each wave owns a (BM/WM) x (BN/WN) block of C as
// TM x TN accumulators of v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32.
// Loop per K-tile: issue global loads of tile t+1 into registers, run the MFMAs of
// tile t from LDS, write the registers into the other LDS buffer, one barrier.
#include "kernels.hpp"

namespace ctc {

namespace {

constexpr int kLayOuterVec = 1, kLayScalar = 2;

template <typename T> struct Traits;

template <> struct Traits<double> {
    typedef double acc_t __attribute__((ext_vector_type(4)));
    typedef double vec_t __attribute__((ext_vector_type(2)));
    static constexpr int VEC = 2;
    static constexpr int PAD_K = 2;
    static __device__ __forceinline__ acc_t mfma(double a, double b, acc_t c)
    {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    // accumulator register j of lane `lane` holds row acc_row, column lane & 15
    static __device__ __forceinline__ int acc_row(int lane, int j) { return (lane >> 4) + 4 * j; }
};

template <> struct Traits<float> {
    typedef float acc_t __attribute__((ext_vector_type(4)));
    typedef float vec_t __attribute__((ext_vector_type(4)));
    static constexpr int VEC = 4;
    static constexpr int PAD_K = 4;
    static __device__ __forceinline__ acc_t mfma(float a, float b, acc_t c)
    {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ int acc_row(int lane, int j) { return (lane >> 4) * 4 + j; }
};

template <typename T, int VEC> struct VecOf { typedef typename Traits<T>::vec_t type; };
template <typename T> struct VecOf<T, 1> { typedef T type; };

template <typename V> __device__ __forceinline__ V zero_of()
{
    V v = {};
    return v;
}

// Stages a BR x BK operand tile: global -> registers -> LDS.
// "line": the dimension enumerated across passes (rows when K-major, k when outer-major);
// "vector position": the dimension enumerated across neighbouring lanes.
template <typename T, int LAY, int BR, int BK, int NT> struct Stager {
    static constexpr bool KMAJ = (LAY != kLayOuterVec);
    static constexpr int VEC = (LAY == kLayScalar) ? 1 : Traits<T>::VEC;
    static constexpr int LDS_STRIDE = KMAJ ? BK + Traits<T>::PAD_K : BR + 16;
    static constexpr int LDS_ELEMS = KMAJ ? BR * LDS_STRIDE : BK * LDS_STRIDE;
    static constexpr int VPL = (KMAJ ? BK : BR) / VEC;        // vectors per line
    static constexpr int LINES_PER_PASS = NT / VPL;
    static constexpr int LINES = KMAJ ? BR : BK;
    static constexpr int NV = LINES / LINES_PER_PASS;          // vectors per thread per tile
    static_assert(NT % VPL == 0 && LINES % LINES_PER_PASS == 0 && NV >= 1, "tile/thread mismatch");
    typedef typename VecOf<T, VEC>::type vec_t;

    // offsets that stay fixed for the whole block (rows) and per K-tile (k)
    static constexpr int NFIX = KMAJ ? NV : 1;
    static constexpr int NSTEP = KMAJ ? 1 : NV;
    int64_t fix_off[NFIX];
    bool fix_ok[NFIX];
    int64_t step_off[NSTEP];
    bool step_ok[NSTEP];
    vec_t regs[NV];
    bool regs_ok[NV];  // validity of the tile held in regs (step_off may already describe a later tile)
    int vpos, line0;

    __device__ __forceinline__ void init(const int64_t *__restrict__ tab_rows, int64_t r0, int64_t R, int t)
    {
        vpos = t % VPL;
        line0 = t / VPL;
#pragma unroll
        for (int i = 0; i < NFIX; ++i) {
            int64_t r = KMAJ ? r0 + line0 + i * LINES_PER_PASS : r0 + (int64_t)vpos * VEC;
            fix_ok[i] = r < R;
            int64_t rc = fix_ok[i] ? r : (R - VEC);
            fix_off[i] = tab_rows[rc];
        }
    }

    __device__ __forceinline__ void fetch_step(const int64_t *__restrict__ tab_k, int64_t k0, int64_t K)
    {
#pragma unroll
        for (int i = 0; i < NSTEP; ++i) {
            int64_t k = KMAJ ? k0 + (int64_t)vpos * VEC : k0 + line0 + i * LINES_PER_PASS;
            step_ok[i] = k < K;
            int64_t kc = step_ok[i] ? k : (K - VEC);
            step_off[i] = tab_k[kc];
        }
    }

    __device__ __forceinline__ void issue(const T *__restrict__ base)
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int64_t off = fix_off[KMAJ ? i : 0] + step_off[KMAJ ? 0 : i];
            regs_ok[i] = fix_ok[KMAJ ? i : 0] && step_ok[KMAJ ? 0 : i];
            regs[i] = *reinterpret_cast<const vec_t *>(base + off);
        }
    }

    __device__ __forceinline__ void commit(T *__restrict__ lds)
    {
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            vec_t v = regs_ok[i] ? regs[i] : zero_of<vec_t>();
            const int line = line0 + i * LINES_PER_PASS;
            *reinterpret_cast<vec_t *>(lds + line * LDS_STRIDE + vpos * VEC) = v;
        }
    }

    // MFMA operand element for tile row rr (0..BR) and k index kq (0..BK)
    static __device__ __forceinline__ T frag(const T *__restrict__ lds, int rr, int kq)
    {
        return KMAJ ? lds[rr * LDS_STRIDE + kq] : lds[kq * LDS_STRIDE + rr];
    }
};

template <typename T, int LA, int LB, int BM, int BN, int BK, int WM, int WN>
__global__ void __launch_bounds__(64 * WM * WN, 2) gemm_kernel(const GemmParams p)
{
    constexpr int NT = 64 * WM * WN;
    constexpr int TM = BM / WM / 16;
    constexpr int TN = BN / WN / 16;
    typedef Traits<T> Tr;
    typedef typename Tr::acc_t acc_t;
    typedef Stager<T, LA, BM, BK, NT> SA;
    typedef Stager<T, LB, BN, BK, NT> SB;

    __shared__ __attribute__((aligned(16))) T smem[2 * (SA::LDS_ELEMS + SB::LDS_ELEMS)];
    constexpr int STAGE_ELEMS = SA::LDS_ELEMS + SB::LDS_ELEMS;

    const int t = threadIdx.x;
    const int lane = t & 63;
    const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm0 = (wave / WN) * (TM * 16);
    const int wn0 = (wave % WN) * (TN * 16);

    // Block -> tile.  Blocks b and b+8 share an XCD (and its L2): give each XCD a
    // contiguous run of tile ids, then walk tiles in groups of `group` tile rows so
    // that concurrently running blocks share A and B panels.
    const int nblk = p.nbm * p.nbn;
    int sid;
    {
        const int bid = blockIdx.x;
        const int xcd = bid & 7, q = nblk >> 3, r = nblk & 7;
        sid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
    const int width = p.group * p.nbn;
    const int first_m = (sid / width) * p.group;
    const int gsz = min(p.nbm - first_m, p.group);
    const int tm = first_m + (sid % width) % gsz;
    const int tn = (sid % width) / gsz;
    const int64_t m0 = (int64_t)tm * BM;
    const int64_t n0 = (int64_t)tn * BN;
    const int bz = blockIdx.y;

    const T *A = static_cast<const T *>(p.A) + p.bat_a[bz];
    const T *B = static_cast<const T *>(p.B) + p.bat_b[bz];
    T *C = static_cast<T *>(p.C) + p.bat_c[bz];

    SA sa;
    SB sb;
    sa.init(p.row_a, m0, p.M, t);
    sb.init(p.col_b, n0, p.N, t);

    acc_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) acc[i][j] = zero_of<acc_t>();

    const int nk = (int)((p.K + BK - 1) / BK);

    sa.fetch_step(p.k_a, 0, p.K);
    sb.fetch_step(p.k_b, 0, p.K);
    sa.issue(A);
    sb.issue(B);
    sa.commit(smem);
    sb.commit(smem + SA::LDS_ELEMS);
    if (nk > 1) {
        sa.fetch_step(p.k_a, BK, p.K);
        sb.fetch_step(p.k_b, BK, p.K);
    }
    __syncthreads();

    const int frow = lane & 15;
    const int fk = lane >> 4;

    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = kt + 1 < nk;
        if (more) {
            sa.issue(A);
            sb.issue(B);
            if (kt + 2 < nk) {
                // offsets for the tile after next: their latency hides under the MFMAs
                sa.fetch_step(p.k_a, (int64_t)(kt + 2) * BK, p.K);
                sb.fetch_step(p.k_b, (int64_t)(kt + 2) * BK, p.K);
            }
        }
        const T *la = smem + cur * STAGE_ELEMS;
        const T *lb = la + SA::LDS_ELEMS;
#pragma unroll
        for (int kk = 0; kk < BK / 4; ++kk) {
            T af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = SA::frag(la, wm0 + 16 * i + frow, kk * 4 + fk);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = SB::frag(lb, wn0 + 16 * j + frow, kk * 4 + fk);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = Tr::mfma(af[i], bf[j], acc[i][j]);
        }
        if (more) {
            T *na = smem + (cur ^ 1) * STAGE_ELEMS;
            sa.commit(na);
            sb.commit(na + SA::LDS_ELEMS);
        }
        __syncthreads();
    }

    // Epilogue: lanes 0..15 of each quarter-wave hold 16 consecutive N positions of
    // one row, so every store instruction writes four contiguous runs of C.
    const T alpha = (T)p.alpha, beta = (T)p.beta;
    int64_t coff[TN];
    bool cok[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int64_t n = n0 + wn0 + 16 * j + frow;
        cok[j] = n < p.N;
        coff[j] = p.col_c[cok[j] ? n : 0];
    }
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t m = m0 + wm0 + 16 * i + Tr::acc_row(lane, r);
            const bool rok = m < p.M;
            const int64_t roff = p.row_c[rok ? m : 0];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                if (rok && cok[j]) {
                    T *dst = C + roff + coff[j];
                    T v = alpha * acc[i][j][r];
                    if (beta != (T)0) v += beta * *dst;
                    *dst = v;
                }
            }
        }
    }
}

template <typename T, int LA, int LB, int BM, int BN>
hipError_t launch_one(const GemmParams &p, hipStream_t stream)
{
    constexpr int BK = 16, WM = 2, WN = 2;
    GemmParams q = p;
    q.nbm = (int)((p.M + BM - 1) / BM);
    q.nbn = (int)((p.N + BN - 1) / BN);
    if (q.group < 1) q.group = 1;
    dim3 grid((unsigned)(q.nbm * q.nbn), (unsigned)p.batch, 1);
    dim3 block(64 * WM * WN, 1, 1);
    hipLaunchKernelGGL((gemm_kernel<T, LA, LB, BM, BN, BK, WM, WN>), grid, block, 0, stream, q);
    return hipGetLastError();
}

template <typename T, int BM, int BN>
hipError_t launch_layouts(int la, int lb, const GemmParams &p, hipStream_t stream)
{
    switch (la * 3 + lb) {
    case 0: return launch_one<T, 0, 0, BM, BN>(p, stream);
    case 1: return launch_one<T, 0, 1, BM, BN>(p, stream);
    case 2: return launch_one<T, 0, 2, BM, BN>(p, stream);
    case 3: return launch_one<T, 1, 0, BM, BN>(p, stream);
    case 4: return launch_one<T, 1, 1, BM, BN>(p, stream);
    case 5: return launch_one<T, 1, 2, BM, BN>(p, stream);
    case 6: return launch_one<T, 2, 0, BM, BN>(p, stream);
    case 7: return launch_one<T, 2, 1, BM, BN>(p, stream);
    case 8: return launch_one<T, 2, 2, BM, BN>(p, stream);
    default: return hipErrorInvalidValue;
    }
}

template <typename T>
hipError_t launch_tiles(int la, int lb, int bm, int bn, const GemmParams &p, hipStream_t stream)
{
    if (bm == 128 && bn == 128) return launch_layouts<T, 128, 128>(la, lb, p, stream);
    if (bm == 64 && bn == 64) return launch_layouts<T, 64, 64>(la, lb, p, stream);
    return hipErrorInvalidValue;
}

}  // namespace

hipError_t launch_gemm(int dtype, int layout_a, int layout_b, int bm, int bn, const GemmParams &p,
                       hipStream_t stream)
{
    if (layout_a < 0 || layout_a > 2 || layout_b < 0 || layout_b > 2) return hipErrorInvalidValue;
    if (p.M <= 0 || p.N <= 0 || p.K <= 0 || p.batch <= 0) return hipErrorInvalidValue;
    if (p.batch > 65535) return hipErrorInvalidValue;
    if (dtype == 0) return launch_tiles<double>(layout_a, layout_b, bm, bn, p, stream);
    if (dtype == 1) return launch_tiles<float>(layout_a, layout_b, bm, bn, p, stream);
    return hipErrorInvalidValue;
}

}  // namespace ctc
