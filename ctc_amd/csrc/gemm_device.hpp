// Device-side building blocks shared by the GEMM kernels: MFMA traits per element type
// and the operand stager (global -> registers -> LDS through element-offset tables).
#pragma once

#include "kernels.hpp"

#include <cstdint>
#include <type_traits>

namespace ctc {
namespace gemm_detail {

constexpr int kLayOuterVec = 1, kLayScalar = 2;

template <typename T> struct Traits;

// value of the lane's quad neighbour chosen by a DPP quad permutation
template <int CTRL> __device__ __forceinline__ int dpp_quad(int x)
{
    return __builtin_amdgcn_update_dpp(0, x, CTRL, 0xF, 0xF, false);
}
constexpr int kDppSwap1 = 0xB1;  // quad_perm [1,0,3,2]: lane ^ 1
constexpr int kDppSwap2 = 0x4E;  // quad_perm [2,3,0,1]: lane ^ 2

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

    // 16-byte stores of an accumulator: lanes l and l^1 hold neighbouring columns of the
    // same four rows; after one exchange the even lane owns both columns of row register
    // 2s and the odd lane both columns of row register 2s+1.  CV columns per store, NCV
    // stores per accumulator; store s of a lane covers row cvec_row, columns
    // (lane & 15 & ~(CV-1)) .. +CV-1.  Every lane of the wave must call cvec.
    static constexpr int CV = 2, NCV = 2;
    static __device__ __forceinline__ int cvec_row(int lane, int s) { return (lane >> 4) + 4 * (2 * s + (lane & 1)); }
    static __device__ __forceinline__ vec_t cvec(const acc_t &a, int lane, int s)
    {
        const bool odd = lane & 1;
        const double lo_row = a[2 * s], hi_row = a[2 * s + 1];
        const double send = odd ? lo_row : hi_row;
        const double recv = __hiloint2double(dpp_quad<kDppSwap1>(__double2hiint(send)),
                                             dpp_quad<kDppSwap1>(__double2loint(send)));
        vec_t v;
        v[0] = odd ? recv : lo_row;
        v[1] = odd ? hi_row : recv;
        return v;
    }
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

    // 16-byte stores: the four lanes of a quad hold four neighbouring columns of the same
    // four rows; a 4x4 transpose inside the quad (two exchanges) leaves lane q with the four
    // columns of row register q.
    static constexpr int CV = 4, NCV = 1;
    static __device__ __forceinline__ int cvec_row(int lane, int) { return (lane >> 4) * 4 + (lane & 3); }
    static __device__ __forceinline__ vec_t cvec(const acc_t &a, int lane, int)
    {
        const bool b0 = lane & 1, b1 = lane & 2;
        // exchange with lane ^ 1: registers whose index bit 0 differs from the lane's bit 0
        float s0 = b0 ? a[0] : a[1], s1 = b0 ? a[2] : a[3];
        float r0 = __int_as_float(dpp_quad<kDppSwap1>(__float_as_int(s0)));
        float r1 = __int_as_float(dpp_quad<kDppSwap1>(__float_as_int(s1)));
        float t0 = b0 ? r0 : a[0], t1 = b0 ? a[1] : r0, t2 = b0 ? r1 : a[2], t3 = b0 ? a[3] : r1;
        // exchange with lane ^ 2: bit 1
        s0 = b1 ? t0 : t2;
        s1 = b1 ? t1 : t3;
        r0 = __int_as_float(dpp_quad<kDppSwap2>(__float_as_int(s0)));
        r1 = __int_as_float(dpp_quad<kDppSwap2>(__float_as_int(s1)));
        vec_t v;
        v[0] = b1 ? r0 : t0;
        v[1] = b1 ? r1 : t1;
        v[2] = b1 ? t2 : r0;
        v[3] = b1 ? t3 : r1;
        return v;
    }
};

template <typename T, int VEC> struct VecOf { typedef typename Traits<T>::vec_t type; };
template <typename T> struct VecOf<T, 1> { typedef T type; };

template <typename V> __device__ __forceinline__ V zero_of()
{
    V v = {};
    return v;
}

// sentinel in a tile's staged C offsets: the row / column lies outside C
constexpr int64_t kNoOffset = INT64_MIN;

// Epilogue of a BM x BN tile held as TM x TN 16 x 16 accumulators per wave (wave origin wm0, wn0; s_coff:
// the tile's BM row offsets, then its BN column offsets, kNoOffset outside C).  partial: the workgroup
// multiplied a part of a tile's K range; its raw sums go to slab `slab_index` of the workspace.  Every
// lane of the wave must call.
template <typename T, int BM, int BN, int TM, int TN>
__device__ __forceinline__ void store_tile_slab(const GemmParams &p, typename Traits<T>::acc_t (&acc)[TM][TN],
                                                const int64_t *s_coff, T *C, int wm0, int wn0, int lane, bool partial,
                                                int64_t slab_index)
{
    typedef Traits<T> Tr;
    const int frow = lane & 15;
    // Epilogue: lanes 0..15 of each quarter-wave hold 16 consecutive N positions of
    // one row, so every store instruction writes four contiguous runs of C.  The
    // beta test is hoisted so the common beta == 0 path is a straight run of stores.
    if (partial) {
        // raw accumulators of this K chunk: tile-shaped slab [BM][BN] in the workspace
        T *slab = static_cast<T *>(p.ws) + slab_index * (BM * BN);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = wm0 + 16 * i + Tr::acc_row(lane, r);
#pragma unroll
                for (int j = 0; j < TN; ++j) slab[row * BN + wn0 + 16 * j + frow] = acc[i][j][r];
            }
        return;
    }
    const T alpha = (T)p.alpha, beta = (T)p.beta;
    int64_t coff[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) coff[j] = s_coff[BM + wn0 + 16 * j + frow];
    int64_t roff[TM][4];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) roff[i][r] = s_coff[wm0 + 16 * i + Tr::acc_row(lane, r)];
    auto store_rows = [&](auto with_beta_t) {
        constexpr bool WITH_BETA = decltype(with_beta_t)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#ifdef CTC_EXP_NOSTORE
                    if (roff[i][r] != kNoOffset && coff[j] != kNoOffset && acc[i][j][r] == (T)1.2345e30) {
#else
                    if (roff[i][r] != kNoOffset && coff[j] != kNoOffset) {
#endif
                        T *dst = C + roff[i][r] + coff[j];
                        T v = alpha * acc[i][j][r];
                        if (WITH_BETA) v += beta * *dst;
                        *dst = v;
                    }
                }
            }
        }
    };
    // Same with 16-byte stores (see Traits::cvec): half (f64) or a quarter (f32) of the store
    // instructions, which is what bounds the epilogue of contractions with a short K.
    // STREAM: C does not fit the Infinity Cache, so its lines are stored with the non-temporal hint and
    // leave the caches to the operand panels (config-4 contraction with |C| = 32.8 GB: 41.4 -> 47.7 TFLOP/s).
    auto store_rows_vec = [&](auto with_beta_t, auto stream_t) {
        constexpr bool WITH_BETA = decltype(with_beta_t)::value;
        constexpr bool STREAM = decltype(stream_t)::value;
        typedef typename Tr::vec_t vec_t;
        int64_t cvoff[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) cvoff[j] = s_coff[BM + wn0 + 16 * j + (frow & ~(Tr::CV - 1))];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
#pragma unroll
            for (int s = 0; s < Tr::NCV; ++s) {
                const int64_t rvoff = s_coff[wm0 + 16 * i + Tr::cvec_row(lane, s)];
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    vec_t v = Tr::cvec(acc[i][j], lane, s) * alpha;
#ifdef CTC_EXP_NOSTORE
                    if (rvoff != kNoOffset && cvoff[j] != kNoOffset && v[0] == (T)1.2345e30) {
#else
                    if (rvoff != kNoOffset && cvoff[j] != kNoOffset) {
#endif
                        vec_t *dst = reinterpret_cast<vec_t *>(C + rvoff + cvoff[j]);
                        if (WITH_BETA) v += *dst * beta;
                        if (STREAM) __builtin_nontemporal_store(v, dst);
                        else *dst = v;
                    }
                }
            }
        }
    };
    if (p.c_vec > 1) {
        if (beta != (T)0) store_rows_vec(std::true_type(), std::false_type());
        else if (p.c_stream) store_rows_vec(std::false_type(), std::true_type());
        else store_rows_vec(std::false_type(), std::false_type());
    } else {
        if (beta == (T)0) store_rows(std::false_type());
        else store_rows(std::true_type());
    }
}

// The same for K chunk `chunk` of split tile `sid` (every tile from dp_tiles on is cut into p.split chunks).
template <typename T, int BM, int BN, int TM, int TN>
__device__ __forceinline__ void store_tile(const GemmParams &p, typename Traits<T>::acc_t (&acc)[TM][TN],
                                           const int64_t *s_coff, T *C, int wm0, int wn0, int lane, bool partial,
                                           int sid, int chunk)
{
    store_tile_slab<T, BM, BN, TM, TN>(p, acc, s_coff, C, wm0, wn0, lane, partial, (int64_t)(sid - p.dp_tiles) * p.split + chunk);
}

// Stages a BR x BK operand tile: global -> registers -> LDS.
// "line": the dimension enumerated across passes (rows when K-major, k when outer-major);
// "vector position": the dimension enumerated across neighbouring lanes.
// Rows past the operand's extent are clamped to its last rows (their products land in
// C elements that are never stored); only the K tail has to read as zero, and only in
// the last K-tile, so the steady-state path carries no masking at all.  The k tables
// are padded by the host with kKTablePad entries so lookups need no clamping.
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
    int64_t step_off[NSTEP];
    vec_t regs[NV];
    int vpos, line0;

    // rows a load may start at: every row when K-major, multiples of VEC when outer-major
    static constexpr int ROW_VEC = KMAJ ? 1 : VEC;

    __device__ __forceinline__ void init_thread(int t)
    {
        vpos = t % VPL;
        line0 = t / VPL;
    }

    __device__ __forceinline__ void init(const int64_t *__restrict__ tab_rows, int64_t r0, int64_t R, int t)
    {
        init_thread(t);
#pragma unroll
        for (int i = 0; i < NFIX; ++i) {
            const int64_t r = KMAJ ? r0 + line0 + i * LINES_PER_PASS : r0 + (int64_t)vpos * VEC;
            fix_off[i] = tab_rows[r < R ? r : (R - ROW_VEC)];
        }
    }

    // Same offsets from a table of the tile's BR row offsets (entry i: row r0 + i, clamped
    // by whoever filled it), e.g. one staged in LDS.
    __device__ __forceinline__ void load_fix(const int64_t *tile_rows)
    {
#pragma unroll
        for (int i = 0; i < NFIX; ++i) fix_off[i] = tile_rows[KMAJ ? line0 + i * LINES_PER_PASS : vpos * VEC];
    }

    __device__ __forceinline__ void fetch_step(const int64_t *__restrict__ tab_k, int64_t k0)
    {
#pragma unroll
        for (int i = 0; i < NSTEP; ++i)
            step_off[i] = tab_k[KMAJ ? k0 + vpos * VEC : k0 + line0 + i * LINES_PER_PASS];
    }

    __device__ __forceinline__ void issue_one(int i, const T *__restrict__ base)
    {
        const int64_t off = fix_off[KMAJ ? i : 0] + step_off[KMAJ ? 0 : i];
        regs[i] = *reinterpret_cast<const vec_t *>(base + off);
    }

    // k0: first k of the tile held in regs; MASK_K: that tile crosses K
    template <bool MASK_K>
    __device__ __forceinline__ void commit_one(int i, T *__restrict__ lds, int64_t k0, int64_t K)
    {
        vec_t v = regs[i];
        const int line = line0 + i * LINES_PER_PASS;
        if (MASK_K) {
            const int64_t k = KMAJ ? k0 + vpos * VEC : k0 + line;
            if (k >= K) v = zero_of<vec_t>();
        }
        *reinterpret_cast<vec_t *>(lds + line * LDS_STRIDE + vpos * VEC) = v;
    }

    // MFMA operand element for tile row rr (0..BR) and k index kq (0..BK)
    static __device__ __forceinline__ T frag(const T *__restrict__ lds, int rr, int kq)
    {
        return KMAJ ? lds[rr * LDS_STRIDE + kq] : lds[kq * LDS_STRIDE + rr];
    }
};

}  // namespace gemm_detail
}  // namespace ctc
