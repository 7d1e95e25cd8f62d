// Mode permutation kernels for gfx950.
//
// A permutation is described by three fused indices and their element-offset
// tables in the input and the output:
//   x  - runs along the input's fastest modes  (input addresses contiguous in x)
//   y  - runs along the output's fastest modes (output addresses contiguous in y)
//   o  - every remaining mode
//
// Vector kernel (the fast path; 16-byte accesses on both sides).  V = 16 / element
// size, R = 1 or 2.  A workgroup moves a (16VR x 16VR) patch of (x, y): thread (vx, vq) loads,
// for each of the R x R sub-patches, the V rows y = V*vq .. V*vq+V-1 as 16-byte vectors at
// x = V*vx, so 16 neighbouring lanes read 256 contiguous bytes and a patch row is R*256
// contiguous bytes; it transposes its V x V micro-tiles in registers and writes the vectors
// into an LDS image indexed [x][y] whose 16-byte chunks are XOR-swizzled within each 256-byte
// group with the row's vx, which makes both the writes (8 lanes: 8 different chunks) and the
// reads (16 lanes of one row: 16 different chunks) bank-conflict free without padding.
// In the store phase 16 neighbouring lanes write 256 contiguous bytes of the output.
// A workgroup walks o_per_block consecutive `o` values with the same (x, y) patch (the host
// picks 4 when that lengthens the contiguous runs of one side, else 1); the loads of the next
// patch are issued before the stores of the current one.
// The kernel runs at the speed of its access pattern: with the LDS transposition taken out
// (wrong results) every shape measured kept its rate to within 2 %.
// Extents that are no multiple of V (4- and 8-byte elements): the same kernel with vectors at
// element-aligned addresses and an element tail at the end of each run.
//
// Scalar kernel: the same patch scheme with element accesses and a padded LDS tile, for 2-byte
// elements with odd extents and for short odd fastest extents.
// Row kernels: input and output share their fastest mode, nothing to transpose.
//
// The fill kernel generates synthetic operands directly in HBM.
#include "kernels.hpp"

#include <cstdlib>
#include <type_traits>

namespace ctc {

namespace {

constexpr int kTile = 64;
constexpr int kThreads = 256;
#ifndef CTC_PERM_XCD
#define CTC_PERM_XCD 1
#endif

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

typedef u32x4 u32x4_elem_aligned __attribute__((aligned(4)));

// RAGGED: a side's index may end inside a vector (extent not a multiple of V) and rows start at element-
// aligned addresses only.  Whole vectors still move as one 16-byte access (global accesses need dword
// alignment only); the vector that hangs over the end of a run is moved element by element.  The planner
// keeps such a side to ONE mode, so no vector straddles two runs.
template <typename E, int R, bool RAGGED, int NT>
__global__ void __launch_bounds__(kThreads) permute_vec_kernel(const PermuteParams p, int ntx, int nty, int no)
{
    constexpr int V = 16 / sizeof(E);
    constexpr int T = 16 * V * R;  // patch edge in elements; one LDS row = T * sizeof(E) = R * 256 bytes
    static_assert(!RAGGED || sizeof(E) >= 4, "element-aligned vectors need dword alignment");
    using vec_t = typename std::conditional<RAGGED, u32x4_elem_aligned, u32x4>::type;
    __shared__ __attribute__((aligned(16))) E tile[T * T];

    // Workgroups b and b+8 run on one XCD and share its L2.  Hand each XCD a contiguous run of patches:
    // patches next to each other complete each other's DRAM rows, and they then pass through one L2 at
    // about the same time.
    int64_t bid = blockIdx.x;
#if CTC_PERM_XCD
    {
        const int64_t nblk = gridDim.x, q = nblk >> 3, r = nblk & 7, xcd = bid & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    }
#endif
    const int tx = (int)(bid % ntx);
    const int ty = (int)((bid / ntx) % nty);
    const int64_t ob = bid / ((int64_t)ntx * nty);
    const int64_t x0 = (int64_t)tx * T, y0 = (int64_t)ty * T;
    const int lo = threadIdx.x & 15;  // vector index along the contiguous direction
    const int hi = threadIdx.x >> 4;  // 0..15

    // load side: vectors at x = x0 + V*(lo + 16*rx), rows y = y0 + V*(hi + 16*ry) + r
    int nx[R];         // valid elements of the vector
    int64_t in_xo[R];
#pragma unroll
    for (int rx = 0; rx < R; ++rx) {
        const int64_t lx = x0 + (int64_t)V * (lo + 16 * rx);
        nx[rx] = lx + V <= p.X ? V : lx < p.X ? (RAGGED ? (int)(p.X - lx) : V) : 0;
        in_xo[rx] = p.in_x[nx[rx] ? lx : 0];
    }
    int64_t in_yo[R][V];  // -1: row outside
#pragma unroll
    for (int ry = 0; ry < R; ++ry)
#pragma unroll
        for (int r = 0; r < V; ++r) {
            const int64_t y = y0 + (int64_t)V * (hi + 16 * ry) + r;
            in_yo[ry][r] = y < p.Y ? p.in_y[y] : -1;
        }
    // store side: vectors at y = y0 + V*(lo + 16*ry), rows x = x0 + hi + 16*r
    int ny[R];
    int64_t out_yo[R];
#pragma unroll
    for (int ry = 0; ry < R; ++ry) {
        const int64_t sy = y0 + (int64_t)V * (lo + 16 * ry);
        ny[ry] = sy + V <= p.Y ? V : sy < p.Y ? (RAGGED ? (int)(p.Y - sy) : V) : 0;
        out_yo[ry] = p.out_y[ny[ry] ? sy : 0];
    }
    int64_t out_xo[V * R];  // -1: row outside
#pragma unroll
    for (int r = 0; r < V * R; ++r) {
        const int64_t x = x0 + hi + 16 * r;
        out_xo[r] = x < p.X ? p.out_x[x] : -1;
    }

    union Vec { u32x4 v; E e[V]; };
    Vec rows[R][R][V];  // [ry][rx][r]
    const auto load_rows = [&](int64_t o) {
        const E *in = static_cast<const E *>(p.in) + p.in_o[o];
#pragma unroll
        for (int ry = 0; ry < R; ++ry)
#pragma unroll
            for (int r = 0; r < V; ++r)
#pragma unroll
                for (int rx = 0; rx < R; ++rx)
                    if (nx[rx] && in_yo[ry][r] >= 0) {
                        const E *src = in + in_xo[rx] + in_yo[ry][r];
                        if (!RAGGED || nx[rx] == V) {
                            if (NT & 1) rows[ry][rx][r].v = __builtin_nontemporal_load(reinterpret_cast<const vec_t *>(src));
                            else rows[ry][rx][r].v = *reinterpret_cast<const vec_t *>(src);
                        } else {
#pragma unroll
                            for (int c = 0; c < V - 1; ++c)
                                if (c < nx[rx]) rows[ry][rx][r].e[c] = src[c];
                        }
                    }
    };

    const int64_t o_begin = ob * p.o_per_block;
    const int64_t o_end = o_begin + p.o_per_block < p.O ? o_begin + p.o_per_block : p.O;
    load_rows(o_begin);
    for (int64_t o = o_begin; o < o_end; ++o) {
        E *out = static_cast<E *>(p.out) + p.out_o[o];
        if (o != o_begin) __syncthreads();  // previous patch fully read back
        // micro-tile transpose: column c of V loaded rows becomes the vector for one x.  The LDS image is
        // [x][y]; the 16-byte chunks of every 256-byte group of a row are XOR-swizzled with the row's vector
        // index, which makes the writes (8 lanes: 8 different chunks) and the reads (16 lanes of one row:
        // 16 different chunks) bank-conflict free without padding.
#pragma unroll
        for (int ry = 0; ry < R; ++ry)
#pragma unroll
            for (int rx = 0; rx < R; ++rx)
#pragma unroll
                for (int c = 0; c < V; ++c) {
                    Vec col;
#pragma unroll
                    for (int r = 0; r < V; ++r) col.e[r] = rows[ry][rx][r].e[c];
                    const int xr = V * (lo + 16 * rx) + c;  // LDS row
                    const int chunk = (hi ^ lo) + 16 * ry;
                    *reinterpret_cast<u32x4 *>(&tile[xr * T + chunk * V]) = col.v;
                }
        // the next patch's rows are on their way while this one is written out
        if (o + 1 < o_end) load_rows(o + 1);
        __syncthreads();
#pragma unroll
        for (int r = 0; r < V * R; ++r) {
            const int xr = hi + 16 * r;
#pragma unroll
            for (int ry = 0; ry < R; ++ry) {
                const int chunk = (lo ^ ((xr / V) & 15)) + 16 * ry;
                Vec v;
                v.v = *reinterpret_cast<const u32x4 *>(&tile[xr * T + chunk * V]);
                if (ny[ry] && out_xo[r] >= 0) {
                    E *dst = out + out_yo[ry] + out_xo[r];
                    if (!RAGGED || ny[ry] == V) {
                        if (NT & 2) __builtin_nontemporal_store(v.v, reinterpret_cast<vec_t *>(dst));
                        else *reinterpret_cast<vec_t *>(dst) = v.v;
                    } else {
#pragma unroll
                        for (int c = 0; c < V - 1; ++c)
                            if (c < ny[ry]) dst[c] = v.e[c];
                    }
                }
            }
        }
    }
}

template <typename E>
__global__ void __launch_bounds__(kThreads) permute_tiled_kernel(const PermuteParams p, int ntx, int nty)
{
    __shared__ E tile[kTile][kTile + 1];
    const int64_t bid = blockIdx.x;
    const int tx = (int)(bid % ntx);
    const int ty = (int)((bid / ntx) % nty);
    const int64_t o = bid / ((int64_t)ntx * nty);
    const int64_t x0 = (int64_t)tx * kTile, y0 = (int64_t)ty * kTile;
    const int lx = threadIdx.x & 63;
    const int ly = threadIdx.x >> 6;  // 0..3

    const E *in = static_cast<const E *>(p.in) + p.in_o[o];
    E *out = static_cast<E *>(p.out) + p.out_o[o];

    // all sixteen row offsets first, then all sixteen loads in flight at once (a lookup followed by its
    // load, row after row, left one or two loads in flight per thread: 1.6 TB/s)
    constexpr int ROWS = kTile / 4;
    {
        const int64_t x = x0 + lx;
        const bool xok = x < p.X;
        const int64_t xoff = p.in_x[xok ? x : 0];
        int64_t yo[ROWS];
        E v[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int64_t y = y0 + ly + 4 * r;
            yo[r] = (xok && y < p.Y) ? p.in_y[y] : -1;
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
            if (yo[r] >= 0) v[r] = in[xoff + yo[r]];
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
            if (yo[r] >= 0) tile[ly + 4 * r][lx] = v[r];
    }
    __syncthreads();
    {
        const int64_t y = y0 + lx;
        const bool yok = y < p.Y;
        const int64_t yoff = p.out_y[yok ? y : 0];
        int64_t xo[ROWS];
#pragma unroll
        for (int r = 0; r < ROWS; ++r) {
            const int64_t x = x0 + ly + 4 * r;
            xo[r] = (yok && x < p.X) ? p.out_x[x] : -1;
        }
#pragma unroll
        for (int r = 0; r < ROWS; ++r)
            if (xo[r] >= 0) out[yoff + xo[r]] = tile[lx][ly + 4 * r];
    }
}

// Input and output share the fastest mode: copy runs of X contiguous elements.
template <typename E>
__global__ void __launch_bounds__(kThreads) permute_rows_kernel(const PermuteParams p, int chunks)
{
    const int64_t bid = blockIdx.x;
    const int64_t o = bid / chunks;
    const int64_t x0 = (bid % chunks) * (int64_t)(kThreads * 4);
    const E *in = static_cast<const E *>(p.in) + p.in_o[o];
    E *out = static_cast<E *>(p.out) + p.out_o[o];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t x = x0 + threadIdx.x + r * kThreads;
        if (x < p.X) out[x] = in[x];
    }
}

// Same, 16 bytes per lane (the tables are in elements), kRowsU vectors per thread; 1, 2 and 4 measure
// within 3 % of each other (5.2 - 5.4 TB/s on 64 KB rows).
#ifndef CTC_PERM_ROWS_U
#define CTC_PERM_ROWS_U 2
#endif
constexpr int kRowsU = CTC_PERM_ROWS_U;
template <typename E>
__global__ void __launch_bounds__(kThreads) permute_rows_vec_kernel(const PermuteParams p, int chunks)
{
    const int64_t bid = blockIdx.x;
    const int64_t o = bid / chunks;
    const int64_t x0 = (bid % chunks) * (int64_t)(kThreads * kRowsU);
    const u32x4 *in = reinterpret_cast<const u32x4 *>(static_cast<const E *>(p.in) + p.in_o[o]);
    u32x4 *out = reinterpret_cast<u32x4 *>(static_cast<E *>(p.out) + p.out_o[o]);
    const int64_t nvec = p.X / (16 / (int)sizeof(E));
    u32x4 v[kRowsU];
#pragma unroll
    for (int r = 0; r < kRowsU; ++r) {
        const int64_t x = x0 + threadIdx.x + r * kThreads;
        if (x < nvec) v[r] = in[x];
    }
#pragma unroll
    for (int r = 0; r < kRowsU; ++r) {
        const int64_t x = x0 + threadIdx.x + r * kThreads;
        if (x < nvec) out[x] = v[r];
    }
}

// Rows shorter than a workgroup's kThreads * U slots: a workgroup copies `rows` whole rows, slot s of
// the workgroup being unit s % units of its row s / units (a unit is a 16-byte vector or, without
// vectors, an element).  With one row per workgroup a 224-byte row kept 14 of 256 threads busy
// (0.8 TB/s on 24 x 40 x 32 x 48 x 56 fp32 with the last mode in place; 1.5 TB/s with 448-byte rows).
template <typename E, typename UNIT, int U>
__global__ void __launch_bounds__(kThreads) permute_short_rows_kernel(const PermuteParams p, int units, int rows)
{
    const int64_t o0 = (int64_t)blockIdx.x * rows;
    UNIT v[U];
    int64_t dst[U];
#pragma unroll
    for (int r = 0; r < U; ++r) {
        const int s = threadIdx.x + r * kThreads;
        const int row = s / units, x = s - row * units;
        const int64_t o = o0 + row;
        dst[r] = -1;
        if (row < rows && o < p.O) {
            v[r] = reinterpret_cast<const UNIT *>(static_cast<const E *>(p.in) + p.in_o[o])[x];
            dst[r] = p.out_o[o] * (int64_t)sizeof(E) + (int64_t)x * (int64_t)sizeof(UNIT);
        }
    }
#pragma unroll
    for (int r = 0; r < U; ++r)
        if (dst[r] >= 0) *reinterpret_cast<UNIT *>(static_cast<char *>(p.out) + dst[r]) = v[r];
}

__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    uint64_t z = x + 0x9E3779B97F4A7C15ULL;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

__global__ void fill_f64_kernel(double *out, int64_t n, uint64_t seed)
{
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n;
         k += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t x = splitmix64(seed + (uint64_t)k);
        out[k] = 2.0 * ((double)(x >> 11) * (1.0 / 9007199254740992.0)) - 1.0;
    }
}

__global__ void fill_f32_kernel(float *out, int64_t n, uint64_t seed)
{
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n;
         k += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t x = splitmix64(seed + (uint64_t)k);
        out[k] = 2.0f * ((float)(x >> 40) * (1.0f / 16777216.0f)) - 1.0f;
    }
}

__global__ void fill_bf16_kernel(uint16_t *out, int64_t n, uint64_t seed)
{
    for (int64_t k = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; k < n;
         k += (int64_t)gridDim.x * blockDim.x) {
        const uint64_t x = splitmix64(seed + (uint64_t)k);
        const float f = 2.0f * ((float)(x >> 40) * (1.0f / 16777216.0f)) - 1.0f;
        uint32_t u = __float_as_uint(f);
        u += 0x7FFFu + ((u >> 16) & 1u);  // round to nearest even; f is finite
        out[k] = (uint16_t)(u >> 16);
    }
}

template <typename E> hipError_t launch_permute_t(const PermuteParams &p, hipStream_t stream)
{
    constexpr int V = 16 / (int)sizeof(E);
    const uintptr_t addr_bits = reinterpret_cast<uintptr_t>(p.in) | reinterpret_cast<uintptr_t>(p.out);
    if (addr_bits & (sizeof(E) - 1)) return hipErrorInvalidValue;
    const bool aligned = (addr_bits & 15) == 0;
    const bool vec = aligned && p.vec_mode == 2;
    // whole vectors at element-aligned addresses: ragged extents, or 16-byte-divisible extents behind a
    // base address that is not 16-byte aligned
    const bool ragged = !vec && p.vec_mode >= 1 && sizeof(E) >= 4;
    if (p.same_fast) {
        const int64_t units = vec ? p.X / V : p.X;
        const int per_block = kThreads * (vec ? kRowsU : 4);
        const int64_t chunks = (units + per_block - 1) / per_block;
        const int64_t blocks = chunks * p.O;
        if (blocks <= 0 || blocks > 0x7FFFFFFF) return hipErrorInvalidValue;
        if (2 * units <= per_block) {
            // several rows per workgroup
            const int rows = (int)(per_block / units);
            const int64_t nb = (p.O + rows - 1) / rows;
            if (nb > 0x7FFFFFFF) return hipErrorInvalidValue;
            if (vec)
                hipLaunchKernelGGL((permute_short_rows_kernel<E, u32x4, kRowsU>), dim3((unsigned)nb), dim3(kThreads), 0, stream, p,
                                   (int)units, rows);
            else
                hipLaunchKernelGGL((permute_short_rows_kernel<E, E, 4>), dim3((unsigned)nb), dim3(kThreads), 0, stream, p,
                                   (int)units, rows);
        } else if (vec)
            hipLaunchKernelGGL((permute_rows_vec_kernel<E>), dim3((unsigned)blocks), dim3(kThreads), 0, stream, p, (int)chunks);
        else
            hipLaunchKernelGGL((permute_rows_kernel<E>), dim3((unsigned)blocks), dim3(kThreads), 0, stream, p, (int)chunks);
    } else if (vec || ragged) {
        if (p.o_per_block < 1) return hipErrorInvalidValue;
        const int64_t no = (p.O + p.o_per_block - 1) / p.o_per_block;
        // Patch edge: 16 or 32 vectors (256 or 512 contiguous bytes a row).  The kernel is bound by the
        // access pattern alone (it runs no faster with the transposition through LDS taken out), and 512-byte
        // pieces measure 2 - 45 % faster than 256-byte ones (4096^2 fp32 transpose 5.7 -> 6.1 TB/s, a
        // batch of 112 x 80 transposes 3.7 -> 5.4: the whole matrix is one patch).  The small patch stays
        // for sides it fills better and for problems with too few large patches to occupy the chip.
        bool big = false;
        if (sizeof(E) >= 4) {
            const auto padded = [&](int64_t n, int64_t t) { return (n + t - 1) / t * t; };
            const int64_t t1 = 16 * V, t2 = 32 * V;
            const double a1 = (double)padded(p.X, t1) * (double)padded(p.Y, t1), a2 = (double)padded(p.X, t2) * (double)padded(p.Y, t2);
            const int64_t blocks2 = (padded(p.X, t2) / t2) * (padded(p.Y, t2) / t2) * no;
            big = blocks2 >= 512 && a2 <= 1.1 * a1;
        }
        // Streaming hints, as measured with 128-element patches of fp32 (TB/s for loads+stores streaming,
        // stores only, neither):
        //   128^4 reversal 5.35 / 5.20 / 5.17      odd extents 97 x 128 x 80 x 111 reversed 4.02 / 3.27 / 3.47
        //   4096^2 transpose 4.8 / 5.94 / 5.84     1000 x 3 x 1000 x 36 -> dbca 3.90 / 3.85 / 4.03
        // Loads stream when a patch row is a whole run of the input, or half of one: nothing next to it is read soon.  With
        // longer runs the neighbouring patch reads the bytes right behind, and streaming loads lose.  Stores
        // stream along with the loads, and otherwise when the pieces written are whole 128-byte lines;
        // partial lines are better left to the L2 to put together.
        PermuteParams q = p;
        const int64_t tile = (big ? 32 : 16) * V;
        q.nt_loads = (ragged || p.in_run <= 2 * tile) ? 1 : 0;  // ragged: 97 x 128 x 80 x 111 reversed 4.0 against 3.3 - 3.5
        q.nt_stores = (q.nt_loads || ((p.out_run * (int64_t)sizeof(E)) % 128 == 0 && (reinterpret_cast<uintptr_t>(p.out) & 127) == 0)) ? 1 : 0;
        const auto launch = [&](auto kernel, int T) -> hipError_t {
            const int64_t ntx = (p.X + T - 1) / T, nty = (p.Y + T - 1) / T;
            const int64_t blocks = ntx * nty * no;
            if (blocks <= 0 || blocks > 0x7FFFFFFF) return hipErrorInvalidValue;
            hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(kThreads), 0, stream, q, (int)ntx, (int)nty, (int)no);
            return hipSuccess;
        };
        const auto pick_nt = [&](auto r_tag, auto ragged_tag) -> hipError_t {
            constexpr int R = decltype(r_tag)::value;
            constexpr bool RG = decltype(ragged_tag)::value;
            switch (q.nt_loads + 2 * q.nt_stores) {
            case 0: return launch(permute_vec_kernel<E, R, RG, 0>, 16 * V * R);
            case 1: return launch(permute_vec_kernel<E, R, RG, 1>, 16 * V * R);
            case 2: return launch(permute_vec_kernel<E, R, RG, 2>, 16 * V * R);
            default: return launch(permute_vec_kernel<E, R, RG, 3>, 16 * V * R);
            }
        };
        using std::integral_constant;
        hipError_t e = hipSuccess;
        if constexpr (sizeof(E) >= 4) {
            if (big) e = vec ? pick_nt(integral_constant<int, 2>(), std::false_type()) : pick_nt(integral_constant<int, 2>(), std::true_type());
            else e = vec ? pick_nt(integral_constant<int, 1>(), std::false_type()) : pick_nt(integral_constant<int, 1>(), std::true_type());
        } else
            e = pick_nt(integral_constant<int, 1>(), std::false_type());
        if (e != hipSuccess) return e;
    } else {
        const int64_t ntx = (p.X + kTile - 1) / kTile, nty = (p.Y + kTile - 1) / kTile;
        const int64_t blocks = ntx * nty * p.O;
        if (blocks <= 0 || blocks > 0x7FFFFFFF) return hipErrorInvalidValue;
        hipLaunchKernelGGL((permute_tiled_kernel<E>), dim3((unsigned)blocks), dim3(kThreads), 0, stream, p,
                           (int)ntx, (int)nty);
    }
    return hipGetLastError();
}

}  // namespace

hipError_t launch_permute(int elem_size, const PermuteParams &p, hipStream_t stream)
{
    if (p.X <= 0 || p.Y <= 0 || p.O <= 0) return hipErrorInvalidValue;
    switch (elem_size) {
    case 2: return launch_permute_t<uint16_t>(p, stream);
    case 4: return launch_permute_t<uint32_t>(p, stream);
    case 8: return launch_permute_t<uint64_t>(p, stream);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_fill(void *dev, int64_t n, uint64_t seed, int64_t first, int dtype, hipStream_t stream)
{
    if (n <= 0) return hipSuccess;
    const int threads = 256;
    int64_t blocks = (n + threads - 1) / threads;
    if (blocks > 256 * 16) blocks = 256 * 16;
    const uint64_t s = seed + (uint64_t)first;
    switch (dtype) {
    case 0: hipLaunchKernelGGL(fill_f64_kernel, dim3((unsigned)blocks), dim3(threads), 0, stream, (double *)dev, n, s); break;
    case 1: hipLaunchKernelGGL(fill_f32_kernel, dim3((unsigned)blocks), dim3(threads), 0, stream, (float *)dev, n, s); break;
    case 2: hipLaunchKernelGGL(fill_bf16_kernel, dim3((unsigned)blocks), dim3(threads), 0, stream, (uint16_t *)dev, n, s); break;
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

}  // namespace ctc
