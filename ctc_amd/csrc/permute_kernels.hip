// Mode permutation kernels for gfx950.
//
// A permutation is described by three fused indices and their element-offset
// tables in the input and the output:
//   x  - runs along the input's fastest modes  (input addresses contiguous in x)
//   y  - runs along the output's fastest modes (output addresses contiguous in y)
//   o  - every remaining mode
//
// Vector kernel (the fast path; 16-byte accesses on both sides).  V = 16 / element
// size.  A workgroup moves a (16V x 16V) patch of (x, y): thread (vx, vq) loads the V
// rows y = V*vq .. V*vq+V-1 as 16-byte vectors at x = V*vx, so 16 neighbouring lanes
// read 256 contiguous bytes; it transposes its V x V micro-tile in registers and writes
// V vectors into an LDS image indexed [x][y] whose 16-byte chunks are XOR-swizzled with
// the row's vx, which makes both the writes (8 lanes: 8 different chunks) and the
// reads (16 lanes of one row: 16 different chunks) bank-conflict free without padding.
// In the store phase 16 neighbouring lanes write 256 contiguous bytes of the output.
// A workgroup walks o_per_block consecutive `o` values with the same (x, y) patch (the host
// picks 4 when that lengthens the contiguous runs of one side, else 1).
//
// Scalar kernel: the same patch scheme with element accesses and a padded LDS tile,
// for extents or strides that do not allow 16-byte vectors.
// Row kernels: input and output share their fastest mode, nothing to transpose.
//
// The fill kernel generates synthetic operands directly in HBM.
#include "kernels.hpp"

namespace ctc {

namespace {

constexpr int kTile = 64;
constexpr int kThreads = 256;
// bit 0: non-temporal loads, bit 1: non-temporal stores.  The output of a permutation is not
// read again by the kernel, and streaming stores gain 4 % on the 128^4 reversal (4.98 -> 5.18 TB/s);
// non-temporal loads change nothing.
#ifndef CTC_PERM_NT
#define CTC_PERM_NT 2
#endif

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <typename E>
__global__ void __launch_bounds__(kThreads) permute_vec_kernel(const PermuteParams p, int ntx, int nty, int no)
{
    constexpr int V = 16 / sizeof(E);
    constexpr int T = 16 * V;  // patch edge in elements; one LDS row = T * sizeof(E) = 256 bytes
    __shared__ __attribute__((aligned(16))) E tile[T * T];

    const int64_t bid = blockIdx.x;
    const int tx = (int)(bid % ntx);
    const int ty = (int)((bid / ntx) % nty);
    const int64_t ob = bid / ((int64_t)ntx * nty);
    const int64_t x0 = (int64_t)tx * T, y0 = (int64_t)ty * T;
    const int lo = threadIdx.x & 15;  // vector index along the contiguous direction
    const int hi = threadIdx.x >> 4;  // 0..15

    // load side: x = x0 + V*lo, rows y = y0 + V*hi + r
    const int64_t lx = x0 + (int64_t)V * lo;
    const bool lx_ok = lx < p.X;
    const int64_t in_xo = p.in_x[lx_ok ? lx : 0];
    int64_t in_yo[V];
    bool ly_ok[V];
#pragma unroll
    for (int r = 0; r < V; ++r) {
        const int64_t y = y0 + (int64_t)V * hi + r;
        ly_ok[r] = lx_ok && y < p.Y;
        in_yo[r] = p.in_y[y < p.Y ? y : 0];
    }
    // store side: y = y0 + V*lo, rows x = x0 + hi + 16*r
    const int64_t sy = y0 + (int64_t)V * lo;
    const bool sy_ok = sy < p.Y;
    const int64_t out_yo = p.out_y[sy_ok ? sy : 0];
    int64_t out_xo[V];
    bool sx_ok[V];
#pragma unroll
    for (int r = 0; r < V; ++r) {
        const int64_t x = x0 + hi + 16 * r;
        sx_ok[r] = sy_ok && x < p.X;
        out_xo[r] = p.out_x[x < p.X ? x : 0];
    }

    const int64_t o_begin = ob * p.o_per_block;
    const int64_t o_end = o_begin + p.o_per_block < p.O ? o_begin + p.o_per_block : p.O;
    for (int64_t o = o_begin; o < o_end; ++o) {
        const E *in = static_cast<const E *>(p.in) + p.in_o[o] + in_xo;
        E *out = static_cast<E *>(p.out) + p.out_o[o] + out_yo;

        union Vec { u32x4 v; E e[V]; };
        Vec rows[V];
#pragma unroll
        for (int r = 0; r < V; ++r)
            if (ly_ok[r]) {
                if (CTC_PERM_NT & 1) rows[r].v = __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(in + in_yo[r]));
                else rows[r].v = *reinterpret_cast<const u32x4 *>(in + in_yo[r]);
            }
        if (o != o_begin) __syncthreads();  // previous patch fully read back
        // micro-tile transpose: column c of the V loaded rows becomes the vector for x = V*lo + c
#pragma unroll
        for (int c = 0; c < V; ++c) {
            Vec col;
#pragma unroll
            for (int r = 0; r < V; ++r) col.e[r] = rows[r].e[c];
            const int xr = V * lo + c;            // LDS row
            const int chunk = hi ^ lo;            // y chunk V*hi.., swizzled with the row's vx
            *reinterpret_cast<u32x4 *>(&tile[xr * T + chunk * V]) = col.v;
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < V; ++r) {
            const int xr = hi + 16 * r;
            const int chunk = lo ^ (xr / V);
            const u32x4 v = *reinterpret_cast<const u32x4 *>(&tile[xr * T + chunk * V]);
            if (sx_ok[r]) {
                if (CTC_PERM_NT & 2) __builtin_nontemporal_store(v, reinterpret_cast<u32x4 *>(out + out_xo[r]));
                else *reinterpret_cast<u32x4 *>(out + out_xo[r]) = v;
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
    const bool aligned = ((reinterpret_cast<uintptr_t>(p.in) | reinterpret_cast<uintptr_t>(p.out)) & 15) == 0;
    const bool vec = aligned && p.vec_ok[sizeof(E) == 2 ? 0 : sizeof(E) == 4 ? 1 : 2];
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
    } else if (vec) {
        constexpr int T = 16 * V;
        const int64_t ntx = (p.X + T - 1) / T, nty = (p.Y + T - 1) / T;
        if (p.o_per_block < 1) return hipErrorInvalidValue;
        const int64_t no = (p.O + p.o_per_block - 1) / p.o_per_block;
        const int64_t blocks = ntx * nty * no;
        if (blocks <= 0 || blocks > 0x7FFFFFFF) return hipErrorInvalidValue;
        hipLaunchKernelGGL((permute_vec_kernel<E>), dim3((unsigned)blocks), dim3(kThreads), 0, stream, p, (int)ntx,
                           (int)nty, (int)no);
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
