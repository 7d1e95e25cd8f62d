// Mode permutation kernels for gfx950.
//
// A permutation is described by three fused indices and their element-offset
// tables in the input and the output:
//   x  - runs along the input's fastest modes  (input addresses contiguous in x)
//   y  - runs along the output's fastest modes (output addresses contiguous in y)
//   o  - every remaining mode
// The tiled kernel moves a 64(x) x 64(y) patch through LDS: loads put the wave's
// lanes along x (coalesced reads), stores put them along y (coalesced writes).
// The LDS patch is padded to 65 columns so the transposed read touches each bank
// once (4-byte elements) or each bank pair once (8-byte elements).
// When input and output share their fastest mode there is nothing to transpose and
// the row-copy kernel streams contiguous runs instead.
//
// The fill kernel generates synthetic operands directly in HBM.
#include "kernels.hpp"

namespace ctc {

namespace {

constexpr int kTile = 64;
constexpr int kThreads = 256;

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

    {
        const int64_t x = x0 + lx;
        const bool xok = x < p.X;
        const int64_t xoff = p.in_x[xok ? x : 0];
#pragma unroll 4
        for (int r = 0; r < kTile / 4; ++r) {
            const int yy = ly + 4 * r;
            const int64_t y = y0 + yy;
            if (xok && y < p.Y) tile[yy][lx] = in[xoff + p.in_y[y]];
        }
    }
    __syncthreads();
    {
        const int64_t y = y0 + lx;
        const bool yok = y < p.Y;
        const int64_t yoff = p.out_y[yok ? y : 0];
#pragma unroll 4
        for (int r = 0; r < kTile / 4; ++r) {
            const int xx = ly + 4 * r;
            const int64_t x = x0 + xx;
            if (yok && x < p.X) out[yoff + p.out_x[x]] = tile[lx][xx];
        }
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
    if (p.same_fast) {
        const int64_t chunks = (p.X + kThreads * 4 - 1) / (kThreads * 4);
        const int64_t blocks = chunks * p.O;
        if (blocks <= 0 || blocks > 0x7FFFFFFF) return hipErrorInvalidValue;
        hipLaunchKernelGGL((permute_rows_kernel<E>), dim3((unsigned)blocks), dim3(kThreads), 0, stream, p,
                           (int)chunks);
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
