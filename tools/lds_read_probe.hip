// LDS read throughput of one CU by instruction and address pattern: 8 waves (one workgroup of 512 per CU) issue
// batches of 16 reads, wait, and fold the values; reports bytes per core clock per CU.
//   hipcc --offload-arch=gfx950 -O3 -o lds_read_probe tools/lds_read_probe.hip && ./lds_read_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef unsigned int u2_t __attribute__((ext_vector_type(2)));
typedef unsigned int u4_t __attribute__((ext_vector_type(4)));

constexpr int kIters = 2000, kBatch = 16;

// MODE 0: ds_read_b128 of a K-major image (64-byte rows, chunk ^ ((row >> 2) & 3))
// MODE 1: ds_read_b64_tr_b16, lines 8 g + q, 32-byte piece at (2 q + (g & 1)) ^ block
// MODE 2: the same addresses with plain ds_read_b64
// MODE 3: ds_read_b64_tr_b16, piece at (2 q + (g >> 1)) ^ block
// MODE 4: ds_read_b64_tr_b16, lane l at 8 l (512 consecutive bytes)
// MODE 5: ds_read_b64, lane l at 8 l
// MODE 7: ds_read_b128 of a K-major image (64-byte rows, chunk ^ (-(row >> 2) & 3))
// MODE 6: ds_read_b64_tr_b16, piece at (q + 4 (g & 1)) ^ block
template <int MODE> __global__ void __launch_bounds__(512, 1) probe(unsigned long long *ticks, unsigned *sink)
{
    __shared__ __attribute__((aligned(1024))) unsigned char smem[65536];
    for (int i = threadIdx.x; i < 16384; i += 512) reinterpret_cast<unsigned *>(smem)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3, frow = lane & 15;
    unsigned base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem + (wave & 1) * 32768;
    unsigned addr;
    if (MODE == 0) addr = base + frow * 64 + ((g ^ ((frow >> 2) & 3)) << 4);
    else if (MODE == 7) addr = base + frow * 64 + ((g ^ ((0 - (frow >> 2)) & 3)) << 4);
    else if (MODE == 1 || MODE == 2) addr = base + 512 * (8 * g + q) + 8 * pp + 32 * (2 * q + (g & 1));
    else if (MODE == 3) addr = base + 512 * (8 * g + q) + 8 * pp + 32 * (2 * q + (g >> 1));
    else if (MODE == 6) addr = base + 512 * (8 * g + q) + 8 * pp + 32 * (q + 4 * (g & 1));
    else addr = base + 8 * lane;
    unsigned acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < kIters; ++it) {
        if (MODE == 0 || MODE == 7) {
            u4_t v[kBatch];
#pragma unroll
            for (int b = 0; b < kBatch; ++b) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v[b]) : "v"(addr), "n"(0), "n"(b * 1024));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int b = 0; b < kBatch; ++b) {
                asm volatile("" : "+v"(v[b]));
                acc ^= v[b][0];
            }
        } else {
            u2_t v[kBatch];
#pragma unroll
            for (int b = 0; b < kBatch; ++b) {
                // block b of the line: XOR on the piece index for the swizzled modes, next 512 bytes otherwise
                const unsigned a = (MODE >= 4) ? addr + 512 * b : ((addr ^ (32 * (b & 7))) + 2048 * (b >> 3));
                if (MODE == 2 || MODE == 5) asm volatile("ds_read_b64 %0, %1" : "=v"(v[b]) : "v"(a));
                else asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v[b]) : "v"(a));
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int b = 0; b < kBatch; ++b) {
                asm volatile("" : "+v"(v[b]));
                acc ^= v[b][0];
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 512 + threadIdx.x] = acc;
}

template <int MODE> void run(const char *what, int bytes_per_lane)
{
    const int blocks = 256;
    unsigned long long *ticks;
    unsigned *sink;
    (void)hipMalloc(&ticks, blocks * sizeof(*ticks));
    (void)hipMalloc(&sink, blocks * 512 * sizeof(unsigned));
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(512), 0, 0, ticks, sink);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks);
    (void)hipMemcpy(h.data(), ticks, blocks * sizeof(*ticks), hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto t : h) mean += (double)t;
    mean /= blocks;
    const double bytes = (double)kIters * kBatch * 8 /*waves*/ * 64 * bytes_per_lane;
    printf("%-64s %7.1f bytes / clock / CU   (%.1f clocks per wave instruction)\n", what, bytes / mean, mean / ((double)kIters * kBatch * 8));
    (void)hipFree(ticks);
    (void)hipFree(sink);
}

int main()
{
    run<0>("ds_read_b128, K-major image, chunk ^ ((row >> 2) & 3)", 16);
    run<7>("ds_read_b128, K-major image, chunk ^ (-(row >> 2) & 3)", 16);
    run<1>("ds_read_b64_tr_b16, piece (2q + (g&1)) ^ block", 8);
    run<2>("ds_read_b64,        piece (2q + (g&1)) ^ block", 8);
    run<3>("ds_read_b64_tr_b16, piece (2q + (g>>1)) ^ block", 8);
    run<6>("ds_read_b64_tr_b16, piece (q + 4(g&1)) ^ block", 8);
    run<4>("ds_read_b64_tr_b16, lane l at 8 l", 8);
    run<5>("ds_read_b64,        lane l at 8 l", 8);
    return 0;
}
