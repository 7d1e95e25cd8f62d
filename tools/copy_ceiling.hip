// Device-to-device copy rate of hand-written streaming kernels on the current GPU, in several
// shapes (vectors in flight per thread, grid size, non-temporal hints): the ceiling a permutation
// kernel can be compared with.  Build: hipcc --offload-arch=gfx950 -O3 tools/copy_ceiling.hip -o tools/copy_ceiling
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                     \
    do {                                                                             \
        hipError_t e_ = (x);                                                         \
        if (e_ != hipSuccess) {                                                      \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                  \
            exit(1);                                                                 \
        }                                                                            \
    } while (0)

// Each workgroup copies runs of U * 256 vectors; a thread holds U vectors in flight.
template <int U, bool NT_LOAD, bool NT_STORE>
__global__ void __launch_bounds__(256) copy_kernel(const u4 *__restrict__ in, u4 *__restrict__ out, long long nvec)
{
    const long long chunk = 256LL * U;
    for (long long base = blockIdx.x * chunk; base < nvec; base += gridDim.x * chunk) {
        u4 v[U];
#pragma unroll
        for (int r = 0; r < U; ++r) {
            const long long i = base + threadIdx.x + 256LL * r;
            if (i < nvec) v[r] = NT_LOAD ? __builtin_nontemporal_load(in + i) : in[i];
        }
#pragma unroll
        for (int r = 0; r < U; ++r) {
            const long long i = base + threadIdx.x + 256LL * r;
            if (i < nvec) {
                if (NT_STORE) __builtin_nontemporal_store(v[r], out + i);
                else out[i] = v[r];
            }
        }
    }
}

template <int U, bool NL, bool NS> void run(const char *name, const u4 *in, u4 *out, long long nvec, int blocks)
{
    const long long chunk = 256LL * U;
    long long exact = (nvec + chunk - 1) / chunk;
    unsigned grid = (unsigned)(blocks > 0 && blocks < exact ? blocks : exact);
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((copy_kernel<U, NL, NS>), dim3(grid), dim3(256), 0, 0, in, out, nvec);
    CHECK(hipEventRecord(e0, 0));
    const int iters = 10;
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((copy_kernel<U, NL, NS>), dim3(grid), dim3(256), 0, 0, in, out, nvec);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    ms /= iters;
    printf("%-28s U=%d grid=%8u  %.3f ms  %.2f TB/s\n", name, U, grid, ms, 2.0 * nvec * 16 / ms * 1e-9);
}

int main(int argc, char **argv)
{
    const long long gib = argc > 1 ? atoll(argv[1]) : 1;
    const long long bytes = gib << 30, nvec = bytes / 16;
    u4 *in, *out;
    CHECK(hipMalloc(&in, bytes));
    CHECK(hipMalloc(&out, bytes));
    CHECK(hipMemset(in, 1, bytes));
    CHECK(hipMemset(out, 2, bytes));
    {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        for (int i = 0; i < 2; ++i) CHECK(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0));
        CHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < 10; ++i) CHECK(hipMemcpyAsync(out, in, bytes, hipMemcpyDeviceToDevice, 0));
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-28s %.3f ms  %.2f TB/s\n", "hipMemcpyDtoD", ms / 10, 2.0 * bytes / (ms / 10) * 1e-9);
    }
    printf("%lld GiB in, %lld GiB out\n", gib, gib);
    const int grids[] = {0, 2048, 4096, 8192};
    for (int g : grids) {
        run<1, false, false>("plain", in, out, nvec, g);
        run<2, false, false>("plain", in, out, nvec, g);
        run<4, false, false>("plain", in, out, nvec, g);
        run<8, false, false>("plain", in, out, nvec, g);
        run<4, false, true>("nt store", in, out, nvec, g);
        run<4, true, true>("nt load + nt store", in, out, nvec, g);
        run<8, true, true>("nt load + nt store", in, out, nvec, g);
    }
    return 0;
}
