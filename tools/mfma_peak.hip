// Microbenchmark: sustained rate of v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32 issued
// back to back from registers (no memory traffic), at 1, 2 and 4 waves per SIMD, with the
// in-kernel shader clock (s_memtime ticks per s_memrealtime 100 MHz tick) reported beside it.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o tools/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC, bool ZERO>
__global__ void __launch_bounds__(256) mfma_f64(double *out, unsigned long long *stamps, int iters, double a0, double b0)
{
    d4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = d4{0, 0, 0, 0};
    double a = ZERO ? 0.0 : a0 + 0.001 * threadIdx.x, b = ZERO ? 0.0 : b0 - 0.003 * threadIdx.x;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int NACC>
__global__ void __launch_bounds__(256) mfma_f32(float *out, int iters, float a0, float b0)
{
    f4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
    float a = a0 + threadIdx.x, b = b0 - threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F> double time_ms(F f)
{
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    f();
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    f();
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int NACC, bool ZERO> void run(double *out, unsigned long long *st, int blocks, int iters, const char *tag)
{
    double ms = time_ms([&] { hipLaunchKernelGGL((mfma_f64<NACC, ZERO>), dim3(blocks), dim3(256), 0, 0, out, st, iters, 1.0, 2.0); });
    std::vector<unsigned long long> h(2 * blocks);
    (void)hipMemcpy(h.data(), st, 16 * blocks, hipMemcpyDeviceToHost);
    std::vector<double> cyc, clk;
    for (int i = 0; i < blocks; ++i) { cyc.push_back((double)h[2 * i]); clk.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 100.0); }
    std::sort(cyc.begin(), cyc.end()); std::sort(clk.begin(), clk.end());
    double fl = (double)blocks * 4 * iters * NACC * 2048.0;
    int wps = blocks / 256;
    printf("f64 %-5s %2d acc %d waves/SIMD: %7.2f ms %6.2f TFLOP/s  cycles/MFMA/SIMD %.1f  clock %.0f MHz\n", tag, NACC, wps, ms,
           fl / ms * 1e-9, cyc[blocks / 2] / ((double)iters * NACC * wps), clk[blocks / 2]);
}

int main()
{
    double *out; (void)hipMalloc(&out, 256 * 16 * 256 * 8);
    unsigned long long *st; (void)hipMalloc(&st, 16 * 4096);
    const int iters = 20000;
    for (int bpc = 1; bpc <= 4; bpc *= 2) {
        int blocks = 256 * bpc;
        run<16, false>(out, st, blocks, iters, "rand");
        run<16, true>(out, st, blocks, iters, "zero");
        run<4, false>(out, st, blocks, iters * 4, "rand");
        run<1, false>(out, st, blocks, iters * 16, "rand");
        double ms = time_ms([&] { hipLaunchKernelGGL(mfma_f32<16>, dim3(blocks), dim3(256), 0, 0, (float *)out, iters, 1.0f, 2.0f); });
        printf("f32 16x16x4  16 acc  %d waves/SIMD: %.2f ms  %.2f TFLOP/s\n", bpc, ms, (double)blocks * 4 * iters * 16 * 2048.0 / ms * 1e-9);
    }
    return 0;
}
