// Can the chip run fp64 MFMAs at their full rate while it streams stores to HBM?  Times
//   M   register-only v_mfma_f64_16x16x4_f64, one wave per SIMD on every CU;
//   W   a write-only streaming kernel;
//   M|W both at once on two streams;
//   F   one kernel whose waves issue one 1 KB store per SPM MFMAs (the store/MFMA ratio of a
//       128 x 128 fp64 C tile with K = 64 is one store per 8 MFMAs).
// Build: hipcc --offload-arch=gfx950 -O3 tools/overlap_probe.hip -o tools/overlap_probe
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#define CHECK(x)                                                        \
    do {                                                                \
        hipError_t e_ = (x);                                            \
        if (e_ != hipSuccess) {                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));     \
            exit(1);                                                    \
        }                                                               \
    } while (0)

__global__ void __launch_bounds__(256) mfma_only(double *out, int iters, double a0, double b0)
{
    d4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = d4{0, 0, 0, 0};
    const double a = a0 + 0.001 * threadIdx.x, b = b0 - 0.003 * threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void __launch_bounds__(256) write_only(d2 *out, long long nvec, double v)
{
    const long long i = blockIdx.x * 256LL + threadIdx.x;
    if (i < nvec) out[i] = d2{v + (double)threadIdx.x, v};
}

// SPM MFMAs between two stores; every wave owns a contiguous region of `out` and walks it.
// clk: core-clock ticks and 100 MHz ticks workgroup 0 ran for (the shader clock under this load)
template <int SPM>
__global__ void __launch_bounds__(256) fused(d2 *out, long long vec_per_wave, int iters, double a0, double b0, unsigned long long *clk)
{
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    d4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = d4{0, 0, 0, 0};
    const double a = a0 + 0.001 * threadIdx.x, b = b0 - 0.003 * threadIdx.x;
    const long long wave = blockIdx.x * 4LL + (threadIdx.x >> 6);
    d2 *dst = out + wave * vec_per_wave + (threadIdx.x & 63);
    long long pos = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            if (i % SPM == SPM - 1) {
                dst[pos] = d2{a, b};
                pos += 64;
                if (pos >= vec_per_wave) pos = 0;
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    double s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 1.2345e300) out[0] = d2{s, s};
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = __builtin_readcyclecounter() - c0;
        clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

// Bursts: MPB MFMAs, then MPB / 8 stores back to back (the same one-store-per-8-MFMAs ratio).  With
// rounds == 1 and one workgroup per C tile this is the shape of a GEMM kernel with a store epilogue.
__global__ void __launch_bounds__(256) bursty(d2 *out, long long vec_per_wave, long long regions, int rounds, int mpb, double a0, double b0)
{
    d4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = d4{0, 0, 0, 0};
    const double a = a0 + 0.001 * threadIdx.x, b = b0 - 0.003 * threadIdx.x;
    const long long wave = blockIdx.x * 4LL + (threadIdx.x >> 6);
    d2 *dst = out + ((wave % regions) * vec_per_wave) + (threadIdx.x & 63);
    long long pos = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int it = 0; it < mpb / 16; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        for (int st = 0; st < mpb / 8; ++st) {
            dst[pos] = d2{acc[st & 15][0], acc[st & 15][1]};
            pos += 64;
            if (pos >= vec_per_wave) pos = 0;
        }
    }
}

static double now_ms()
{
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main()
{
    const long long wbytes = 8LL << 30, nvec = wbytes / 16;
    const int iters = 6000;  // 96,000 MFMAs per wave: about 2.6 ms
    double *mout;
    d2 *wout;
    CHECK(hipMalloc(&mout, 1024 * 256 * 8));
    CHECK(hipMalloc(&wout, wbytes));
    unsigned long long *clk;
    CHECK(hipMalloc(&clk, 16));
    hipStream_t s1, s2;
    CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
    const double mflop = 256.0 * 4 * iters * 16 * 2048.0;
    auto launch_m = [&](hipStream_t s) { hipLaunchKernelGGL(mfma_only, dim3(256), dim3(256), 0, s, mout, iters, 1.0, 2.0); };
    auto launch_w = [&](hipStream_t s) { hipLaunchKernelGGL(write_only, dim3((unsigned)(nvec / 256)), dim3(256), 0, s, wout, nvec, 3.0); };
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipDeviceSynchronize());
        double t0 = now_ms();
        launch_m(s1);
        CHECK(hipDeviceSynchronize());
        double tm = now_ms() - t0;
        t0 = now_ms();
        launch_w(s2);
        CHECK(hipDeviceSynchronize());
        double tw = now_ms() - t0;
        t0 = now_ms();
        launch_m(s1);
        launch_w(s2);
        launch_w(s2);
        CHECK(hipDeviceSynchronize());
        double tb = now_ms() - t0;
        printf("M alone %.3f ms (%.1f TFLOP/s) | W alone %.3f ms (%.2f TB/s) | M with 2 W %.3f ms (sum %.3f, max %.3f)\n", tm,
               mflop / tm * 1e-9, tw, wbytes / tw * 1e-9, tb, tm + 2 * tw, tm > 2 * tw ? tm : 2 * tw);
    }
    // fused: 256 blocks (one wave per SIMD) or 512 (two)
    for (int blocks = 256; blocks <= 512; blocks *= 2) {
        const long long vec_per_wave = nvec / (blocks * 4LL);
        auto run = [&](auto kernel, int spm) {
            for (int w = 0; w < 4; ++w) hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, wout, vec_per_wave, iters, 1.0, 2.0, clk);
            CHECK(hipDeviceSynchronize());
            double t0 = now_ms();
            hipLaunchKernelGGL(kernel, dim3(blocks), dim3(256), 0, 0, wout, vec_per_wave, iters, 1.0, 2.0, clk);
            CHECK(hipDeviceSynchronize());
            double t = now_ms() - t0;
            unsigned long long h[2];
            CHECK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
            const double fl = blocks * 4.0 * iters * 16 * 2048.0;
            const double by = blocks * 4.0 * iters * (16 / spm) * 1024.0;
            // MFMA cycles a SIMD needs for its waves' MFMAs (64 each) over the core cycles the workgroup took
            const double duty = (blocks / 256) * (double)iters * 16 * 64 / (double)h[0];
            printf("fused %d blocks, 1 store / %2d MFMA: %.3f ms  %.1f TFLOP/s  %.2f TB/s written | shader clock %.2f GHz, MFMA cycles / core cycles %.3f\n",
                   blocks, spm, t, fl / t * 1e-9, by / t * 1e-9, (double)h[0] / (10.0 * (double)h[1]), duty);
        };
        run(fused<16>, 16);
        run(fused<8>, 8);
        run(fused<4>, 4);
    }
    // bursts of 256 MFMAs + 32 stores per wave: persistent waves, then one workgroup per burst
    {
        const int total_rounds = 6000 * 16 / 256;  // same MFMA count per SIMD slot as above
        for (int blocks = 256; blocks <= 512; blocks *= 2) {
            const long long vec_per_wave = nvec / (blocks * 4LL);
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipDeviceSynchronize());
                double t0 = now_ms();
                hipLaunchKernelGGL(bursty, dim3(blocks), dim3(256), 0, 0, wout, vec_per_wave, nvec / vec_per_wave, total_rounds, 256, 1.0, 2.0);
                CHECK(hipDeviceSynchronize());
                double t = now_ms() - t0;
                if (rep)
                    printf("bursty persistent %d blocks: %.3f ms  %.1f TFLOP/s  %.2f TB/s written\n", blocks, t,
                           blocks * 4.0 * total_rounds * 256 * 2048.0 / t * 1e-9, blocks * 4.0 * total_rounds * 32 * 1024.0 / t * 1e-9);
            }
        }
        for (int per_cu = 1; per_cu <= 2; ++per_cu) {
            const int blocks = 256 * per_cu * total_rounds;  // one burst per workgroup
            const long long vec_per_wave = 32 * 64;
            for (int rep = 0; rep < 2; ++rep) {
                CHECK(hipDeviceSynchronize());
                double t0 = now_ms();
                hipLaunchKernelGGL(bursty, dim3(blocks), dim3(256), 0, 0, wout, vec_per_wave, nvec / vec_per_wave, 1, 256, 1.0, 2.0);
                CHECK(hipDeviceSynchronize());
                double t = now_ms() - t0;
                if (rep)
                    printf("one burst per workgroup, %d workgroups: %.3f ms  %.1f TFLOP/s  %.2f TB/s written\n", blocks, t,
                           blocks * 4.0 * 256 * 2048.0 / t * 1e-9, blocks * 4.0 * 32 * 1024.0 / t * 1e-9);
            }
        }
    }
    return 0;
}
