// Calibration microbenchmark: sustained fp32 MFMA rate on gfx950 as a function of the number of
// independent accumulators per wave, waves per SIMD and MFMA shape.  Build and run:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_peak benchmarks/mfma_peak.hip && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-6f, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / NACC; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i)
        for (int r = 0; r < 4; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F>
double run(F launch, double flop_per_mfma, int blocks, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    launch(blocks, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch(blocks, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)blocks * 4 * iters * 16;
    return mfmas * flop_per_mfma / (ms * 1e-3) / 1e12;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 256 * 8 * sizeof(float) * 4);
    const int iters = 20000;
    for (int wpb = 1; wpb <= 4; wpb *= 2) {  // blocks per CU (one 4-wave block = one wave per SIMD)
        const int blocks = 256 * wpb;
        printf("waves/SIMD %d\n", wpb);
#define R32(N) printf("  32x32x2  nacc %d : %7.1f TF\n", N, run([&](int b, int it) { hipLaunchKernelGGL(k32<N>, dim3(b), dim3(256), 0, 0, out, it, 1.0f, 0.5f); }, 4096.0, blocks, iters));
#define R16(N) printf("  16x16x4  nacc %d : %7.1f TF\n", N, run([&](int b, int it) { hipLaunchKernelGGL(k16<N>, dim3(b), dim3(256), 0, 0, out, it, 1.0f, 0.5f); }, 2048.0, blocks, iters));
        R32(1) R32(2) R32(4) R16(1) R16(2) R16(4) R16(8)
    }
    return 0;
}
