// What the chip sustains on bare bf16 MFMAs (development aid, not part of libdwcgan_hip.so).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 benchmarks/mfma_bf16_ceiling.hip -o benchmarks/bin/mfma_bf16_ceiling
// MI355X lowers its shader clock under dense matrix work (MI355X_MICROARCH.md, "DVFS give-back"): the 2.5 PFLOP/s roof is
// 256 CUs x 4 SIMDs x 1024 MAC/cycle at 2.4 GHz, and a loop of back-to-back MFMAs on RANDOM operands does not hold 2.4 GHz.
// This prints, for both bf16 MFMA shapes, operands in registers (no LDS, no memory traffic in the loop), one and two waves
// per SIMD, random and all-zero operands, launches of about 0.2 / 1 / 3 ms: TFLOP/s by wall clock, fraction of 2.5 PF, and
// the clock the kernel saw (delta s_memtime / delta s_memrealtime x 100 MHz, median over workgroups).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int SHAPE>      // 16: 16x16x32, 32: 32x32x16
__global__ __launch_bounds__(512) void mfma_loop(const bf16x8* __restrict__ in, float* __restrict__ out, unsigned long long* stamps, int iters) {
    // 8 distinct operand pairs and 8 (16x16) / 4 (32x32) independent accumulators per wave
    bf16x8 a[8], b[8];
    for (int i = 0; i < 8; ++i) {
        a[i] = in[(threadIdx.x + 512 * i) & 4095];
        b[i] = in[(threadIdx.x + 512 * i + 2048) & 4095];
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    if constexpr (SHAPE == 16) {
        f32x4 acc[8];
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i], b[(i + u) & 7], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        f32x16 acc[4];
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i + (u & 1) * 4], b[(i + u) & 7], acc[i], 0, 0, 0);
        }
        for (int i = 0; i < 4; ++i)
            for (int r = 0; r < 16; ++r) s += acc[i][r];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) {
        stamps[2 * blockIdx.x] = t1 - t0;
        stamps[2 * blockIdx.x + 1] = r1 - r0;
    }
}

int main() {
    const int CUS = 256;
    std::vector<unsigned short> h(4096 * 8);
    bf16x8* din[2];
    float* dout;
    unsigned long long* dst;
    CK(hipMalloc(&din[0], 4096 * 16));
    CK(hipMalloc(&din[1], 4096 * 16));
    CK(hipMalloc(&dout, 2 * CUS * 512 * 4));
    CK(hipMalloc(&dst, 2 * CUS * 16));
    srand(1);
    for (auto& v : h) {      // random bf16 in [-1, 1)
        const float f = (float)rand() / 2147483648.0f * 2.f - 1.f;
        unsigned u;
        memcpy(&u, &f, 4);
        v = (unsigned short)(u >> 16);
    }
    CK(hipMemcpy(din[0], h.data(), 4096 * 16, hipMemcpyHostToDevice));
    CK(hipMemset(din[1], 0, 4096 * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("bare bf16 MFMA loop, operands in registers, 256 workgroups (one per CU)\n");
    for (int data = 0; data < 2; ++data)
        for (int shape : {16, 32})
            for (int threads : {256, 512})
                for (int iters : {1500, 8000, 24000}) {
                    // per iteration and wave: 32 x 16x16x32 (16 cycles each) or 16 x 32x32x16 (32 cycles): 512 MFMA cycles
                    const double flop = 2.0 * 16 * 16 * 32 * 32 * (double)iters * (threads / 64) * CUS;
                    std::vector<float> ms;
                    for (int rep = 0; rep < 12; ++rep) {
                        CK(hipEventRecord(e0, 0));
                        if (shape == 16) hipLaunchKernelGGL(mfma_loop<16>, dim3(CUS), dim3(threads), 0, 0, din[data], dout, dst, iters);
                        else hipLaunchKernelGGL(mfma_loop<32>, dim3(CUS), dim3(threads), 0, 0, din[data], dout, dst, iters);
                        CK(hipEventRecord(e1, 0));
                        CK(hipEventSynchronize(e1));
                        float t;
                        CK(hipEventElapsedTime(&t, e0, e1));
                        if (rep >= 4) ms.push_back(t);
                    }
                    std::sort(ms.begin(), ms.end());
                    std::vector<unsigned long long> st(2 * CUS);
                    CK(hipMemcpy(st.data(), dst, 2 * CUS * 8, hipMemcpyDeviceToHost));
                    std::vector<double> clk, cyc;
                    for (int b = 0; b < CUS; ++b) {
                        clk.push_back((double)st[2 * b] / (double)st[2 * b + 1] * 100.0);
                        cyc.push_back((double)st[2 * b]);
                    }
                    std::sort(clk.begin(), clk.end());
                    std::sort(cyc.begin(), cyc.end());
                    const double t = ms[ms.size() / 2] * 1e-3;
                    const double ideal = 512.0 * iters * (threads / 256);       // MFMA issue cycles per SIMD
                    printf("  %-6s %s  %d wave(s)/SIMD  %7.3f ms: %7.1f TFLOP/s = %.3f of 2.5 PF | in-kernel clock %4.0f MHz | issue %.3f of the loop's cycles\n",
                           data ? "zeros" : "random", shape == 16 ? "16x16x32" : "32x32x16", threads / 256, t * 1e3, flop / t * 1e-12, flop / t / 2.5e15,
                           clk[CUS / 2], ideal / cyc[CUS / 2]);
                }
    return 0;
}
