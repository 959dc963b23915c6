// Do back-to-back MFMAs into the SAME accumulator stall?  (development aid, not part of libdwcgan_hip.so)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 benchmarks/mfma_chain_probe.hip -o benchmarks/bin/mfma_chain_probe
// One workgroup per CU, 1 or 2 waves per SIMD, v_mfma_f32_32x32x16_bf16 on zero operands (the clock stays at 2.4 GHz, see
// mfma_bf16_ceiling.hip), 24 MFMAs per iteration over 4 accumulators in runs of RUN consecutive instructions on one accumulator
// (RUN = 1: the accumulators rotate; RUN = 6: the weight-gradient kernels' source order).  Prints shader cycles per MFMA.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int RUN>
__global__ __launch_bounds__(512) void chain(const bf16x8* __restrict__ in, float* __restrict__ out, unsigned long long* cyc, int iters) {
    bf16x8 a[6], b[6];
    for (int i = 0; i < 6; ++i) {
        a[i] = in[(threadIdx.x + 512 * i) & 4095];
        b[i] = in[(threadIdx.x + 512 * i + 2048) & 4095];
    }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 24 / RUN; ++g)
#pragma unroll
            for (int u = 0; u < RUN; ++u) {
                const int k = g * RUN + u;
                acc[g & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[k % 6], b[(k / 6 + k) % 6], acc[g & 3], 0, 0, 0);
            }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    const int CUS = 256, iters = 4000;
    bf16x8* din;
    float* dout;
    unsigned long long* dc;
    CK(hipMalloc(&din, 4096 * 16));
    CK(hipMemset(din, 0, 4096 * 16));
    CK(hipMalloc(&dout, CUS * 512 * 4));
    CK(hipMalloc(&dc, CUS * 8));
    for (int threads : {256, 512})
        for (int run : {1, 2, 3, 6}) {
            for (int rep = 0; rep < 3; ++rep) {
                if (run == 1) hipLaunchKernelGGL(chain<1>, dim3(CUS), dim3(threads), 0, 0, din, dout, dc, iters);
                else if (run == 2) hipLaunchKernelGGL(chain<2>, dim3(CUS), dim3(threads), 0, 0, din, dout, dc, iters);
                else if (run == 3) hipLaunchKernelGGL(chain<3>, dim3(CUS), dim3(threads), 0, 0, din, dout, dc, iters);
                else hipLaunchKernelGGL(chain<6>, dim3(CUS), dim3(threads), 0, 0, din, dout, dc, iters);
                CK(hipDeviceSynchronize());
            }
            std::vector<unsigned long long> c(CUS);
            CK(hipMemcpy(c.data(), dc, CUS * 8, hipMemcpyDeviceToHost));
            std::sort(c.begin(), c.end());
            // s_memtime counts at 100 MHz x (shader clock / 100 MHz)?  it is the shader clock on this part (mfma_bf16_ceiling.hip)
            printf("  %d wave(s)/SIMD  runs of %d on one accumulator: %.2f shader cycles per MFMA and SIMD\n", threads / 256, run,
                   (double)c[CUS / 2] / ((double)iters * 24 * (threads / 256)));
        }
    return 0;
}
