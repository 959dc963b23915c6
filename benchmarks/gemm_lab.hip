// GEMM laboratory (development aid): the K-loop structure of conv_gemm_kernel on a plain dense
// product C[M][N] = A[M][K] * B[N][K]^T, with switches to ablate parts of it.  Used to find out
// what separates the conv kernel from the fp32 MFMA roof without the conv addressing in the way.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o benchmarks/bin/gemm_lab benchmarks/gemm_lab.hip
//   benchmarks/bin/gemm_lab [M N K]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 32;
enum { NO_GLOBAL = 1, NO_BARRIER = 2, PIN = 4, NO_LDS_READ = 8, SAME_A = 16, SPREAD = 32 };

__device__ __forceinline__ void glds16(const float* src, float* lds) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)lds, 16, 0, 0);
}

// shader-clock probe: (s_memtime delta, s_memrealtime delta @100 MHz) of workgroup 0 -> average MHz under this load
__device__ long long g_probe[2];
struct ClockProbe {
    long long c0, r0;
    __device__ ClockProbe() : c0(clock64()), r0(wall_clock64()) {}
    __device__ void done() {
        if (blockIdx.x == 0 && threadIdx.x == 0) {
            g_probe[0] = clock64() - c0;
            g_probe[1] = wall_clock64() - r0;
        }
    }
};

template <int BM, int BN, int WM, int WN, int TM, int TN, int FLAGS>
__global__ __launch_bounds__(256) void gemm_v0(const float* __restrict__ A, const float* __restrict__ Bm, float* __restrict__ C,
                                               int M, int N, int K, int tiles_n) {
    constexpr int A_PASSES = BM / 32, B_PASSES = BN / 32;
    constexpr int A_TILE = BM * BK, B_TILE = BN * BK;
    ClockProbe probe;
    __shared__ __attribute__((aligned(16))) float smem[2 * (A_TILE + B_TILE)];
    float* sA = smem;
    float* sB = smem + 2 * A_TILE;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    int bid = blockIdx.x;
    {
        const int nb = gridDim.x;
        const int q = nb >> 3, r = nb & 7, x = bid & 7, y = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    }
    const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int arow = t >> 3;
    const int acol = (((t & 7) ^ ((arow >> 1) & 7))) * 4;
    const float* a_ptr[A_PASSES];
    const float* b_ptr[B_PASSES];
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) a_ptr[i] = A + (size_t)min(((FLAGS & SAME_A) ? 0 : m0) + arow + 32 * i, M - 1) * K + acol;
#pragma unroll
    for (int p = 0; p < B_PASSES; ++p) b_ptr[p] = Bm + (size_t)min(n0 + arow + 32 * p, N - 1) * K + acol;
    const int nk = K / BK;
    auto stage_slab = [&](int kt, int buf) {
        float* la = sA + buf * A_TILE + wave * (8 * BK);
        float* lb = sB + buf * B_TILE + wave * (8 * BK);
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) glds16(a_ptr[i] + kt * BK, la + i * 32 * BK);
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p) glds16(b_ptr[p] + kt * BK, lb + p * 32 * BK);
    };
    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int fsw = (l31 >> 1) & 7;
    int frag_off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) frag_off[q] = ((2 * q + hi) ^ fsw) * 4;
    const int a_row = (wm * TM * 32 + l31) * BK;
    const int b_row = (wn * TN * 32 + l31) * BK;
    f32x4 fa[2][TM], fb[2][TN];
    auto load_frags = [&](int set, int buf, int q) {
        if (FLAGS & NO_LDS_READ) return;
        const float* a = sA + buf * A_TILE + a_row + frag_off[q];
        const float* b = sB + buf * B_TILE + b_row + frag_off[q];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[set][i] = *reinterpret_cast<const f32x4*>(a + i * 32 * BK);
#pragma unroll
        for (int n = 0; n < TN; ++n) fb[set][n] = *reinterpret_cast<const f32x4*>(b + n * 32 * BK);
    };
    auto mfma_group = [&](int set) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][i][j], fb[set][n][j], acc[i][n], 0, 0, 0);
    };
    auto bar = [&]() {
        if (FLAGS & NO_BARRIER) return;
        if (FLAGS & PIN) __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (FLAGS & PIN) __builtin_amdgcn_sched_barrier(0);
    };
    if (FLAGS & NO_LDS_READ) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[s][i] = f32x4{1.f, 2.f, 3.f, 4.f} * (float)(t + i);
#pragma unroll
            for (int i = 0; i < TN; ++i) fb[s][i] = f32x4{1.f, 2.f, 3.f, 4.f} * (float)(t - i);
        }
    }
    stage_slab(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int buf = 0;
    load_frags(0, 0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const bool more = kt + 1 < nk;
        if (more && !(FLAGS & NO_GLOBAL)) stage_slab(kt + 1, buf ^ 1);
        load_frags(1, buf, 1);
        mfma_group(0);
        load_frags(0, buf, 2);
        mfma_group(1);
        load_frags(1, buf, 3);
        mfma_group(0);
        bar();
        if (more) load_frags(0, buf ^ 1, 0);
        mfma_group(1);
        buf ^= 1;
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const int m = m0 + row;
            if (m >= M) continue;
#pragma unroll
            for (int n = 0; n < TN; ++n) {
                const int col = n0 + (wn * TN + n) * 32 + l31;
                if (col < N) C[(size_t)m * N + col] = acc[i][n][r];
            }
        }
    probe.done();
}

struct Problem {
    int M, N, K;
    float *A, *B, *C;
    std::vector<float> hA, hB;
};

static double check(Problem& p) {   // max relative error over a sample of entries, against a double dot product
    std::vector<float> hC((size_t)p.M * p.N);
    hipMemcpy(hC.data(), p.C, hC.size() * 4, hipMemcpyDeviceToHost);
    double worst = 0;
    for (int s = 0; s < 64; ++s) {
        const int m = (int)((1315423911u * (unsigned)(s + 1)) % (unsigned)p.M), n = (int)((2654435761u * (unsigned)(s + 7)) % (unsigned)p.N);
        double ref = 0;
        for (int k = 0; k < p.K; ++k) ref += (double)p.hA[(size_t)m * p.K + k] * p.hB[(size_t)n * p.K + k];
        const double e = fabs(ref - hC[(size_t)m * p.N + n]) / (fabs(ref) + 1e-3);
        if (e > worst) worst = e;
    }
    return worst;
}

template <typename F>
static void bench(const char* name, Problem& p, F launch, bool verify) {
    hipMemset(p.C, 0, (size_t)p.M * p.N * 4);
    launch();
    hipDeviceSynchronize();
    if (hipGetLastError() != hipSuccess) {
        printf("%-46s LAUNCH FAILED\n", name);
        return;
    }
    const double err = verify ? check(p) : -1;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int reps = getenv("REPS") ? atoi(getenv("REPS")) : 10;
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double tf = 2.0 * p.M * p.N * p.K / (ms * 1e-3) / 1e12;
    long long pr[2] = {0, 1};
    hipMemcpyFromSymbol(pr, HIP_SYMBOL(g_probe), sizeof(pr));
    printf("%-46s %8.3f ms %7.1f TF %5.1f%%  err %.1e  sclk %.0f MHz\n", name, ms, tf, 100 * tf / 157.3, err, 100.0 * pr[0] / pr[1]);
    fflush(stdout);
}

template <int BM, int BN, int WM, int WN, int TM, int TN, int FLAGS>
static void run_v0(const char* name, Problem& p) {
    const int tm = (p.M + BM - 1) / BM, tn = (p.N + BN - 1) / BN;
    bench(name, p, [&]() {
        hipLaunchKernelGGL((gemm_v0<BM, BN, WM, WN, TM, TN, FLAGS>), dim3(tm * tn), dim3(256), 0, 0, p.A, p.B, p.C, p.M, p.N, p.K, tn);
    }, FLAGS == 0 || FLAGS == PIN);  // ablations compute something else
}

#include "gemm_lab_v1.inc"

int main(int argc, char** argv) {
    Problem p;
    p.M = argc > 3 ? atoi(argv[1]) : 49152;
    p.N = argc > 3 ? atoi(argv[2]) : 256;
    p.K = argc > 3 ? atoi(argv[3]) : 2304;
    printf("M=%d N=%d K=%d\n", p.M, p.N, p.K);
    p.hA.resize((size_t)p.M * p.K);
    p.hB.resize((size_t)p.N * p.K);
    unsigned s = 12345;
    for (auto& v : p.hA) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
    for (auto& v : p.hB) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
    hipMalloc(&p.A, p.hA.size() * 4);
    hipMalloc(&p.B, p.hB.size() * 4);
    hipMalloc(&p.C, (size_t)p.M * p.N * 4);
    hipMemcpy(p.A, p.hA.data(), p.hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(p.B, p.hB.data(), p.hB.size() * 4, hipMemcpyHostToDevice);

    run_v0<128, 128, 2, 2, 2, 2, 0>("v0 128x128 baseline", p);
    run_v0<128, 128, 2, 2, 2, 2, PIN>("v0 128x128 pinned barrier", p);
    run_v0<128, 128, 2, 2, 2, 2, SAME_A>("v0 128x128 all tiles load the same A rows", p);
    run_v0<128, 128, 2, 2, 2, 2, NO_GLOBAL>("v0 128x128 no global loads", p);
    run_v0<128, 128, 2, 2, 2, 2, NO_GLOBAL | NO_BARRIER>("v0 128x128 no global, no barrier", p);
    run_v0<128, 128, 2, 2, 2, 2, NO_GLOBAL | NO_BARRIER | NO_LDS_READ>("v0 128x128 MFMA only", p);
    run_v0<128, 128, 2, 2, 2, 2, NO_LDS_READ>("v0 128x128 global+barrier, no LDS read", p);
    run_v0<128, 64, 2, 2, 2, 1, 0>("v0 128x64 baseline", p);
    run_v0<256, 64, 4, 1, 2, 2, 0>("v0 256x64 baseline (waves 4x1, 64x64 each)", p);
    run_v0<128, 64, 2, 2, 2, 1, NO_GLOBAL | NO_BARRIER>("v0 128x64 no global, no barrier", p);
    run_v0<128, 64, 2, 2, 2, 1, NO_GLOBAL | NO_BARRIER | NO_LDS_READ>("v0 128x64 MFMA only", p);
    run_lab_v1(p);
    run_lab_v2(p);
    run_lab_v3(p);
    run_lab_v4(p);
    return 0;
}

