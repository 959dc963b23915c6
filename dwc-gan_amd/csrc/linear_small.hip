// nn.Linear (+ ReLU) on activations of a few dozen to a few hundred rows: the style -> AdaIN-parameter MLP (reference
// networks.py:491-503, LinearBlock :587-634) and the style encoder's mapping (networks_v2.py:116-121), forward and backward.
//
// Rounds 1-5 ran these as 1x1 convolutions on the im2col GEMM: 16..384 rows against 64..4096 columns and a 64..256-deep contraction --
// 20-29 us per launch whatever the size (one 64-row tile mostly empty, a split-K reduce behind it), 55 launches and 1.2 ms per c1 step,
// plus the activation-backward / bias-gradient pass and the weight-gradient reduce behind each of them in the backward.  They are far too
// small for any of that machinery: here ONE wave owns a 16 x 16 output tile and walks the contraction with v_mfma_f32_16x16x4_f32 --
// fp32 operands, fp32 products, fp32 accumulation (the exact fp32 matrix instruction: 1/16 of the 16-bit rate, irrelevant at 0.03-0.8
// GFLOP) -- reading its operands straight from global memory / L2 (16 bytes per lane where the contraction runs along the rows), no LDS,
// no barrier, no scratch.  Three kernels: forward (+ bias, + ReLU), data gradient (ReLU mask applied while dY is loaded) and weight +
// bias gradient (mask likewise; contraction over the rows), i.e. three launches per layer and step instead of 6-8.
#include "dwc_common.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4v lin_mfma(float a, float b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// y[M][N] = act(x[M][K] . w[N][K]^T + bias[N]).  K % 16 == 0, N % 16 == 0.  One wave per 16 x 16 tile, four N tiles per workgroup.
// Operand of MFMA i of a 16-deep step: lane (r = l % 16, q = l / 16) holds A[row r][k0 + 4 q + i] and B[k0 + 4 q + i][column r]: one
// 16-byte load per operand and step, four instructions on it.
__global__ __launch_bounds__(256) void linear_small_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               const float* __restrict__ bias, float* __restrict__ y, int M, int N, int K,
                                                               int relu, int tiles_n) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int tn = wave % tiles_n, tm = wave / tiles_n;
    if (tm * 16 >= M) return;
    const int r = lane & 15, q = lane >> 4;
    const int m = min(tm * 16 + r, M - 1);                      // (rows past M: a valid address, results not stored)
    const float* xa = x + (size_t)m * K + 4 * q;
    const float* wb = w + (size_t)(tn * 16 + r) * K + 4 * q;
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
    f32x4v a = *reinterpret_cast<const f32x4v*>(xa), b = *reinterpret_cast<const f32x4v*>(wb);
    for (int k0 = 16; k0 <= K; k0 += 16) {
        f32x4v an = a, bn = b;
        if (k0 < K) {                                           // next step's operands in flight during this step's products
            an = *reinterpret_cast<const f32x4v*>(xa + k0);
            bn = *reinterpret_cast<const f32x4v*>(wb + k0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = lin_mfma(a[i], b[i], acc);
        a = an;
        b = bn;
    }
    // D[row 4 q + v][column r]
    const int n = tn * 16 + r;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int row = tm * 16 + 4 * q + v;
        float o = acc[v] + bv;
        if (relu) o = o < 0.f ? 0.f : o;                        // NaN-preserving
        if (row < M) y[(size_t)row * N + n] = o;
    }
}

// dx[M][K] = g[M][N] . w[N][K], g = dy (relu: dy where y > 0, else 0).  N % 16 == 0, K % 16 == 0.  One wave per 16 x 16 tile of dx.
__global__ __launch_bounds__(256) void linear_small_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ yact,
                                                                 const float* __restrict__ w, float* __restrict__ dx, int M, int N, int K,
                                                                 int tiles_k) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int tk = wave % tiles_k, tm = wave / tiles_k;
    if (tm * 16 >= M) return;
    const int r = lane & 15, q = lane >> 4;
    const int m = min(tm * 16 + r, M - 1);
    const float* ga = dy + (size_t)m * N + 4 * q;
    const float* ya = yact ? yact + (size_t)m * N + 4 * q : nullptr;
    const float* wb = w + (size_t)(4 * q) * K + tk * 16 + r;    // B[n0 + 4 q + i][column r]: four rows of w, 64 bytes each across the lanes
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
    for (int n0 = 0; n0 < N; n0 += 16) {
        f32x4v a = *reinterpret_cast<const f32x4v*>(ga + n0);
        if (ya) {
            const f32x4v yv = *reinterpret_cast<const f32x4v*>(ya + n0);
#pragma unroll
            for (int i = 0; i < 4; ++i) a[i] = yv[i] > 0.f ? a[i] : 0.f;
        }
        float b[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) b[i] = wb[(size_t)(n0 + i) * K];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = lin_mfma(a[i], b[i], acc);
    }
    const int k = tk * 16 + r;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int row = tm * 16 + 4 * q + v;
        if (row < M) dx[(size_t)row * K + k] = acc[v];
    }
}

// dw[N][K] = g^T . x (contraction over the M rows), db[N] = column sums of g (written by the tiles of the first K column); g as above.
// One wave per 16 x 16 tile of dw.
__global__ __launch_bounds__(256) void linear_small_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ yact,
                                                                 const float* __restrict__ x, float* __restrict__ dw, float* __restrict__ db,
                                                                 int M, int N, int K, int tiles_k) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6, lane = threadIdx.x & 63;
    const int tk = wave % tiles_k, tn = wave / tiles_k;
    if (tn * 16 >= N) return;
    const int r = lane & 15, q = lane >> 4;
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
    float colsum = 0.f;
    for (int m0 = 0; m0 < M; m0 += 16) {
        float a[4], b[4];                                        // A[n = r][m0 + 4 q + i], B[m0 + 4 q + i][k = r]
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + 4 * q + i;
            const bool ok = m < M;
            const size_t gi = (size_t)min(m, M - 1) * N + tn * 16 + r;
            float gv = ok ? dy[gi] : 0.f;
            if (yact && ok) gv = yact[gi] > 0.f ? gv : 0.f;
            a[i] = gv;
            b[i] = ok ? x[(size_t)m * K + tk * 16 + r] : 0.f;
            colsum += gv;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = lin_mfma(a[i], b[i], acc);
    }
    // D[row n = 4 q + v][column k = r]
#pragma unroll
    for (int v = 0; v < 4; ++v) dw[(size_t)(tn * 16 + 4 * q + v) * K + tk * 16 + r] = acc[v];
    if (db && tk == 0) {
        // lanes (r, q = 0..3) hold the partial sums of column n = r over their rows: fold the four lane groups (fixed order)
        const float s1 = __shfl(colsum, r + 16), s2 = __shfl(colsum, r + 32), s3 = __shfl(colsum, r + 48);
        if (q == 0) db[tn * 16 + r] = ((colsum + s1) + s2) + s3;
    }
}

bool linear_small_shape_ok(int M, int N, int K) { return M > 0 && N >= 16 && K >= 16 && !(N & 15) && !(K & 15) && M <= 4096; }

}  // namespace

extern "C" {

int dwc_linear_small_ok(int M, int N, int K) { return linear_small_shape_ok(M, N, K) ? 1 : 0; }

int dwc_linear_small_fwd(const float* x, const float* w, const float* bias, float* y, int M, int N, int K, int relu, void* stream) {
    if (!x || !w || !y || !linear_small_shape_ok(M, N, K)) return DWC_EINVAL;
    const int tiles_n = N / 16, tiles_m = (M + 15) / 16;
    const int waves = tiles_n * tiles_m;
    hipLaunchKernelGGL(linear_small_fwd_kernel, dim3((waves + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, w, bias, y, M, N, K, relu, tiles_n);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

int dwc_linear_small_bwd(const float* dy, const float* y_relu, const float* x, const float* w, float* dx, float* dw, float* db, int M,
                         int N, int K, void* stream) {
    if (!dy || !x || !w || !linear_small_shape_ok(M, N, K)) return DWC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dx) {
        const int tiles_k = K / 16, waves = tiles_k * ((M + 15) / 16);
        hipLaunchKernelGGL(linear_small_dgrad_kernel, dim3((waves + 3) / 4), dim3(256), 0, st, dy, y_relu, w, dx, M, N, K, tiles_k);
        DWC_LAUNCH_CHECK();
    }
    if (dw) {
        const int tiles_k = K / 16, waves = tiles_k * (N / 16);
        hipLaunchKernelGGL(linear_small_wgrad_kernel, dim3((waves + 3) / 4), dim3(256), 0, st, dy, y_relu, x, dw, db, M, N, K, tiles_k);
        DWC_LAUNCH_CHECK();
    }
    return DWC_OK;
}

}  // extern "C"
