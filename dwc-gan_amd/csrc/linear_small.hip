// nn.Linear (+ ReLU) on activations of a few dozen to a few hundred rows: the style -> AdaIN-parameter MLP (reference
// networks.py:491-503, LinearBlock :587-634) and the style encoder's mapping (networks_v2.py:116-121), forward and backward.
//
// Rounds 1-5 ran these as 1x1 convolutions on the im2col GEMM: 16..384 rows against 64..4096 columns and a 64..256-deep contraction --
// 20-29 us per launch whatever the size (one 64-row tile mostly empty, a split-K reduce behind it), 55 launches and 1.2 ms per c1 step,
// plus the activation-backward / bias-gradient pass and the weight-gradient reduce behind each of them in the backward.  They are far too
// small for any of that machinery: here a wave computes a 16 x 16 output tile and walks the contraction with v_mfma_f32_16x16x4_f32 --
// fp32 operands, fp32 products, fp32 accumulation (the exact fp32 matrix instruction: 1/16 of the 16-bit rate, irrelevant at 0.03-0.8
// GFLOP) -- reading its operands straight from global memory / L2 (16 bytes per lane where the contraction runs along the rows); the waves
// of a workgroup split the contraction of one tile (lin_contract below).  Three kernels: forward (+ bias, + ReLU), data gradient (ReLU mask applied while dY is loaded) and weight +
// bias gradient (mask likewise; contraction over the rows), i.e. three launches per layer and step instead of 6-8.
#include <algorithm>
#include "dwc_common.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4v lin_mfma(float a, float b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int LIN_MAX_WAVES = 16;   // waves of one workgroup = slices of the contraction of ONE output tile
constexpr int LIN_GROUP = 8;        // 16-deep steps whose operands are requested together (one memory round trip per group)

// The launches are latency-bound, not bandwidth- or FLOP-bound (a first version with one wave walking the whole contraction behind a
// one-deep prefetch measured SLOWER than the im2col GEMM it replaced: 256 dependent round trips for the 4096-deep data gradient).  So:
// a workgroup owns one tile, its waves split the contraction, each wave requests the operands of LIN_GROUP steps before the first
// product, and the partial tiles are summed through LDS by wave 0 in wave order (fixed order: results do not depend on scheduling).
template <class LoadA, class LoadB>
__device__ __forceinline__ f32x4v lin_contract(int steps, LoadA load_a, LoadB load_b, float* red) {
    const int nw = blockDim.x >> 6, wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int per = (steps + nw - 1) / nw, s0 = wave * per, s1 = min(steps, s0 + per);
    f32x4v acc = {0.f, 0.f, 0.f, 0.f};
    for (int s = s0; s < s1; s += LIN_GROUP) {
        f32x4v a[LIN_GROUP], b[LIN_GROUP];
#pragma unroll
        for (int u = 0; u < LIN_GROUP; ++u) {
            const int su = min(s + u, s1 - 1);                  // (past the end: a repeated address, products skipped below)
            a[u] = load_a(su);
            b[u] = load_b(su);
        }
#pragma unroll
        for (int u = 0; u < LIN_GROUP; ++u) {
            if (s + u < s1) {
#pragma unroll
                for (int i = 0; i < 4; ++i) acc = lin_mfma(a[u][i], b[u][i], acc);
            }
        }
    }
    if (nw > 1) {
        if (wave) *reinterpret_cast<f32x4v*>(red + (wave * 64 + lane) * 4) = acc;
        __syncthreads();
        if (!wave)
            for (int w = 1; w < nw; ++w) {
                const f32x4v o = *reinterpret_cast<const f32x4v*>(red + (w * 64 + lane) * 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] += o[i];
            }
    }
    return acc;                                                 // complete in wave 0 only
}

// Operand layout of v_mfma_f32_16x16x4_f32 number i of a 16-deep step: lane (r = l % 16, q = l / 16) holds A[row r][k0 + 4 q + i] and
// B[k0 + 4 q + i][column r]; result D[row 4 q + v][column r] in acc[v].

// y[M][N] = act(x[M][K] . w[N][K]^T + bias[N]).  K % 16 == 0, N % 16 == 0.  Both operands: one 16-byte load per lane and step.
__global__ __launch_bounds__(64 * LIN_MAX_WAVES) void linear_small_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                              const float* __restrict__ bias, float* __restrict__ y, int M,
                                                                              int N, int K, int relu, int tiles_n) {
    __shared__ float red[LIN_MAX_WAVES * 256];
    const int tn = blockIdx.x % tiles_n, tm = blockIdx.x / tiles_n, lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
    const int m = min(tm * 16 + r, M - 1);                      // (rows past M: a valid address, results not stored)
    const float* xa = x + (size_t)m * K + 4 * q;
    const float* wb = w + (size_t)(tn * 16 + r) * K + 4 * q;
    const f32x4v acc = lin_contract(K / 16, [&](int s) { return *reinterpret_cast<const f32x4v*>(xa + 16 * s); },
                                    [&](int s) { return *reinterpret_cast<const f32x4v*>(wb + 16 * s); }, red);
    if (threadIdx.x >= 64) return;
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

// dx[M][K] = g[M][N] . w[N][K], g = dy (relu: dy where y > 0, else 0).  N % 16 == 0, K % 16 == 0.  One workgroup per 16 x 16 tile of dx.
__global__ __launch_bounds__(64 * LIN_MAX_WAVES) void linear_small_dgrad_kernel(const float* __restrict__ dy, const float* __restrict__ yact,
                                                                                const float* __restrict__ w, float* __restrict__ dx, int M,
                                                                                int N, int K, int tiles_k) {
    __shared__ float red[LIN_MAX_WAVES * 256];
    const int tk = blockIdx.x % tiles_k, tm = blockIdx.x / tiles_k, lane = threadIdx.x & 63;
    const int r = lane & 15, q = lane >> 4;
    const int m = min(tm * 16 + r, M - 1);
    const float* ga = dy + (size_t)m * N + 4 * q;
    const float* ya = yact ? yact + (size_t)m * N + 4 * q : nullptr;
    const float* wb = w + (size_t)(4 * q) * K + tk * 16 + r;    // B[n0 + 4 q + i][column r]: four rows of w, 64 bytes each across the lanes
    const f32x4v acc = lin_contract(
        N / 16,
        [&](int s) {
            f32x4v a = *reinterpret_cast<const f32x4v*>(ga + 16 * s);
            if (ya) {
                const f32x4v yv = *reinterpret_cast<const f32x4v*>(ya + 16 * s);
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = yv[i] > 0.f ? a[i] : 0.f;
            }
            return a;
        },
        [&](int s) {
            f32x4v b;
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = wb[(size_t)(16 * s + i) * K];
            return b;
        },
        red);
    if (threadIdx.x >= 64) return;
    const int k = tk * 16 + r;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
        const int row = tm * 16 + 4 * q + v;
        if (row < M) dx[(size_t)row * K + k] = acc[v];
    }
}

// dw[N][K] = g^T . x (contraction over the M rows), db[N] = column sums of g; g as above.  One workgroup per 16 x 16 tile of dw; the bias
// gradient by its own workgroups behind the tiles (blockIdx >= tiles: 16 columns each, the rows split over the waves likewise).
__global__ __launch_bounds__(64 * LIN_MAX_WAVES) void linear_small_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ yact,
                                                                                const float* __restrict__ x, float* __restrict__ dw,
                                                                                float* __restrict__ db, int M, int N, int K, int tiles_k,
                                                                                int tiles) {
    __shared__ float red[LIN_MAX_WAVES * 256];
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    if ((int)blockIdx.x >= tiles) {                             // ---- bias gradient: columns 16 (blockIdx - tiles) ..
        const int n = ((int)blockIdx.x - tiles) * 16 + r, part = threadIdx.x >> 4, parts = blockDim.x >> 4;
        float s = 0.f;
        for (int m = part; m < M; m += parts) {
            const size_t gi = (size_t)m * N + n;
            float gv = dy[gi];
            if (yact) gv = yact[gi] > 0.f ? gv : 0.f;
            s += gv;
        }
        red[part * 16 + r] = s;
        __syncthreads();
        if (threadIdx.x < 16) {
            float t = 0.f;
            for (int p = 0; p < parts; ++p) t += red[p * 16 + r];
            db[n] = t;
        }
        return;
    }
    const int tk = blockIdx.x % tiles_k, tn = blockIdx.x / tiles_k;
    const float* gcol = dy + tn * 16 + r;
    const float* ycol = yact ? yact + tn * 16 + r : nullptr;
    const float* xcol = x + tk * 16 + r;
    const f32x4v acc = lin_contract(
        (M + 15) / 16,
        [&](int s) {                                            // A[n = r][m0 + 4 q + i]
            f32x4v a;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = 16 * s + 4 * q + i;
                const size_t gi = (size_t)min(m, M - 1) * N;
                float gv = gcol[gi];
                if (ycol) gv = ycol[gi] > 0.f ? gv : 0.f;
                a[i] = m < M ? gv : 0.f;
            }
            return a;
        },
        [&](int s) {                                            // B[m0 + 4 q + i][k = r]
            f32x4v b;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int m = 16 * s + 4 * q + i;
                const float xv = xcol[(size_t)min(m, M - 1) * K];
                b[i] = m < M ? xv : 0.f;
            }
            return b;
        },
        red);
    if (threadIdx.x >= 64) return;
#pragma unroll
    for (int v = 0; v < 4; ++v) dw[(size_t)(tn * 16 + 4 * q + v) * K + tk * 16 + r] = acc[v];
}

// waves per workgroup for a contraction of `steps` 16-deep steps: about one group of requests per wave
int linear_small_waves(int steps) { return std::max(1, std::min(LIN_MAX_WAVES, (steps + LIN_GROUP - 1) / LIN_GROUP)); }

bool linear_small_shape_ok(int M, int N, int K) { return M > 0 && N >= 16 && K >= 16 && !(N & 15) && !(K & 15) && M <= 4096; }

}  // namespace

extern "C" {

int dwc_linear_small_ok(int M, int N, int K) { return linear_small_shape_ok(M, N, K) ? 1 : 0; }

int dwc_linear_small_fwd(const float* x, const float* w, const float* bias, float* y, int M, int N, int K, int relu, void* stream) {
    if (!x || !w || !y || !linear_small_shape_ok(M, N, K)) return DWC_EINVAL;
    const int tiles_n = N / 16, tiles_m = (M + 15) / 16;
    hipLaunchKernelGGL(linear_small_fwd_kernel, dim3(tiles_n * tiles_m), dim3(64 * linear_small_waves(K / 16)), 0, (hipStream_t)stream, x, w,
                       bias, y, M, N, K, relu, tiles_n);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

int dwc_linear_small_bwd(const float* dy, const float* y_relu, const float* x, const float* w, float* dx, float* dw, float* db, int M,
                         int N, int K, void* stream) {
    if (!dy || !x || !w || !linear_small_shape_ok(M, N, K)) return DWC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (dx) {
        const int tiles_k = K / 16;
        hipLaunchKernelGGL(linear_small_dgrad_kernel, dim3(tiles_k * ((M + 15) / 16)), dim3(64 * linear_small_waves(N / 16)), 0, st, dy,
                           y_relu, w, dx, M, N, K, tiles_k);
        DWC_LAUNCH_CHECK();
    }
    if (dw) {
        const int tiles_k = K / 16, tiles = tiles_k * (N / 16);
        hipLaunchKernelGGL(linear_small_wgrad_kernel, dim3(tiles + (db ? N / 16 : 0)), dim3(64 * linear_small_waves((M + 15) / 16)), 0, st,
                           dy, y_relu, x, dw, db, M, N, K, tiles_k, tiles);
        DWC_LAUNCH_CHECK();
    }
    return DWC_OK;
}

}  // extern "C"
