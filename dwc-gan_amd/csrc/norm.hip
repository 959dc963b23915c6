// Instance norm / AdaIN and MUNIT LayerNorm, forward and backward, NHWC fp32 (gfx950).
//
//  * nn.InstanceNorm2d (reference networks.py:545) and AdaptiveInstanceNorm2d
//    (reference networks.py:706-719: F.batch_norm over a (1,B*C,H,W) view = per-(n,c) biased
//    variance, eps inside the sqrt, then weight[n,c]*xhat + bias[n,c]).
//  * LayerNorm (reference networks.py:736-752: per-sample mean, UNBIASED std, divide by
//    (std+eps), per-channel gamma/beta).
//  * the ReLU that follows the norm in Conv2dBlock (networks.py:583-584) and ResBlock's
//    residual add (networks.py:521) are fused into the apply pass.
//
// All of these are HBM-bound: one statistics pass (16-byte loads, channels on the lanes so a
// wave reads whole 1 KiB rows, cross-row reduction through LDS) and one apply pass.
// Statistics are accumulated about a per-(n,c) pivot (the first pixel) so that
// E[x^2]-E[x]^2 does not cancel.  Partials go to caller scratch and are combined in a fixed
// order (bitwise reproducible; no atomics).
#include <stdlib.h>

#include "dwc_common.h"

namespace {

// rows-per-sample are split into `chunks`; geometry shared by the statistics kernels
struct RowSplit {
    int chunks, rows_per_chunk;
};

RowSplit plan_rows(int B, int HW) {
    int chunks = HW / 64;
    int cap = 2048 / (B > 0 ? B : 1);
    if (cap < 1) cap = 1;
    if (chunks > cap) chunks = cap;
    if (chunks < 1) chunks = 1;
    RowSplit r;
    r.rows_per_chunk = (HW + chunks - 1) / chunks;
    r.chunks = (HW + r.rows_per_chunk - 1) / r.rows_per_chunk;
    return r;
}

// ---------------------------------------------------------------------------------------
// instance norm
// ---------------------------------------------------------------------------------------
// partial[(n*chunks+chunk)*C + c] = sum (x-pivot), second plane = sum (x-pivot)^2
template <typename T>
__global__ __launch_bounds__(256) void in_stats_partial(const T* __restrict__ x, float* __restrict__ part, int HW, int C,
                                                        int rows_per_chunk, size_t plane) {
    __shared__ f32x4 sm[2][256];
    const int cq = C >> 2;
    const int groups = 256 / cq;
    const int col = threadIdx.x % cq, rg = threadIdx.x / cq;
    const int n = blockIdx.y, chunk = blockIdx.x;
    const T* xs = x + (size_t)n * HW * C;
    const f32x4 piv = ld4(xs, col);
    const int r0 = chunk * rows_per_chunk;
    const int r1 = min(HW, r0 + rows_per_chunk);
    f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
    if (rg < groups)
        for (int r = r0 + rg; r < r1; r += groups) {
            const f32x4 v = ld4(xs, (size_t)r * cq + col) - piv;
            s1 += v;
            s2 += v * v;
        }
    sm[0][threadIdx.x] = s1;
    sm[1][threadIdx.x] = s2;
    __syncthreads();
    if (rg == 0) {
        for (int g = 1; g < groups; ++g) {
            s1 += sm[0][g * cq + col];
            s2 += sm[1][g * cq + col];
        }
        const size_t o = ((size_t)(n * gridDim.x + chunk) * C) + col * 4;
        *reinterpret_cast<f32x4*>(part + o) = s1;
        *reinterpret_cast<f32x4*>(part + plane + o) = s2;
    }
}

template <typename T>
__global__ void in_stats_final(const T* __restrict__ x, const float* __restrict__ part, float* __restrict__ mean,
                               float* __restrict__ rstd, int B, int HW, int C, int chunks, size_t plane, float eps) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * C) return;
    const int n = idx / C, c = idx - n * C;
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < chunks; ++k) {
        s1 += part[(size_t)(n * chunks + k) * C + c];
        s2 += part[plane + (size_t)(n * chunks + k) * C + c];
    }
    const float inv = 1.f / (float)HW;
    const float d = s1 * inv;
    const float var = fmaxf(s2 * inv - d * d, 0.f);
    mean[idx] = (float)x[(size_t)n * HW * C + c] + d;
    rstd[idx] = 1.f / sqrtf(var + eps);
}

template <typename T>
__global__ __launch_bounds__(256) void in_apply(const T* __restrict__ x, const float* __restrict__ mean,
                                                const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, const T* __restrict__ residual,
                                                T* __restrict__ y, int HW, int C, size_t total4, int relu) {
    const int cq = C >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = i % cq;
        const size_t n = i / ((size_t)HW * cq);
        const size_t s = n * cq + c4;
        f32x4 sc = reinterpret_cast<const f32x4*>(rstd)[s];
        if (gamma) sc *= reinterpret_cast<const f32x4*>(gamma)[s];
        f32x4 v = (ld4(x, i) - reinterpret_cast<const f32x4*>(mean)[s]) * sc;
        if (beta) v += reinterpret_cast<const f32x4*>(beta)[s];
        if (relu) {
            v[0] = v[0] < 0.f ? 0.f : v[0]; v[1] = v[1] < 0.f ? 0.f : v[1]; v[2] = v[2] < 0.f ? 0.f : v[2]; v[3] = v[3] < 0.f ? 0.f : v[3];   // NaN-preserving
        }
        if (residual) v += ld4(residual, i);
        st4(y, i, v);
    }
}

// effective upstream gradient: masks by the ReLU that followed the norm
__device__ __forceinline__ f32x4 relu_mask(f32x4 dy, f32x4 pre, int relu) {
    if (relu) {
        dy[0] = pre[0] > 0.f ? dy[0] : 0.f;
        dy[1] = pre[1] > 0.f ? dy[1] : 0.f;
        dy[2] = pre[2] > 0.f ? dy[2] : 0.f;
        dy[3] = pre[3] > 0.f ? dy[3] : 0.f;
    }
    return dy;
}

template <typename T>
__global__ __launch_bounds__(256) void in_bwd_partial(const T* __restrict__ dy, const T* __restrict__ x,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ part, int HW, int C, int rows_per_chunk, size_t plane,
                                                      int relu) {
    __shared__ f32x4 sm[2][256];
    const int cq = C >> 2;
    const int groups = 256 / cq;
    const int col = threadIdx.x % cq, rg = threadIdx.x / cq;
    const int n = blockIdx.y, chunk = blockIdx.x;
    const size_t base = (size_t)n * HW * cq;
    const T* xs = x + base * 4;
    const T* ds = dy + base * 4;
    const size_t s = (size_t)n * cq + col;
    const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[s];
    const f32x4 rs = reinterpret_cast<const f32x4*>(rstd)[s];
    f32x4 ga = {1, 1, 1, 1}, be = {0, 0, 0, 0};
    if (gamma) ga = reinterpret_cast<const f32x4*>(gamma)[s];
    if (beta) be = reinterpret_cast<const f32x4*>(beta)[s];
    const int r0 = chunk * rows_per_chunk;
    const int r1 = min(HW, r0 + rows_per_chunk);
    f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
    if (rg < groups)
        for (int r = r0 + rg; r < r1; r += groups) {
            const f32x4 xh = (ld4(xs, (size_t)r * cq + col) - mu) * rs;
            const f32x4 g = relu_mask(ld4(ds, (size_t)r * cq + col), xh * ga + be, relu);
            s1 += g;
            s2 += g * xh;
        }
    sm[0][threadIdx.x] = s1;
    sm[1][threadIdx.x] = s2;
    __syncthreads();
    if (rg == 0) {
        for (int g = 1; g < groups; ++g) {
            s1 += sm[0][g * cq + col];
            s2 += sm[1][g * cq + col];
        }
        const size_t o = ((size_t)(n * gridDim.x + chunk) * C) + col * 4;
        *reinterpret_cast<f32x4*>(part + o) = s1;
        *reinterpret_cast<f32x4*>(part + plane + o) = s2;
    }
}

// sums[0..BC) = sum dy_eff, sums[BC..2BC) = sum dy_eff*xhat; also dgamma/dbeta when requested
__global__ void in_bwd_final(const float* __restrict__ part, float* __restrict__ sums, float* __restrict__ dgamma,
                             float* __restrict__ dbeta, int BC, int C, int chunks, size_t plane) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= BC) return;
    const int n = idx / C, c = idx - n * C;
    float s1 = 0.f, s2 = 0.f;
    for (int k = 0; k < chunks; ++k) {
        s1 += part[(size_t)(n * chunks + k) * C + c];
        s2 += part[plane + (size_t)(n * chunks + k) * C + c];
    }
    sums[idx] = s1;
    sums[BC + idx] = s2;
    if (dgamma) dgamma[idx] = s2;
    if (dbeta) dbeta[idx] = s1;
}

template <typename T>
__global__ __launch_bounds__(256) void in_bwd_apply(const T* __restrict__ dy, const T* __restrict__ x,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const float* __restrict__ sums, T* __restrict__ dx, int HW, int C, int BC,
                                                    size_t total4, int relu) {
    const int cq = C >> 2;
    const float inv_hw = 1.f / (float)HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = i % cq;
        const size_t n = i / ((size_t)HW * cq);
        const size_t s = n * cq + c4;
        const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[s];
        const f32x4 rs = reinterpret_cast<const f32x4*>(rstd)[s];
        f32x4 ga = {1, 1, 1, 1}, be = {0, 0, 0, 0};
        if (gamma) ga = reinterpret_cast<const f32x4*>(gamma)[s];
        if (beta) be = reinterpret_cast<const f32x4*>(beta)[s];
        const f32x4 s1 = reinterpret_cast<const f32x4*>(sums)[s];
        const f32x4 s2 = reinterpret_cast<const f32x4*>(sums + BC)[s];
        const f32x4 xh = (ld4(x, i) - mu) * rs;
        const f32x4 g = relu_mask(ld4(dy, i), xh * ga + be, relu);
        st4(dx, i, ga * rs * (g - s1 * inv_hw - xh * (s2 * inv_hw)));
    }
}

// ---------------------------------------------------------------------------------------
// instance norm, ONE launch per direction for the maps that dominate the step (the 32x32 ResBlock / AdaIN layers): a
// workgroup owns one sample and 32 channels (HW rows x 128 bytes fp32 / 64 bytes bf16 -- at most 512 KB, so its second pass
// re-reads what it has just pulled through L2), reduces the statistics through LDS in a fixed order, and applies.  Replaces
// three launches (partial statistics, combine, apply) and one of the two (forward) / two of the four (backward) HBM reads.
// ---------------------------------------------------------------------------------------
constexpr int IN_CG = 32;                // channels per workgroup
constexpr int IN_RG = 256 / (IN_CG / 4); // row groups: 32

// sum over the IN_RG row groups of two f32x4 per thread; result valid for threads with rg == 0 ... all (broadcast through LDS)
__device__ __forceinline__ void in_block_reduce(f32x4& s1, f32x4& s2, f32x4 (*sm)[256]) {
    const int t = threadIdx.x, col = t % (IN_CG / 4);
    sm[0][t] = s1;
    sm[1][t] = s2;
    __syncthreads();
    if (t < IN_CG / 4) {
        f32x4 a = sm[0][t], b = sm[1][t];
        for (int g = 1; g < IN_RG; ++g) {            // fixed order: bitwise reproducible
            a += sm[0][g * (IN_CG / 4) + t];
            b += sm[1][g * (IN_CG / 4) + t];
        }
        sm[0][t] = a;
        sm[1][t] = b;
    }
    __syncthreads();
    s1 = sm[0][col];
    s2 = sm[1][col];
}

template <typename T>
__global__ __launch_bounds__(256) void in_fused_fwd(const T* __restrict__ x, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, const T* __restrict__ residual, T* __restrict__ y,
                                                    float* __restrict__ mean, float* __restrict__ rstd, int HW, int C, float eps,
                                                    int relu) {
    __shared__ f32x4 sm[2][256];
    const int cq = C >> 2;
    const int col = threadIdx.x % (IN_CG / 4), rg = threadIdx.x / (IN_CG / 4);
    const int n = blockIdx.y, c4 = blockIdx.x * (IN_CG / 4) + col;
    const size_t base = (size_t)n * HW * cq + c4;
    const f32x4 piv = ld4(x, base);
    f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
    for (int r = rg; r < HW; r += IN_RG) {
        const f32x4 v = ld4(x, base + (size_t)r * cq) - piv;
        s1 += v;
        s2 += v * v;
    }
    in_block_reduce(s1, s2, sm);
    const float inv = 1.f / (float)HW;
    const f32x4 d = s1 * inv;
    f32x4 var = s2 * inv - d * d;
    f32x4 mu, rs;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        mu[k] = piv[k] + d[k];
        rs[k] = 1.f / sqrtf(fmaxf(var[k], 0.f) + eps);
    }
    const size_t sidx = (size_t)n * cq + c4;
    if (rg == 0) {
        reinterpret_cast<f32x4*>(mean)[sidx] = mu;
        reinterpret_cast<f32x4*>(rstd)[sidx] = rs;
    }
    f32x4 sc = rs, sh = {0, 0, 0, 0};
    if (gamma) sc *= reinterpret_cast<const f32x4*>(gamma)[sidx];
    if (beta) sh = reinterpret_cast<const f32x4*>(beta)[sidx];
    for (int r = rg; r < HW; r += IN_RG) {
        const size_t i = base + (size_t)r * cq;
        f32x4 v = (ld4(x, i) - mu) * sc + sh;
        if (relu) {
            v[0] = v[0] < 0.f ? 0.f : v[0]; v[1] = v[1] < 0.f ? 0.f : v[1]; v[2] = v[2] < 0.f ? 0.f : v[2]; v[3] = v[3] < 0.f ? 0.f : v[3];   // NaN-preserving
        }
        if (residual) v += ld4(residual, i);
        st4(y, i, v);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void in_fused_bwd(const T* __restrict__ dy, const T* __restrict__ x, const float* __restrict__ mean,
                                                    const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                    const float* __restrict__ beta, T* __restrict__ dx, float* __restrict__ dgamma,
                                                    float* __restrict__ dbeta, int HW, int C, int relu) {
    __shared__ f32x4 sm[2][256];
    const int cq = C >> 2;
    const int col = threadIdx.x % (IN_CG / 4), rg = threadIdx.x / (IN_CG / 4);
    const int n = blockIdx.y, c4 = blockIdx.x * (IN_CG / 4) + col;
    const size_t base = (size_t)n * HW * cq + c4;
    const size_t sidx = (size_t)n * cq + c4;
    const f32x4 mu = reinterpret_cast<const f32x4*>(mean)[sidx];
    const f32x4 rs = reinterpret_cast<const f32x4*>(rstd)[sidx];
    f32x4 ga = {1, 1, 1, 1}, be = {0, 0, 0, 0};
    if (gamma) ga = reinterpret_cast<const f32x4*>(gamma)[sidx];
    if (beta) be = reinterpret_cast<const f32x4*>(beta)[sidx];
    f32x4 s1 = {0, 0, 0, 0}, s2 = {0, 0, 0, 0};
    for (int r = rg; r < HW; r += IN_RG) {
        const size_t i = base + (size_t)r * cq;
        const f32x4 xh = (ld4(x, i) - mu) * rs;
        const f32x4 g = relu_mask(ld4(dy, i), xh * ga + be, relu);
        s1 += g;
        s2 += g * xh;
    }
    in_block_reduce(s1, s2, sm);
    if (rg == 0) {
        if (dgamma) reinterpret_cast<f32x4*>(dgamma)[sidx] = s2;
        if (dbeta) reinterpret_cast<f32x4*>(dbeta)[sidx] = s1;
    }
    const float inv_hw = 1.f / (float)HW;
    const f32x4 k1 = s1 * inv_hw, k2 = s2 * inv_hw, gs = ga * rs;
    for (int r = rg; r < HW; r += IN_RG) {
        const size_t i = base + (size_t)r * cq;
        const f32x4 xh = (ld4(x, i) - mu) * rs;
        const f32x4 g = relu_mask(ld4(dy, i), xh * ga + be, relu);
        st4(dx, i, gs * (g - k1 - xh * k2));
    }
}

// Opt-in (DWC_IN_FUSED=1): measured 1 % SLOWER than the three-launch form on both bench configurations (c1 219.3 vs 221.8,
// c2 975 vs 986 images/s) -- one 256-thread workgroup per (sample, 32 channels) walks its slice twice at far less memory
// parallelism than the chunked statistics + grid-wide apply passes, and that costs more than the launches and the re-read save.
static bool in_fused_ok(int B, int HW, int C) {
    static const int on = getenv("DWC_IN_FUSED") ? atoi(getenv("DWC_IN_FUSED")) : 0;
    return on && !(C % IN_CG) && HW <= 4096 && HW >= IN_RG && (long)B * (C / IN_CG) >= 16;
}

// ---------------------------------------------------------------------------------------
// MUNIT layer norm (statistics per sample over C*HW)
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void ln_stats_partial(const T* __restrict__ x, float* __restrict__ part, int HW, int C,
                                                        int rows_per_chunk) {
    __shared__ float sm[4];
    const int cq = C >> 2;
    const int n = blockIdx.y, chunk = blockIdx.x;
    const T* xs = x + (size_t)n * HW * C;
    const float piv = (float)x[(size_t)n * HW * C];
    const size_t e0 = (size_t)chunk * rows_per_chunk * cq;
    const size_t e1 = min((size_t)HW * cq, e0 + (size_t)rows_per_chunk * cq);
    float s1 = 0.f, s2 = 0.f;
    for (size_t e = e0 + threadIdx.x; e < e1; e += 256) {
        const f32x4 v = ld4(xs, e) - piv;
        s1 += (v[0] + v[1]) + (v[2] + v[3]);
        s2 += (v[0] * v[0] + v[1] * v[1]) + (v[2] * v[2] + v[3] * v[3]);
    }
    s1 = dwc_block_sum_256(s1, sm);
    s2 = dwc_block_sum_256(s2, sm);
    if (threadIdx.x == 0) {
        part[(size_t)(n * gridDim.x + chunk) * 2] = s1;
        part[(size_t)(n * gridDim.x + chunk) * 2 + 1] = s2;
    }
}

template <typename T>
__global__ void ln_stats_final(const T* __restrict__ x, const float* __restrict__ part, float* __restrict__ mean,
                               float* __restrict__ inv, int B, int HW, int C, int chunks, float eps) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= B) return;
    double s1 = 0.0, s2 = 0.0;
    for (int k = 0; k < chunks; ++k) {
        s1 += part[(size_t)(n * chunks + k) * 2];
        s2 += part[(size_t)(n * chunks + k) * 2 + 1];
    }
    const double N = (double)HW * C;
    const double d = s1 / N;
    double var = (s2 - s1 * d) / (N - 1.0);  // unbiased (torch.std default), reference networks.py:745
    if (var < 0) var = 0;
    mean[n] = (float)((double)(float)x[(size_t)n * HW * C] + d);
    inv[n] = (float)(1.0 / (sqrt(var) + (double)eps));
}

template <typename T>
__global__ __launch_bounds__(256) void ln_apply(const T* __restrict__ x, const float* __restrict__ mean,
                                                const float* __restrict__ inv, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, T* __restrict__ y, int HW, int C, size_t total4,
                                                int relu) {
    const int cq = C >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = i % cq;
        const size_t n = i / ((size_t)HW * cq);
        const float mu = mean[n], iv = inv[n];
        const f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[c4];
        const f32x4 be = reinterpret_cast<const f32x4*>(beta)[c4];
        f32x4 v = (ld4(x, i) - mu) * iv * ga + be;
        if (relu) {
            v[0] = v[0] < 0.f ? 0.f : v[0]; v[1] = v[1] < 0.f ? 0.f : v[1]; v[2] = v[2] < 0.f ? 0.f : v[2]; v[3] = v[3] < 0.f ? 0.f : v[3];   // NaN-preserving
        }
        st4(y, i, v);
    }
}

// per block: sample sums (sum g, sum g*(x-mu)) with g = dy_eff*gamma, and per-channel
// partials of dgamma (dy_eff*xhat) and dbeta (dy_eff)
template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_partial(const T* __restrict__ dy, const T* __restrict__ x,
                                                      const float* __restrict__ mean, const float* __restrict__ inv,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ part_s, float* __restrict__ part_c, int HW, int C,
                                                      int rows_per_chunk, int relu) {
    __shared__ f32x4 sm[2][256];
    __shared__ float sr[4];
    const int cq = C >> 2;
    const int groups = 256 / cq;
    const int col = threadIdx.x % cq, rg = threadIdx.x / cq;
    const int n = blockIdx.y, chunk = blockIdx.x;
    const size_t base = (size_t)n * HW * cq;
    const T* xs = x + base * 4;
    const T* ds = dy + base * 4;
    const float mu = mean[n], iv = inv[n];
    const f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[col];
    const f32x4 be = reinterpret_cast<const f32x4*>(beta)[col];
    const int r0 = chunk * rows_per_chunk;
    const int r1 = min(HW, r0 + rows_per_chunk);
    f32x4 dg = {0, 0, 0, 0}, db = {0, 0, 0, 0};
    float t1 = 0.f, t2 = 0.f;
    if (rg < groups)
        for (int r = r0 + rg; r < r1; r += groups) {
            const f32x4 xc = ld4(xs, (size_t)r * cq + col) - mu;
            const f32x4 xh = xc * iv;
            const f32x4 d = relu_mask(ld4(ds, (size_t)r * cq + col), xh * ga + be, relu);
            dg += d * xh;
            db += d;
            const f32x4 g = d * ga;
            t1 += (g[0] + g[1]) + (g[2] + g[3]);
            const f32x4 gx = g * xc;
            t2 += (gx[0] + gx[1]) + (gx[2] + gx[3]);
        }
    sm[0][threadIdx.x] = dg;
    sm[1][threadIdx.x] = db;
    __syncthreads();
    const size_t blk = (size_t)n * gridDim.x + chunk;
    if (rg == 0) {
        for (int g = 1; g < groups; ++g) {
            dg += sm[0][g * cq + col];
            db += sm[1][g * cq + col];
        }
        *reinterpret_cast<f32x4*>(part_c + blk * 2 * C + col * 4) = dg;
        *reinterpret_cast<f32x4*>(part_c + blk * 2 * C + C + col * 4) = db;
    }
    t1 = dwc_block_sum_256(t1, sr);
    t2 = dwc_block_sum_256(t2, sr);
    if (threadIdx.x == 0) {
        part_s[blk * 2] = t1;
        part_s[blk * 2 + 1] = t2;
    }
}

// blocks [0, B): per-sample sums -> sums[n*2..];  blocks [B, B + ceil(C/64)): channel sums -> dgamma/dbeta.
// 1024 threads; fixed summation order.
__global__ __launch_bounds__(1024) void ln_bwd_final(const float* __restrict__ part_s, const float* __restrict__ part_c,
                                                     float* __restrict__ sums, float* __restrict__ dgamma,
                                                     float* __restrict__ dbeta, int B, int C, int chunks) {
    __shared__ float sm[2][16][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    if ((int)blockIdx.x < B) {
        const int n = blockIdx.x;
        float a = 0.f, b = 0.f;
        for (int k = threadIdx.x; k < chunks; k += 1024) {
            a += part_s[(size_t)(n * chunks + k) * 2];
            b += part_s[(size_t)(n * chunks + k) * 2 + 1];
        }
        a = dwc_wave_sum(a);
        b = dwc_wave_sum(b);
        if (cl == 0) {
            sm[0][g][0] = a;
            sm[1][g][0] = b;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int k = 1; k < 16; ++k) {
                a += sm[0][k][0];
                b += sm[1][k][0];
            }
            sums[n * 2] = a;
            sums[n * 2 + 1] = b;
        }
        return;
    }
    const int c = ((int)blockIdx.x - B) * 64 + cl;
    float a = 0.f, b = 0.f;
    if (c < C)
        for (int k = g; k < B * chunks; k += 16) {
            a += part_c[(size_t)k * 2 * C + c];
            b += part_c[(size_t)k * 2 * C + C + c];
        }
    sm[0][g][cl] = a;
    sm[1][g][cl] = b;
    __syncthreads();
    if (g == 0 && c < C) {
#pragma unroll
        for (int k = 1; k < 16; ++k) {
            a += sm[0][k][cl];
            b += sm[1][k][cl];
        }
        dgamma[c] = a;
        dbeta[c] = b;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void ln_bwd_apply(const T* __restrict__ dy, const T* __restrict__ x,
                                                    const float* __restrict__ mean, const float* __restrict__ inv,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const float* __restrict__ sums, T* __restrict__ dx, int HW, int C,
                                                    size_t total4, float eps, int relu) {
    const int cq = C >> 2;
    const float N = (float)HW * (float)C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const int c4 = i % cq;
        const size_t n = i / ((size_t)HW * cq);
        const float mu = mean[n], iv = inv[n];
        const float sigma = 1.f / iv - eps;
        const float mean_g = sums[n * 2] / N;
        // d/dx of 1/(sigma+eps): -(1/(sigma+eps))^2 * (x-mu)/((N-1)*sigma), times sum g*(x-mu)
        const float k2 = sigma > 0.f ? iv * iv * sums[n * 2 + 1] / ((N - 1.f) * sigma) : 0.f;
        const f32x4 ga = reinterpret_cast<const f32x4*>(gamma)[c4];
        const f32x4 be = reinterpret_cast<const f32x4*>(beta)[c4];
        const f32x4 xc = ld4(x, i) - mu;
        const f32x4 d = relu_mask(ld4(dy, i), xc * iv * ga + be, relu);
        st4(dx, i, (d * ga - mean_g) * iv - xc * k2);
    }
}

bool norm_shape_ok(int B, int HW, int C) {
    const int l = dwc_ilog2_exact(C);
    return B > 0 && HW > 0 && l >= 2 && C <= 1024;
}

int grid_for(size_t total4) {
    size_t b = (total4 + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (int)b;
}

}  // namespace

namespace {

size_t instnorm_ws_bytes(int B, int HW, int C) {
    const RowSplit rs = plan_rows(B, HW);
    return ((size_t)2 * B * rs.chunks * C + (size_t)2 * B * C) * sizeof(float);
}

template <typename T>
int instnorm_fwd_t(const T* x, const float* gamma, const float* beta, const T* residual, T* y, float* mean,
                     float* rstd, int B, int HW, int C, float eps, int relu, void* ws, size_t ws_bytes, void* stream) {
    if (!norm_shape_ok(B, HW, C)) return DWC_EINVAL;
    if (!ws || ws_bytes < instnorm_ws_bytes(B, HW, C)) return DWC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (in_fused_ok(B, HW, C)) {
        hipLaunchKernelGGL(in_fused_fwd<T>, dim3(C / IN_CG, B), dim3(256), 0, st, x, gamma, beta, residual, y, mean, rstd, HW, C, eps,
                           relu);
        DWC_LAUNCH_CHECK();
        return DWC_OK;
    }
    const RowSplit rs = plan_rows(B, HW);
    const size_t plane = (size_t)B * rs.chunks * C;
    float* part = (float*)ws;
    hipLaunchKernelGGL(in_stats_partial<T>, dim3(rs.chunks, B), dim3(256), 0, st, x, part, HW, C, rs.rows_per_chunk, plane);
    DWC_LAUNCH_CHECK();
    hipLaunchKernelGGL(in_stats_final<T>, dim3((B * C + 255) / 256), dim3(256), 0, st, x, part, mean, rstd, B, HW, C, rs.chunks,
                       plane, eps);
    DWC_LAUNCH_CHECK();
    const size_t total4 = (size_t)B * HW * (C / 4);
    hipLaunchKernelGGL(in_apply<T>, dim3(grid_for(total4)), dim3(256), 0, st, x, mean, rstd, gamma, beta, residual, y, HW, C,
                       total4, relu);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

template <typename T>
int instnorm_bwd_t(const T* dy, const T* x, const float* mean, const float* rstd, const float* gamma,
                   const float* beta, T* dx, float* dgamma, float* dbeta, int B, int HW, int C, int relu, void* ws,
                     size_t ws_bytes, void* stream) {
    if (!norm_shape_ok(B, HW, C)) return DWC_EINVAL;
    if (!ws || ws_bytes < instnorm_ws_bytes(B, HW, C)) return DWC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (in_fused_ok(B, HW, C)) {
        hipLaunchKernelGGL(in_fused_bwd<T>, dim3(C / IN_CG, B), dim3(256), 0, st, dy, x, mean, rstd, gamma, beta, dx, dgamma, dbeta,
                           HW, C, relu);
        DWC_LAUNCH_CHECK();
        return DWC_OK;
    }
    const RowSplit rs = plan_rows(B, HW);
    const size_t plane = (size_t)B * rs.chunks * C;
    float* part = (float*)ws;
    float* sums = part + 2 * plane;
    hipLaunchKernelGGL(in_bwd_partial<T>, dim3(rs.chunks, B), dim3(256), 0, st, dy, x, mean, rstd, gamma, beta, part, HW, C,
                       rs.rows_per_chunk, plane, relu);
    DWC_LAUNCH_CHECK();
    hipLaunchKernelGGL(in_bwd_final, dim3((B * C + 255) / 256), dim3(256), 0, st, part, sums, dgamma, dbeta, B * C, C, rs.chunks,
                       plane);
    DWC_LAUNCH_CHECK();
    const size_t total4 = (size_t)B * HW * (C / 4);
    hipLaunchKernelGGL(in_bwd_apply<T>, dim3(grid_for(total4)), dim3(256), 0, st, dy, x, mean, rstd, gamma, beta, sums, dx, HW, C,
                       B * C, total4, relu);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

size_t layernorm_ws_bytes(int B, int HW, int C) {
    const RowSplit rs = plan_rows(B, HW);
    return ((size_t)B * rs.chunks * 2 + (size_t)B * rs.chunks * 2 * C + (size_t)2 * B + 16) * sizeof(float);
}

template <typename T>
int layernorm_fwd_t(const T* x, const float* gamma, const float* beta, T* y, float* mean, float* inv, int B, int HW,
                      int C, float eps, int relu, void* ws, size_t ws_bytes, void* stream) {
    if (!norm_shape_ok(B, HW, C) || (size_t)HW * C < 2) return DWC_EINVAL;
    if (!ws || ws_bytes < layernorm_ws_bytes(B, HW, C)) return DWC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const RowSplit rs = plan_rows(B, HW);
    float* part = (float*)ws;
    hipLaunchKernelGGL(ln_stats_partial<T>, dim3(rs.chunks, B), dim3(256), 0, st, x, part, HW, C, rs.rows_per_chunk);
    DWC_LAUNCH_CHECK();
    hipLaunchKernelGGL(ln_stats_final<T>, dim3((B + 63) / 64), dim3(64), 0, st, x, part, mean, inv, B, HW, C, rs.chunks, eps);
    DWC_LAUNCH_CHECK();
    const size_t total4 = (size_t)B * HW * (C / 4);
    hipLaunchKernelGGL(ln_apply<T>, dim3(grid_for(total4)), dim3(256), 0, st, x, mean, inv, gamma, beta, y, HW, C, total4, relu);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

template <typename T>
int layernorm_bwd_t(const T* dy, const T* x, const float* mean, const float* inv, const float* gamma,
                    const float* beta, T* dx, float* dgamma, float* dbeta, int B, int HW, int C, float eps, int relu,
                      void* ws, size_t ws_bytes, void* stream) {
    if (!norm_shape_ok(B, HW, C)) return DWC_EINVAL;
    if (!ws || ws_bytes < layernorm_ws_bytes(B, HW, C)) return DWC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const RowSplit rs = plan_rows(B, HW);
    float* part_s = (float*)ws;
    float* part_c = part_s + (size_t)B * rs.chunks * 2;
    float* sums = part_c + (size_t)B * rs.chunks * 2 * C;
    hipLaunchKernelGGL(ln_bwd_partial<T>, dim3(rs.chunks, B), dim3(256), 0, st, dy, x, mean, inv, gamma, beta, part_s, part_c, HW,
                       C, rs.rows_per_chunk, relu);
    DWC_LAUNCH_CHECK();
    hipLaunchKernelGGL(ln_bwd_final, dim3(B + (C + 63) / 64), dim3(1024), 0, st, part_s, part_c, sums, dgamma, dbeta, B, C,
                       rs.chunks);
    DWC_LAUNCH_CHECK();
    const size_t total4 = (size_t)B * HW * (C / 4);
    hipLaunchKernelGGL(ln_bwd_apply<T>, dim3(grid_for(total4)), dim3(256), 0, st, dy, x, mean, inv, gamma, beta, sums, dx, HW, C,
                       total4, eps, relu);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

}  // namespace

extern "C" {

size_t dwc_instnorm_ws_bytes(int B, int HW, int C) { return instnorm_ws_bytes(B, HW, C); }
size_t dwc_layernorm_ws_bytes(int B, int HW, int C) { return layernorm_ws_bytes(B, HW, C); }

int dwc_instnorm_fwd(const float* x, const float* gamma, const float* beta, const float* residual, float* y, float* mean,
                     float* rstd, int B, int HW, int C, float eps, int relu, void* ws, size_t ws_bytes, void* stream) {
    return instnorm_fwd_t<float>(x, gamma, beta, residual, y, mean, rstd, B, HW, C, eps, relu, ws, ws_bytes, stream);
}
int dwc_instnorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                     const float* beta, float* dx, float* dgamma, float* dbeta, int B, int HW, int C, int relu, void* ws,
                     size_t ws_bytes, void* stream) {
    return instnorm_bwd_t<float>(dy, x, mean, rstd, gamma, beta, dx, dgamma, dbeta, B, HW, C, relu, ws, ws_bytes, stream);
}
int dwc_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* inv, int B, int HW,
                      int C, float eps, int relu, void* ws, size_t ws_bytes, void* stream) {
    return layernorm_fwd_t<float>(x, gamma, beta, y, mean, inv, B, HW, C, eps, relu, ws, ws_bytes, stream);
}
int dwc_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* inv, const float* gamma,
                      const float* beta, float* dx, float* dgamma, float* dbeta, int B, int HW, int C, float eps, int relu,
                      void* ws, size_t ws_bytes, void* stream) {
    return layernorm_bwd_t<float>(dy, x, mean, inv, gamma, beta, dx, dgamma, dbeta, B, HW, C, eps, relu, ws, ws_bytes, stream);
}

/* bf16 activations (x, residual, y, dy, dx); statistics, gamma/beta and their gradients stay fp32 */
int dwc_bf16_instnorm_fwd(const void* x, const float* gamma, const float* beta, const void* residual, void* y, float* mean,
                          float* rstd, int B, int HW, int C, float eps, int relu, void* ws, size_t ws_bytes, void* stream) {
    return instnorm_fwd_t<dwc_bf16>((const dwc_bf16*)x, gamma, beta, (const dwc_bf16*)residual, (dwc_bf16*)y, mean, rstd, B, HW, C,
                                    eps, relu, ws, ws_bytes, stream);
}
int dwc_bf16_instnorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                          const float* beta, void* dx, float* dgamma, float* dbeta, int B, int HW, int C, int relu, void* ws,
                          size_t ws_bytes, void* stream) {
    return instnorm_bwd_t<dwc_bf16>((const dwc_bf16*)dy, (const dwc_bf16*)x, mean, rstd, gamma, beta, (dwc_bf16*)dx, dgamma, dbeta,
                                    B, HW, C, relu, ws, ws_bytes, stream);
}
int dwc_bf16_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* inv, int B,
                           int HW, int C, float eps, int relu, void* ws, size_t ws_bytes, void* stream) {
    return layernorm_fwd_t<dwc_bf16>((const dwc_bf16*)x, gamma, beta, (dwc_bf16*)y, mean, inv, B, HW, C, eps, relu, ws, ws_bytes,
                                     stream);
}
int dwc_bf16_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* inv, const float* gamma,
                           const float* beta, void* dx, float* dgamma, float* dbeta, int B, int HW, int C, float eps, int relu,
                           void* ws, size_t ws_bytes, void* stream) {
    return layernorm_bwd_t<dwc_bf16>((const dwc_bf16*)dy, (const dwc_bf16*)x, mean, inv, gamma, beta, (dwc_bf16*)dx, dgamma, dbeta,
                                     B, HW, C, eps, relu, ws, ws_bytes, stream);
}

}  // extern "C"
