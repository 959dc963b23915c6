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

#include <atomic>

#include <type_traits>

#include "dwc_common.h"

namespace {

// rows-per-sample are split into `chunks`; geometry shared by the statistics kernels (one partial per chunk)
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

// The apply passes carry no partials, so they split finer: a 256-thread block = (256 / cq) row groups x cq column groups
// (cq = C / V) walks up to 32 rows per thread; grid (row chunks, B).  The per-(n,c) statistics and affine parameters are
// loaded ONCE per thread (they used to be re-fetched, behind a 64-bit division, for every 4 elements).
RowSplit plan_apply(int B, int HW, int C, int V) {
    const int cq = C / V > 0 ? C / V : 1;
    const int groups = 256 / cq > 0 ? 256 / cq : 1;
    int rpc = groups * 32;                                                       // ~32 rows per thread: the per-thread statistics
                                                                                 // loads (up to 6 x 32 bytes) are paid once per block
    while (rpc > groups * 4 && (long)B * ((HW + rpc - 1) / rpc) < 1024) rpc /= 2;   // keep >= ~4 blocks per CU
    while ((long)B * ((HW + rpc - 1) / rpc) > 16384 && rpc < HW) rpc *= 2;     // bound the grid on huge tensors
    RowSplit r;
    r.rows_per_chunk = rpc < HW ? rpc : HW;
    r.chunks = (HW + r.rows_per_chunk - 1) / r.rows_per_chunk;
    return r;
}

// sum of `groups` row-group partials held in LDS (fixed order: bitwise reproducible), for the thread with rg == 0
template <int V>
__device__ __forceinline__ void colsum_groups(float (&a)[V], float (&b)[V], float* sm, int groups, int cq, int col, int rg) {
    // layout sm[plane][t][V] with t = rg * cq + col
    float* s0 = sm;
    float* s1 = sm + 256 * V;
#pragma unroll
    for (int k = 0; k < V; ++k) {
        s0[threadIdx.x * V + k] = a[k];
        s1[threadIdx.x * V + k] = b[k];
    }
    __syncthreads();
    if (rg == 0) {
        for (int g = 1; g < groups; ++g) {
#pragma unroll
            for (int k = 0; k < V; ++k) {
                a[k] += s0[(g * cq + col) * V + k];
                b[k] += s1[(g * cq + col) * V + k];
            }
        }
    }
}

// (Rounds 3-4 finalised the statistics inside the partial launch at small batches -- write-through partials, a ticket per sample, the
// last-arriving workgroup of a sample summing them -- to save the *_final launch.  r05 measured what that costs at batch 16 on the
// planes the resident kernels do not take: 2 048 agent-scope acquire/release tickets per launch, in_stats_partial 71.9 us fused
// against 13.0 + 5.2 us as two launches once the final kernel is parallel over the chunks (in_stats_final_wide), in_bwd_partial 89.0
// against 21.4 + 4.8.  The mechanism was removed, and with ABI 8 the `tickets` argument of the entry points.)

// ---------------------------------------------------------------------------------------
// instance norm
// ---------------------------------------------------------------------------------------
// partial[(n*chunks+chunk)*C + c] = sum (x-pivot), second plane = sum (x-pivot)^2
template <typename T>
__global__ __launch_bounds__(256) void in_stats_partial(const T* __restrict__ x, float* __restrict__ part, int HW, int C,
                                                        int rows_per_chunk, size_t plane) {
    constexpr int V = VecOf<T>::V;
    __shared__ float sm[2 * 256 * V];
    const int cq = C / V;
    const int groups = 256 / cq;
    const int col = threadIdx.x % cq, rg = threadIdx.x / cq;
    const int n = blockIdx.y, chunk = blockIdx.x;
    const T* xs = x + (size_t)n * HW * C;
    float piv[V];
    ldv(xs, col, piv);
    const int r0 = chunk * rows_per_chunk;
    const int r1 = min(HW, r0 + rows_per_chunk);
    float s1[V], s2[V];
#pragma unroll
    for (int k = 0; k < V; ++k) s1[k] = s2[k] = 0.f;
    if (rg < groups) {
        int r = r0 + rg;
        for (; r + 3 * groups < r1; r += 4 * groups) {           // four rows in flight per thread
            float v[4][V];
#pragma unroll
            for (int u = 0; u < 4; ++u) ldv(xs, (size_t)(r + u * groups) * cq + col, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int k = 0; k < V; ++k) {
                    const float d = v[u][k] - piv[k];
                    s1[k] += d;
                    s2[k] += d * d;
                }
        }
        for (; r < r1; r += groups) {
            float v[V];
            ldv(xs, (size_t)r * cq + col, v);
#pragma unroll
            for (int k = 0; k < V; ++k) {
                const float d = v[k] - piv[k];
                s1[k] += d;
                s2[k] += d * d;
            }
        }
    }
    colsum_groups<V>(s1, s2, sm, groups, cq, col, rg);
    if (rg == 0) {
        const size_t o = ((size_t)(n * gridDim.x + chunk) * C) + col * V;
        stf<V>(part, o, s1);
        stf<V>(part + plane, o, s2);
    }
}

// Sum of the per-chunk partials of one (sample, channel): 64 channels x 4 chunk lanes per workgroup, chunk lane q takes chunks q, q + 4, ...
// with eight loads per plane in flight, the four lane sums are added in lane order (fixed order: bitwise reproducible).  r05: the
// one-thread-per-channel loops of in_stats_final / in_bwd_final took 17-33 us at batch 16 (64-128 chunks, one dependent load after
// the other), and the fused finalisation in the last-arriving workgroup of a sample 20-60 us more than that (2 048 agent-scope
// acquire/release tickets per launch): in_stats_partial 71.9 us fused against 12.6 + 33.1 unfused at 16 x 128 x 128 x 64.
__device__ __forceinline__ void norm_chunk_lanes(const float* __restrict__ p, size_t plane, int C, int chunks, int q, float& a, float& b) {
    int k = q;
    for (; k + 28 < chunks; k += 32) {
        float va[8], vb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            va[u] = p[(size_t)(k + 4 * u) * C];
            vb[u] = p[plane + (size_t)(k + 4 * u) * C];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) a += va[u], b += vb[u];
    }
    for (; k < chunks; k += 4) {
        a += p[(size_t)k * C];
        b += p[plane + (size_t)k * C];
    }
}

// grid (ceil(C / 64), B), 256 threads
template <typename T>
__global__ __launch_bounds__(256) void in_stats_final_wide(const T* __restrict__ x, const float* __restrict__ part, float* __restrict__ mean,
                                                           float* __restrict__ rstd, int HW, int C, int chunks, size_t plane, float eps) {
    __shared__ float sm[2][4][64];
    const int n = blockIdx.y, cl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float a = 0.f, b = 0.f;
    if (c < C) norm_chunk_lanes(part + (size_t)n * chunks * C + c, plane, C, chunks, q, a, b);
    sm[0][q][cl] = a;
    sm[1][q][cl] = b;
    __syncthreads();
    if (q != 0 || c >= C) return;
    const float s1 = ((sm[0][0][cl] + sm[0][1][cl]) + sm[0][2][cl]) + sm[0][3][cl];
    const float s2 = ((sm[1][0][cl] + sm[1][1][cl]) + sm[1][2][cl]) + sm[1][3][cl];
    const float inv = 1.f / (float)HW;
    const float d = s1 * inv;
    const float var = fmaxf(s2 * inv - d * d, 0.f);
    mean[(size_t)n * C + c] = (float)x[(size_t)n * HW * C + c] + d;
    rstd[(size_t)n * C + c] = 1.f / sqrtf(var + eps);
}

__global__ __launch_bounds__(256) void in_bwd_final_wide(const float* __restrict__ part, float* __restrict__ sums, float* __restrict__ dgamma,
                                                         float* __restrict__ dbeta, int BC, int C, int chunks, size_t plane) {
    __shared__ float sm[2][4][64];
    const int n = blockIdx.y, cl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float a = 0.f, b = 0.f;
    if (c < C) norm_chunk_lanes(part + (size_t)n * chunks * C + c, plane, C, chunks, q, a, b);
    sm[0][q][cl] = a;
    sm[1][q][cl] = b;
    __syncthreads();
    if (q != 0 || c >= C) return;
    const float s1 = ((sm[0][0][cl] + sm[0][1][cl]) + sm[0][2][cl]) + sm[0][3][cl];
    const float s2 = ((sm[1][0][cl] + sm[1][1][cl]) + sm[1][2][cl]) + sm[1][3][cl];
    const size_t idx = (size_t)n * C + c;
    sums[idx] = s1;
    sums[BC + idx] = s2;
    if (dgamma) dgamma[idx] = s2;
    if (dbeta) dbeta[idx] = s1;
}

template <typename T>
__global__ __launch_bounds__(256) void in_apply(const T* __restrict__ x, const float* __restrict__ mean,
                                                const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                const float* __restrict__ beta, const T* __restrict__ residual,
                                                T* __restrict__ y, int HW, int C, int rows_per_chunk, int relu,
                                                unsigned long long* amax = nullptr, unsigned amax_ep = 0) {
    constexpr int V = VecOf<T>::V;
    const int cq = C / V;
    const int groups = 256 / cq;
    const int col = threadIdx.x % cq, rg = threadIdx.x / cq;
    // (threads past the last row group -- none while groups * cq == 256 -- walk no rows below but still reach the publish barrier)
    unsigned am = 0;                                       // largest |y| this thread writes (two-plane conv kernels: absmax slot of y)
    const int n = blockIdx.y;
    const size_t s = (size_t)n * C + col * V;
    float mu[V], sc[V], sh[V];
    ldf<V>(mean, s, mu);
    ldf<V>(rstd, s, sc);
    if (gamma) {
        float ga[V];
        ldf<V>(gamma, s, ga);
#pragma unroll
        for (int k = 0; k < V; ++k) sc[k] *= ga[k];
    }
#pragma unroll
    for (int k = 0; k < V; ++k) sh[k] = 0.f;
    if (beta) ldf<V>(beta, s, sh);
    const size_t base = (size_t)n * HW * cq + col;
    const int r0 = blockIdx.x * rows_per_chunk;
    const int r1 = rg < groups ? min(HW, r0 + rows_per_chunk) : r0;
    auto one = [&](const float (&v)[V], const float* res, float (&o)[V]) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float t = (v[k] - mu[k]) * sc[k] + sh[k];
            if (relu) t = t < 0.f ? 0.f : t;              // NaN-preserving
            o[k] = res ? t + res[k] : t;
        }
        if constexpr (std::is_same<T, float>::value) am = dwc_amax_fold<V>(am, o);
    };
    int r = r0 + rg;
    for (; r + 3 * groups < r1; r += 4 * groups) {        // four rows in flight per thread
        float v[4][V], q[4][V], o[4][V];
#pragma unroll
        for (int u = 0; u < 4; ++u) ldv(x, base + (size_t)(r + u * groups) * cq, v[u]);
        if (residual) {
#pragma unroll
            for (int u = 0; u < 4; ++u) ldv(residual, base + (size_t)(r + u * groups) * cq, q[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            one(v[u], residual ? q[u] : nullptr, o[u]);
            stv(y, base + (size_t)(r + u * groups) * cq, o[u]);
        }
    }
    for (; r < r1; r += groups) {
        const size_t i0 = base + (size_t)r * cq;
        float v0[V], q0[V], o0[V];
        ldv(x, i0, v0);
        if (residual) ldv(residual, i0, q0);
        one(v0, residual ? q0 : nullptr, o0);
        stv(y, i0, o0);
    }
    if constexpr (std::is_same<T, float>::value) {      // (every thread of the block gets here: C is a power of two, groups * cq == 256)
        __shared__ unsigned s_am[4];
        dwc_amax_block_publish(amax, amax_ep, am, s_am);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void in_bwd_partial(const T* __restrict__ dy, const T* __restrict__ x,
                                                      const float* __restrict__ mean, const float* __restrict__ rstd,
                                                      const float* __restrict__ gamma, const float* __restrict__ beta,
                                                      float* __restrict__ part, int HW, int C, int rows_per_chunk, size_t plane,
                                                      int relu) {
    constexpr int V = VecOf<T>::V;
    __shared__ float sm[2 * 256 * V];
    const int cq = C / V;
    const int groups = 256 / cq;
    const int col = threadIdx.x % cq, rg = threadIdx.x / cq;
    const int n = blockIdx.y, chunk = blockIdx.x;
    const size_t base = (size_t)n * HW * cq + col;
    const size_t s = (size_t)n * C + col * V;
    float mu[V], rs[V], ga[V], be[V];
    ldf<V>(mean, s, mu);
    ldf<V>(rstd, s, rs);
#pragma unroll
    for (int k = 0; k < V; ++k) ga[k] = 1.f, be[k] = 0.f;
    if (gamma) ldf<V>(gamma, s, ga);
    if (beta) ldf<V>(beta, s, be);
    const int r0 = chunk * rows_per_chunk;
    const int r1 = min(HW, r0 + rows_per_chunk);
    float s1[V], s2[V];
#pragma unroll
    for (int k = 0; k < V; ++k) s1[k] = s2[k] = 0.f;
    auto acc = [&](const float (&xv)[V], const float (&dv)[V]) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float xh = (xv[k] - mu[k]) * rs[k];
            float g = dv[k];
            // (the forward's own expression and rounding, (x - mu) * (gamma * rstd) + beta: an element within an ulp of zero must get the
            // mask its forward output got)
            if (relu) g = ((xv[k] - mu[k]) * (ga[k] * rs[k]) + be[k]) > 0.f ? g : 0.f;
            s1[k] += g;
            s2[k] += g * xh;
        }
    };
    if (rg < groups) {
        int r = r0 + rg;
        for (; r + groups < r1; r += 2 * groups) {
            const size_t i0 = base + (size_t)r * cq, i1 = i0 + (size_t)groups * cq;
            float x0[V], x1[V], d0[V], d1[V];
            ldv(x, i0, x0);
            ldv(x, i1, x1);
            ldv(dy, i0, d0);
            ldv(dy, i1, d1);
            acc(x0, d0);
            acc(x1, d1);
        }
        for (; r < r1; r += groups) {
            float x0[V], d0[V];
            ldv(x, base + (size_t)r * cq, x0);
            ldv(dy, base + (size_t)r * cq, d0);
            acc(x0, d0);
        }
    }
    colsum_groups<V>(s1, s2, sm, groups, cq, col, rg);
    if (rg == 0) {
        const size_t o = ((size_t)(n * gridDim.x + chunk) * C) + col * V;
        stf<V>(part, o, s1);
        stf<V>(part + plane, o, s2);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void in_bwd_apply(const T* __restrict__ dy, const T* __restrict__ x,
                                                    const float* __restrict__ mean, const float* __restrict__ rstd,
                                                    const float* __restrict__ gamma, const float* __restrict__ beta,
                                                    const float* __restrict__ sums, T* __restrict__ dx, int HW, int C, int BC,
                                                    int rows_per_chunk, int relu, unsigned long long* amax = nullptr,
                                                    unsigned amax_ep = 0) {
    constexpr int V = VecOf<T>::V;
    const int cq = C / V;
    const int groups = 256 / cq;
    const int col = threadIdx.x % cq, rg = threadIdx.x / cq;
    // (threads past the last row group -- none while groups * cq == 256 -- walk no rows below but still reach the publish barrier)
    unsigned am = 0;
    const int n = blockIdx.y;
    const size_t s = (size_t)n * C + col * V;
    const float inv_hw = 1.f / (float)HW;
    float mu[V], rs[V], ga[V], be[V], k1[V], k2[V];
    ldf<V>(mean, s, mu);
    ldf<V>(rstd, s, rs);
#pragma unroll
    for (int k = 0; k < V; ++k) ga[k] = 1.f, be[k] = 0.f;
    if (gamma) ldf<V>(gamma, s, ga);
    if (beta) ldf<V>(beta, s, be);
    ldf<V>(sums, s, k1);
    ldf<V>(sums + BC, s, k2);
#pragma unroll
    for (int k = 0; k < V; ++k) k1[k] *= inv_hw, k2[k] *= inv_hw;
    const size_t base = (size_t)n * HW * cq + col;
    const int r0 = blockIdx.x * rows_per_chunk;
    const int r1 = rg < groups ? min(HW, r0 + rows_per_chunk) : r0;
    auto one = [&](const float (&xv)[V], const float (&dv)[V], float (&o)[V]) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float xh = (xv[k] - mu[k]) * rs[k];
            float g = dv[k];
            // (the forward's own expression and rounding, (x - mu) * (gamma * rstd) + beta: an element within an ulp of zero must get the
            // mask its forward output got)
            if (relu) g = ((xv[k] - mu[k]) * (ga[k] * rs[k]) + be[k]) > 0.f ? g : 0.f;
            o[k] = ga[k] * rs[k] * (g - k1[k] - xh * k2[k]);
        }
        if constexpr (std::is_same<T, float>::value) am = dwc_amax_fold<V>(am, o);
    };
    int r = r0 + rg;
    for (; r + 3 * groups < r1; r += 4 * groups) {        // four rows of x and dy in flight per thread
        float xv[4][V], dv[4][V], o[4][V];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ldv(x, base + (size_t)(r + u * groups) * cq, xv[u]);
            ldv(dy, base + (size_t)(r + u * groups) * cq, dv[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            one(xv[u], dv[u], o[u]);
            stv(dx, base + (size_t)(r + u * groups) * cq, o[u]);
        }
    }
    for (; r < r1; r += groups) {
        const size_t i0 = base + (size_t)r * cq;
        float x0[V], d0[V], o0[V];
        ldv(x, i0, x0);
        ldv(dy, i0, d0);
        one(x0, d0, o0);
        stv(dx, i0, o0);
    }    if constexpr (std::is_same<T, float>::value) {      // (every thread of the block gets here: C is a power of two, groups * cq == 256)
        __shared__ unsigned s_am[4];
        dwc_amax_block_publish(amax, amax_ep, am, s_am);
    }
}

// ---------------------------------------------------------------------------------------
// MUNIT layer norm (statistics per sample over C*HW)
// ---------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void ln_stats_partial(const T* __restrict__ x, float* __restrict__ part, int HW, int C,
                                                        int rows_per_chunk) {
    constexpr int V = VecOf<T>::V;
    __shared__ float sm[4];
    const int cq = C / V;
    const int n = blockIdx.y, chunk = blockIdx.x;
    const T* xs = x + (size_t)n * HW * C;
    const float piv = (float)x[(size_t)n * HW * C];
    const size_t e0 = (size_t)chunk * rows_per_chunk * cq;
    const size_t e1 = min((size_t)HW * cq, e0 + (size_t)rows_per_chunk * cq);
    float s1 = 0.f, s2 = 0.f;
    auto acc = [&](const float (&v)[V]) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float d = v[k] - piv;
            a += d;
            b += d * d;
        }
        s1 += a;
        s2 += b;
    };
    size_t e = e0 + threadIdx.x;
    for (; e + 768 < e1; e += 1024) {
        float v[4][V];
#pragma unroll
        for (int u = 0; u < 4; ++u) ldv(xs, e + 256 * u, v[u]);
#pragma unroll
        for (int u = 0; u < 4; ++u) acc(v[u]);
    }
    for (; e < e1; e += 256) {
        float v[V];
        ldv(xs, e, v);
        acc(v);
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
    int k = 0;
    for (; k + 8 <= chunks; k += 8) {                      // (eight pairs in flight, summed in chunk order)
        float va[8], vb[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            va[u] = part[(size_t)(n * chunks + k + u) * 2];
            vb[u] = part[(size_t)(n * chunks + k + u) * 2 + 1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s1 += va[u], s2 += vb[u];
    }
    for (; k < chunks; ++k) {
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
                                                const float* __restrict__ beta, T* __restrict__ y, int HW, int C, int rows_per_chunk,
                                                int relu, unsigned long long* amax = nullptr, unsigned amax_ep = 0) {
    constexpr int V = VecOf<T>::V;
    const int cq = C / V;
    const int groups = 256 / cq;
    const int col = threadIdx.x % cq, rg = threadIdx.x / cq;
    // (threads past the last row group -- none while groups * cq == 256 -- walk no rows below but still reach the publish barrier)
    unsigned am = 0;
    const int n = blockIdx.y;
    const float mu = mean[n], iv = inv[n];
    float sc[V], sh[V];
    ldf<V>(gamma, (size_t)col * V, sc);
    ldf<V>(beta, (size_t)col * V, sh);
#pragma unroll
    for (int k = 0; k < V; ++k) sc[k] *= iv;
    const size_t base = (size_t)n * HW * cq + col;
    const int r0 = blockIdx.x * rows_per_chunk;
    const int r1 = rg < groups ? min(HW, r0 + rows_per_chunk) : r0;
    auto one = [&](const float (&v)[V], float (&o)[V]) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float t = (v[k] - mu) * sc[k] + sh[k];
            o[k] = (relu && t < 0.f) ? 0.f : t;          // NaN-preserving
        }
        if constexpr (std::is_same<T, float>::value) am = dwc_amax_fold<V>(am, o);
    };
    int r = r0 + rg;
    for (; r + groups < r1; r += 2 * groups) {
        const size_t i0 = base + (size_t)r * cq, i1 = i0 + (size_t)groups * cq;
        float v0[V], v1[V], o0[V], o1[V];
        ldv(x, i0, v0);
        ldv(x, i1, v1);
        one(v0, o0);
        one(v1, o1);
        stv(y, i0, o0);
        stv(y, i1, o1);
    }
    for (; r < r1; r += groups) {
        float v0[V], o0[V];
        ldv(x, base + (size_t)r * cq, v0);
        one(v0, o0);
        stv(y, base + (size_t)r * cq, o0);
    }
    if constexpr (std::is_same<T, float>::value) {      // (every thread of the block gets here: C is a power of two, groups * cq == 256)
        __shared__ unsigned s_am[4];
        dwc_amax_block_publish(amax, amax_ep, am, s_am);
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
    constexpr int V = VecOf<T>::V;
    __shared__ float sm[2 * 256 * V];
    __shared__ float sr[4];
    const int cq = C / V;
    const int groups = 256 / cq;
    const int col = threadIdx.x % cq, rg = threadIdx.x / cq;
    const int n = blockIdx.y, chunk = blockIdx.x;
    const size_t base = (size_t)n * HW * cq + col;
    const float mu = mean[n], iv = inv[n];
    float ga[V], be[V];
    ldf<V>(gamma, (size_t)col * V, ga);
    ldf<V>(beta, (size_t)col * V, be);
    const int r0 = chunk * rows_per_chunk;
    const int r1 = min(HW, r0 + rows_per_chunk);
    float dg[V], db[V];
#pragma unroll
    for (int k = 0; k < V; ++k) dg[k] = db[k] = 0.f;
    float t1 = 0.f, t2 = 0.f;
    auto acc = [&](const float (&xv)[V], const float (&dv)[V]) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float xc = xv[k] - mu;
            const float xh = xc * iv;
            float d = dv[k];
            if (relu) d = (xc * (ga[k] * iv) + be[k]) > 0.f ? d : 0.f;      // (the forward's expression and rounding: ln_apply)
            dg[k] += d * xh;
            db[k] += d;
            const float g = d * ga[k];
            a += g;
            b += g * xc;
        }
        t1 += a;
        t2 += b;
    };
    if (rg < groups) {
        int r = r0 + rg;
        for (; r + groups < r1; r += 2 * groups) {
            const size_t i0 = base + (size_t)r * cq, i1 = i0 + (size_t)groups * cq;
            float x0[V], x1[V], d0[V], d1[V];
            ldv(x, i0, x0);
            ldv(x, i1, x1);
            ldv(dy, i0, d0);
            ldv(dy, i1, d1);
            acc(x0, d0);
            acc(x1, d1);
        }
        for (; r < r1; r += groups) {
            float x0[V], d0[V];
            ldv(x, base + (size_t)r * cq, x0);
            ldv(dy, base + (size_t)r * cq, d0);
            acc(x0, d0);
        }
    }
    colsum_groups<V>(dg, db, sm, groups, cq, col, rg);
    const size_t blk = (size_t)n * gridDim.x + chunk;
    if (rg == 0) {
        stf<V>(part_c, blk * 2 * C + col * V, dg);
        stf<V>(part_c, blk * 2 * C + C + col * V, db);
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
    if (c < C) {
        // (r05: eight loads per sum in flight -- the plain loop was 126 dependent loads at batch 48, 35-47 us per launch)
        const int total = B * chunks;
        int k = g;
        for (; k + 7 * 16 < total; k += 8 * 16) {
            float va[8], vb[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                va[u] = part_c[(size_t)(k + 16 * u) * 2 * C + c];
                vb[u] = part_c[(size_t)(k + 16 * u) * 2 * C + C + c];
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) a += va[u], b += vb[u];
        }
        for (; k < total; k += 16) {
            a += part_c[(size_t)k * 2 * C + c];
            b += part_c[(size_t)k * 2 * C + C + c];
        }
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
                                                    int rows_per_chunk, float eps, int relu, unsigned long long* amax = nullptr,
                                                    unsigned amax_ep = 0) {
    constexpr int V = VecOf<T>::V;
    const int cq = C / V;
    const int groups = 256 / cq;
    const int col = threadIdx.x % cq, rg = threadIdx.x / cq;
    // (threads past the last row group -- none while groups * cq == 256 -- walk no rows below but still reach the publish barrier)
    unsigned am = 0;
    const int n = blockIdx.y;
    const float N = (float)HW * (float)C;
    const float mu = mean[n], iv = inv[n];
    const float sigma = 1.f / iv - eps;
    const float mean_g = sums[n * 2] / N;
    // d/dx of 1/(sigma+eps): -(1/(sigma+eps))^2 * (x-mu)/((N-1)*sigma), times sum g*(x-mu)
    const float k2 = sigma > 0.f ? iv * iv * sums[n * 2 + 1] / ((N - 1.f) * sigma) : 0.f;
    float ga[V], be[V];
    ldf<V>(gamma, (size_t)col * V, ga);
    ldf<V>(beta, (size_t)col * V, be);
    const size_t base = (size_t)n * HW * cq + col;
    const int r0 = blockIdx.x * rows_per_chunk;
    const int r1 = rg < groups ? min(HW, r0 + rows_per_chunk) : r0;
    auto one = [&](const float (&xv)[V], const float (&dv)[V], float (&o)[V]) {
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float xc = xv[k] - mu;
            float d = dv[k];
            if (relu) d = (xc * (ga[k] * iv) + be[k]) > 0.f ? d : 0.f;      // (the forward's expression and rounding: ln_apply)
            o[k] = (d * ga[k] - mean_g) * iv - xc * k2;
        }
        if constexpr (std::is_same<T, float>::value) am = dwc_amax_fold<V>(am, o);
    };
    int r = r0 + rg;
    for (; r + 3 * groups < r1; r += 4 * groups) {        // four rows of x and dy in flight per thread
        float xv[4][V], dv[4][V], o[4][V];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ldv(x, base + (size_t)(r + u * groups) * cq, xv[u]);
            ldv(dy, base + (size_t)(r + u * groups) * cq, dv[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            one(xv[u], dv[u], o[u]);
            stv(dx, base + (size_t)(r + u * groups) * cq, o[u]);
        }
    }
    for (; r < r1; r += groups) {
        const size_t i0 = base + (size_t)r * cq;
        float x0[V], d0[V], o0[V];
        ldv(x, i0, x0);
        ldv(dy, i0, d0);
        one(x0, d0, o0);
        stv(dx, i0, o0);
    }    if constexpr (std::is_same<T, float>::value) {      // (every thread of the block gets here: C is a power of two, groups * cq == 256)
        __shared__ unsigned s_am[4];
        dwc_amax_block_publish(amax, amax_ep, am, s_am);
    }
}

bool norm_shape_ok(int B, int HW, int C, int V) {      // C a power of two with 1 <= C / V <= 256 column groups
    const int l = dwc_ilog2_exact(C);
    return B > 0 && HW > 0 && l >= 2 && C >= V && C <= 1024;
}

}  // namespace

namespace {

// ---------------------------------------------------------------------------------------
// Resident-plane instance norm (r04): ONE pass over HBM per direction for the small planes (HW = 256 / 1024: every ResBlock
// norm of the content encoder and the decoder, 17 of the 19 IN / AdaIN layers).  A 256-thread workgroup owns sample n and
// 64 contiguous bytes of every pixel (16 fp32 / 32 bf16 channels; the two halves of a 128-byte line go to the same XCD, res_block) and keeps its part of the plane in REGISTERS, raw, four
// channels per lane and pixel (16-byte fp32 / 8-byte bf16 pieces: with eight bf16 channels per lane the per-channel
// statistics and parameters of the backward pass no longer fit beside the plane): statistics, then normalise / gradient
// from the same registers -- forward reads x (+ residual) and writes y: 2-3 tensor passes instead of 3-4 (statistics pass +
// apply pass); backward reads dy, x and writes dx: 3 instead of 5; one launch instead of two or three.  The variance is the
// exact two-pass form (the data are resident).  Cross-thread sums: xor-shuffles over the row groups of a wave, then the 8
// wave partials through LDS, summed by every thread in wave order (fixed order: bitwise reproducible).
// ---------------------------------------------------------------------------------------
typedef unsigned res_u32x2 __attribute__((ext_vector_type(2)));
template <typename T> struct Raw4 { typedef f32x4 type; static constexpr int CQ = 4; };         // pieces per pixel and workgroup (64 bytes)
template <> struct Raw4<dwc_bf16> { typedef res_u32x2 type; static constexpr int CQ = 8; };
__device__ __forceinline__ void res_unpack(const f32x4& r, float (&o)[4]) { o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3]; }
__device__ __forceinline__ void res_unpack(const res_u32x2& r0, float (&o)[4]) {
    // (the empty asm makes the piece a NEW value at every use: otherwise hipcc converts every resident piece to fp32 once and
    // keeps the floats -- twice the registers of the raw plane)
    res_u32x2 r = r0;
    asm volatile("" : "+v"(r));
    o[0] = __uint_as_float(r[0] << 16);
    o[1] = __uint_as_float(r[0] & 0xffff0000u);
    o[2] = __uint_as_float(r[1] << 16);
    o[3] = __uint_as_float(r[1] & 0xffff0000u);
}
__device__ __forceinline__ void res_pack(const float (&o)[4], f32x4& r) { r = f32x4{o[0], o[1], o[2], o[3]}; }
__device__ __forceinline__ void res_pack(const float (&o)[4], res_u32x2& r) {
    typedef dwc_bf16 bf16x2r __attribute__((ext_vector_type(2)));
    bf16x2r a = {(dwc_bf16)o[0], (dwc_bf16)o[1]}, b = {(dwc_bf16)o[2], (dwc_bf16)o[3]};      // round to nearest even, as every bf16 store here
    r[0] = __builtin_bit_cast(unsigned, a);
    r[1] = __builtin_bit_cast(unsigned, b);
}

// sum of v[0..NV) over the row groups of the workgroup, for this thread's piece column; every thread gets the totals
template <int CQ, int NV, int THREADS>
__device__ __forceinline__ void res_reduce(float (&v)[NV], float* sm) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cq = threadIdx.x & (CQ - 1);
#pragma unroll
    for (int k = 0; k < NV; ++k) {
        if (CQ <= 4) v[k] += __shfl_xor(v[k], 4);
        if (CQ <= 8) v[k] += __shfl_xor(v[k], 8);
        v[k] += __shfl_xor(v[k], 16);
        v[k] += __shfl_xor(v[k], 32);
    }
    __syncthreads();                                       // the previous reduction's partials have been read
    if (lane < CQ) {
#pragma unroll
        for (int k = 0; k < NV; ++k) sm[(wave * CQ + cq) * NV + k] = v[k];
    }
    __syncthreads();
    if constexpr ((THREADS / 64) * NV > 32) {
        // Many partials (the backward's 8 sums of 8+ waves): every thread summing all of them keeps waves x NV LDS loads in flight
        // beside the resident plane -- 14-16 spilled registers in in_resident_bwd.  One thread per column sums (same order), the
        // totals are read back after one more barrier.
        if (threadIdx.x < CQ * NV) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < THREADS / 64; ++w) t += sm[w * CQ * NV + threadIdx.x];
            sm[threadIdx.x] = t;                           // (slot (wave 0, column) is read by this thread only)
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NV; ++k) v[k] = sm[cq * NV + k];
    } else {
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < THREADS / 64; ++w) t += sm[(w * CQ + cq) * NV + k];
            v[k] = t;
        }
    }
}

// Workgroup -> (sample n, 64-byte channel group grp).  Two groups share every 128-byte line of the plane: they are given to
// workgroups id and id + 8 -- the same XCD (workgroups go round-robin over the 8 XCDs), dispatched together -- so that the line
// comes from HBM once and is found in that XCD's L2 by the second.  `pairs` = B * groups / 2; the grid has ceil(pairs / 8) * 16
// workgroups, the surplus ones return false.
__device__ __forceinline__ bool res_block(int groups, int pairs, int& n, int& grp) {
    const int id = blockIdx.x, xcd = id & 7, k = id >> 3;
    const int pair = (k >> 1) * 8 + xcd;
    if (pair >= pairs) return false;
    const int gp = groups >> 1;
    n = pair / gp;
    grp = 2 * (pair - n * gp) + (k & 1);
    return true;
}

// THREADS: 256 keeps HW/ROWS = 16 (fp32) / 32 (bf16) pieces per thread and tensor, 512 half of that -- chosen so that a kernel
// stays near 128 registers (>= 4 waves per SIMD: the load, reduce and store phases of several workgroups overlap on a CU).
template <typename T, int HW, int THREADS, bool HAS_RES>
__global__ __launch_bounds__(THREADS) void in_resident_fwd(const T* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, const T* __restrict__ residual,
                                                               T* __restrict__ y, float* __restrict__ mean, float* __restrict__ rstd,
                                                               int C, float eps, int relu, int pairs, unsigned long long* amax = nullptr,
                                                               unsigned amax_ep = 0) {
    constexpr int V = 4, CQ = Raw4<T>::CQ, ROWS = THREADS / CQ, NP = HW / ROWS;
    typedef typename Raw4<T>::type Raw;
    __shared__ float sm[(THREADS / 64) * CQ * 2 * V];
    const int cqt = C / V;                                                       // 4-channel pieces per pixel of the tensor
    int n, grp;
    if (!res_block(C / (CQ * V), pairs, n, grp)) return;
    const int cq = threadIdx.x & (CQ - 1), rg = threadIdx.x / CQ;
    // addressing: workgroup-uniform base (+ a uniform stride per pass) and ONE 32-bit byte offset per thread, so that the 2-3 x NP
    // loads / stores share a single address register
    const size_t ubase = ((size_t)n * HW * cqt + grp * CQ) * sizeof(Raw);
    const unsigned toff = (unsigned)((rg * cqt + cq) * sizeof(Raw));
    const size_t pstride = (size_t)ROWS * cqt * sizeof(Raw);
    const char* xb = reinterpret_cast<const char*>(x) + ubase;
    Raw xr[NP], rr[HAS_RES ? NP : 1];
#pragma unroll
    for (int p = 0; p < NP; ++p) xr[p] = *reinterpret_cast<const Raw*>(xb + p * pstride + toff);
    if constexpr (HAS_RES) {
        const char* rb = reinterpret_cast<const char*>(residual) + ubase;
#pragma unroll
        for (int p = 0; p < NP; ++p) rr[p] = *reinterpret_cast<const Raw*>(rb + p * pstride + toff);
    }
    const size_t sidx = (size_t)n * C + (grp * CQ + cq) * V;                     // first channel of this thread's piece
    float sc[V], sh[V];
#pragma unroll
    for (int k = 0; k < V; ++k) sc[k] = 1.f, sh[k] = 0.f;
    if (gamma) ldf<V>(gamma, sidx, sc);
    if (beta) ldf<V>(beta, sidx, sh);
    float mu[V];
#pragma unroll
    for (int k = 0; k < V; ++k) mu[k] = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        float v[V];
        res_unpack(xr[p], v);
#pragma unroll
        for (int k = 0; k < V; ++k) mu[k] += v[k];
    }
    res_reduce<CQ, V, THREADS>(mu, sm);
    const float inv = 1.f / (float)HW;
    float var[V];
#pragma unroll
    for (int k = 0; k < V; ++k) mu[k] *= inv, var[k] = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        float v[V];
        res_unpack(xr[p], v);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float d = v[k] - mu[k];
            var[k] += d * d;
        }
    }
    res_reduce<CQ, V, THREADS>(var, sm);
    float rs[V];
#pragma unroll
    for (int k = 0; k < V; ++k) rs[k] = 1.f / sqrtf(var[k] * inv + eps);
    if (rg == 0) {
        stf<V>(mean, sidx, mu);
        stf<V>(rstd, sidx, rs);
    }
#pragma unroll
    for (int k = 0; k < V; ++k) sc[k] *= rs[k];
    char* yb = reinterpret_cast<char*>(y) + ubase;
    unsigned am = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        float v[V], q[V], o[V];
        res_unpack(xr[p], v);
        if constexpr (HAS_RES) res_unpack(rr[HAS_RES ? p : 0], q);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            float t = (v[k] - mu[k]) * sc[k] + sh[k];
            if (relu) t = t < 0.f ? 0.f : t;              // NaN-preserving
            o[k] = HAS_RES ? t + q[k] : t;
        }
        if constexpr (std::is_same<T, float>::value) am = dwc_amax_fold<V>(am, o);
        Raw out;
        res_pack(o, out);
        *reinterpret_cast<Raw*>(yb + p * pstride + toff) = out;
    }
    if constexpr (std::is_same<T, float>::value) {
        __shared__ unsigned s_am[THREADS / 64];
        dwc_amax_block_publish(amax, amax_ep, am, s_am);
    }
}

template <typename T, int HW, int THREADS>
__global__ __launch_bounds__(THREADS, THREADS >= 512 ? 4 : 1) void in_resident_bwd(const T* __restrict__ dy, const T* __restrict__ x,
                                                               const float* __restrict__ mean, const float* __restrict__ rstd,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               T* __restrict__ dx, float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                               int C, int relu, int pairs, unsigned long long* amax = nullptr,
                                                               unsigned amax_ep = 0) {
    constexpr int V = 4, CQ = Raw4<T>::CQ, ROWS = THREADS / CQ, NP = HW / ROWS;
    typedef typename Raw4<T>::type Raw;
    __shared__ float sm[(THREADS / 64) * CQ * 2 * V];
    const int cqt = C / V;
    int n, grp;
    if (!res_block(C / (CQ * V), pairs, n, grp)) return;
    const int cq = threadIdx.x & (CQ - 1), rg = threadIdx.x / CQ;
    const size_t ubase = ((size_t)n * HW * cqt + grp * CQ) * sizeof(Raw);
    const unsigned toff = (unsigned)((rg * cqt + cq) * sizeof(Raw));
    const size_t pstride = (size_t)ROWS * cqt * sizeof(Raw);
    const char* xb = reinterpret_cast<const char*>(x) + ubase;
    const char* gb = reinterpret_cast<const char*>(dy) + ubase;
    Raw xr[NP], gr[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        xr[p] = *reinterpret_cast<const Raw*>(xb + p * pstride + toff);
        gr[p] = *reinterpret_cast<const Raw*>(gb + p * pstride + toff);
    }
    const size_t sidx = (size_t)n * C + (grp * CQ + cq) * V;
    float mu[V], rs[V], ga[V], be[V];
    ldf<V>(mean, sidx, mu);
    ldf<V>(rstd, sidx, rs);
#pragma unroll
    for (int k = 0; k < V; ++k) ga[k] = 1.f, be[k] = 0.f;
    if (gamma) ldf<V>(gamma, sidx, ga);
    if (beta) ldf<V>(beta, sidx, be);
    float s[2 * V];
#pragma unroll
    for (int k = 0; k < 2 * V; ++k) s[k] = 0.f;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        float xv[V], dv[V];
        res_unpack(xr[p], xv);
        res_unpack(gr[p], dv);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float xh = (xv[k] - mu[k]) * rs[k];
            float g = dv[k];
            // (the forward's own expression and rounding, (x - mu) * (gamma * rstd) + beta: an element within an ulp of zero must get the
            // mask its forward output got)
            if (relu) g = ((xv[k] - mu[k]) * (ga[k] * rs[k]) + be[k]) > 0.f ? g : 0.f;
            s[k] += g;
            s[V + k] += g * xh;
        }
    }
    res_reduce<CQ, 2 * V, THREADS>(s, sm);
    if (rg == 0 && dgamma) {
        float a[V], b[V];
#pragma unroll
        for (int k = 0; k < V; ++k) a[k] = s[k], b[k] = s[V + k];
        stf<V>(dbeta, sidx, a);
        stf<V>(dgamma, sidx, b);
    }
    const float inv = 1.f / (float)HW;
    char* ob = reinterpret_cast<char*>(dx) + ubase;
    unsigned am = 0;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        float xv[V], dv[V], o[V];
        res_unpack(xr[p], xv);
        res_unpack(gr[p], dv);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            const float xh = (xv[k] - mu[k]) * rs[k];
            float g = dv[k];
            // (the forward's own expression and rounding, (x - mu) * (gamma * rstd) + beta: an element within an ulp of zero must get the
            // mask its forward output got)
            if (relu) g = ((xv[k] - mu[k]) * (ga[k] * rs[k]) + be[k]) > 0.f ? g : 0.f;
            o[k] = ga[k] * rs[k] * (g - s[k] * inv - xh * (s[V + k] * inv));
        }
        if constexpr (std::is_same<T, float>::value) am = dwc_amax_fold<V>(am, o);
        Raw out;
        res_pack(o, out);
        *reinterpret_cast<Raw*>(ob + p * pstride + toff) = out;
    }
    if constexpr (std::is_same<T, float>::value) {
        __shared__ unsigned s_am[THREADS / 64];
        dwc_amax_block_publish(amax, amax_ep, am, s_am);
    }
}

// plane size when the resident-plane kernels take this shape, else 0
template <typename T>
int in_resident_hw(int HW, int C) {
    static const bool on = !(getenv("DWC_NORM_RESIDENT") && atoi(getenv("DWC_NORM_RESIDENT")) == 0);      // development: 0 = the multi-pass kernels
    if (!on || C % (2 * Raw4<T>::CQ * 4)) return 0;            // an even number of 64-byte groups
    return (HW == 1024 || HW == 256) ? HW : 0;
}

// The caller's ticket row for one statistics launch, or null when the batch is too large for the fused finalisation to pay
// (measured r03: worth it at small batches -- c1, B = 16..48: 0.80 -> 0.71 ms and 0.86 -> 0.79 ms of statistics kernels per step plus
// 92 launch boundaries; at B >= 128 the per-workgroup publish + ticket costs more than the *_final launch it saves).

size_t instnorm_ws_bytes(int B, int HW, int C) {
    const RowSplit rs = plan_rows(B, HW);
    return ((size_t)2 * B * rs.chunks * C + (size_t)2 * B * C) * sizeof(float);
}

template <typename T>
int instnorm_fwd_t(const T* x, const float* gamma, const float* beta, const T* residual, T* y, float* mean,
                     float* rstd, int B, int HW, int C, float eps, int relu, void* ws, size_t ws_bytes, void* stream,
                     unsigned long long* amax = nullptr, unsigned amax_ep = 0) {
    if (!norm_shape_ok(B, HW, C, VecOf<T>::V)) return DWC_EINVAL;
    if (!ws || ws_bytes < instnorm_ws_bytes(B, HW, C)) return DWC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (const int rhw = in_resident_hw<T>(HW, C)) {          // small planes: one pass, plane resident in registers
        const int pairs = B * (C / (Raw4<T>::CQ * 4)) / 2;
        const dim3 grid((pairs + 7) / 8 * 16);
        // (r05, bf16 planes without a residual, in-step: 256 threads 58.8 us, 512 threads 54.8, 1 024 threads 64.5; with a residual 512
        // threads 81.8 against 89.0 on 1 024; fp32 without a residual: 256 threads 16.1 us, 512 threads 16.7.  Whole 128-byte lines per
        // workgroup (64 bf16 channels, 1 024 threads, no pairing of workgroups over a line): 52.2 against 54.9 us -- not what holds these
        // kernels at 2.4 TB/s; not kept)
        if (rhw == 1024 && !residual && sizeof(T) == 2) hipLaunchKernelGGL((in_resident_fwd<T, 1024, 512, false>), grid, dim3(512), 0, st, x, gamma, beta, residual, y, mean, rstd, C, eps, relu, pairs, amax, amax_ep);
        else if (rhw == 1024 && residual) hipLaunchKernelGGL((in_resident_fwd<T, 1024, 512, true>), grid, dim3(512), 0, st, x, gamma, beta, residual, y, mean, rstd, C, eps, relu, pairs, amax, amax_ep);
        else if (rhw == 1024) hipLaunchKernelGGL((in_resident_fwd<T, 1024, 256, false>), grid, dim3(256), 0, st, x, gamma, beta, residual, y, mean, rstd, C, eps, relu, pairs, amax, amax_ep);
        else if (residual) hipLaunchKernelGGL((in_resident_fwd<T, 256, 256, true>), grid, dim3(256), 0, st, x, gamma, beta, residual, y, mean, rstd, C, eps, relu, pairs, amax, amax_ep);
        else hipLaunchKernelGGL((in_resident_fwd<T, 256, 256, false>), grid, dim3(256), 0, st, x, gamma, beta, residual, y, mean, rstd, C, eps, relu, pairs, amax, amax_ep);
        DWC_LAUNCH_CHECK();
        return DWC_OK;
    }
    const RowSplit rs = plan_rows(B, HW);
    const size_t plane = (size_t)B * rs.chunks * C;
    float* part = (float*)ws;
    hipLaunchKernelGGL(in_stats_partial<T>, dim3(rs.chunks, B), dim3(256), 0, st, x, part, HW, C, rs.rows_per_chunk, plane);
    DWC_LAUNCH_CHECK();
    hipLaunchKernelGGL(in_stats_final_wide<T>, dim3((C + 63) / 64, B), dim3(256), 0, st, x, part, mean, rstd, HW, C, rs.chunks, plane, eps);
    DWC_LAUNCH_CHECK();
    const RowSplit ra = plan_apply(B, HW, C, VecOf<T>::V);
    hipLaunchKernelGGL(in_apply<T>, dim3(ra.chunks, B), dim3(256), 0, st, x, mean, rstd, gamma, beta, residual, y, HW, C,
                       ra.rows_per_chunk, relu, amax, amax_ep);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

template <typename T>
int instnorm_bwd_t(const T* dy, const T* x, const float* mean, const float* rstd, const float* gamma,
                   const float* beta, T* dx, float* dgamma, float* dbeta, int B, int HW, int C, int relu, void* ws,
                     size_t ws_bytes, void* stream, unsigned long long* amax = nullptr, unsigned amax_ep = 0) {
    if (!norm_shape_ok(B, HW, C, VecOf<T>::V)) return DWC_EINVAL;
    if (!ws || ws_bytes < instnorm_ws_bytes(B, HW, C)) return DWC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (const int rhw = in_resident_hw<T>(HW, C)) {
        const int pairs = B * (C / (Raw4<T>::CQ * 4)) / 2;
        const dim3 grid((pairs + 7) / 8 * 16);
        // (r05, per-kernel time inside the bench step: bf16 127.5 us with 512 threads and 14-16 spilled registers, 115.8 with the
        // two-stage reduction alone, 98.2 with 1024 threads -- 8 pieces per thread and tensor, no spills; fp32 31.2 -> 25.4 with the
        // reduction alone, 26.1 with 1024 threads)
        constexpr bool wide = sizeof(T) == 2;
        if (rhw == 1024 && wide) hipLaunchKernelGGL((in_resident_bwd<T, 1024, 1024>), grid, dim3(1024), 0, st, dy, x, mean, rstd, gamma, beta, dx, dgamma, dbeta, C, relu, pairs, amax, amax_ep);
        else if (rhw == 1024) hipLaunchKernelGGL((in_resident_bwd<T, 1024, 512>), grid, dim3(512), 0, st, dy, x, mean, rstd, gamma, beta, dx, dgamma, dbeta, C, relu, pairs, amax, amax_ep);
        else hipLaunchKernelGGL((in_resident_bwd<T, 256, 256>), grid, dim3(256), 0, st, dy, x, mean, rstd, gamma, beta, dx, dgamma, dbeta, C, relu, pairs, amax, amax_ep);
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
    hipLaunchKernelGGL(in_bwd_final_wide, dim3((C + 63) / 64, B), dim3(256), 0, st, part, sums, dgamma, dbeta, B * C, C, rs.chunks, plane);
    DWC_LAUNCH_CHECK();
    const RowSplit ra = plan_apply(B, HW, C, VecOf<T>::V);
    hipLaunchKernelGGL(in_bwd_apply<T>, dim3(ra.chunks, B), dim3(256), 0, st, dy, x, mean, rstd, gamma, beta, sums, dx, HW, C,
                       B * C, ra.rows_per_chunk, relu, amax, amax_ep);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

size_t layernorm_ws_bytes(int B, int HW, int C) {
    const RowSplit rs = plan_rows(B, HW);
    return ((size_t)B * rs.chunks * 2 + (size_t)B * rs.chunks * 2 * C + (size_t)2 * B + 16) * sizeof(float);
}

template <typename T>
int layernorm_fwd_t(const T* x, const float* gamma, const float* beta, T* y, float* mean, float* inv, int B, int HW,
                      int C, float eps, int relu, void* ws, size_t ws_bytes, void* stream, unsigned long long* amax = nullptr,
                      unsigned amax_ep = 0) {
    if (!norm_shape_ok(B, HW, C, VecOf<T>::V) || (size_t)HW * C < 2) return DWC_EINVAL;
    if (!ws || ws_bytes < layernorm_ws_bytes(B, HW, C)) return DWC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const RowSplit rs = plan_rows(B, HW);
    float* part = (float*)ws;
    hipLaunchKernelGGL(ln_stats_partial<T>, dim3(rs.chunks, B), dim3(256), 0, st, x, part, HW, C, rs.rows_per_chunk);
    DWC_LAUNCH_CHECK();
    hipLaunchKernelGGL(ln_stats_final<T>, dim3((B + 63) / 64), dim3(64), 0, st, x, part, mean, inv, B, HW, C, rs.chunks, eps);
    DWC_LAUNCH_CHECK();
    const RowSplit ra = plan_apply(B, HW, C, VecOf<T>::V);
    hipLaunchKernelGGL(ln_apply<T>, dim3(ra.chunks, B), dim3(256), 0, st, x, mean, inv, gamma, beta, y, HW, C, ra.rows_per_chunk, relu,
                       amax, amax_ep);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

template <typename T>
int layernorm_bwd_t(const T* dy, const T* x, const float* mean, const float* inv, const float* gamma,
                    const float* beta, T* dx, float* dgamma, float* dbeta, int B, int HW, int C, float eps, int relu,
                      void* ws, size_t ws_bytes, void* stream, unsigned long long* amax = nullptr, unsigned amax_ep = 0) {
    if (!norm_shape_ok(B, HW, C, VecOf<T>::V)) return DWC_EINVAL;
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
    const RowSplit ra = plan_apply(B, HW, C, VecOf<T>::V);
    hipLaunchKernelGGL(ln_bwd_apply<T>, dim3(ra.chunks, B), dim3(256), 0, st, dy, x, mean, inv, gamma, beta, sums, dx, HW, C,
                       ra.rows_per_chunk, eps, relu, amax, amax_ep);
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

/* The same four with the absmax slot of the tensor they WRITE (y / dx): raised by atomic max on (epoch << 32 | magnitude bits) from
 * the apply pass, so that a two-plane split-product convolution (dwc_h2_*, include/dwcgan_hip.h) that consumes the tensor needs no
 * dwc_absmax pass of its own.  out_amax NULL: exactly the plain entry point. */
int dwc_instnorm_fwd_amax(const float* x, const float* gamma, const float* beta, const float* residual, float* y, float* mean,
                          float* rstd, int B, int HW, int C, float eps, int relu, void* ws, size_t ws_bytes,
                          void* out_amax, unsigned out_epoch, void* stream) {
    return instnorm_fwd_t<float>(x, gamma, beta, residual, y, mean, rstd, B, HW, C, eps, relu, ws, ws_bytes, stream,
                                 (unsigned long long*)out_amax, out_epoch);
}
int dwc_instnorm_bwd_amax(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                          const float* beta, float* dx, float* dgamma, float* dbeta, int B, int HW, int C, int relu, void* ws,
                          size_t ws_bytes, void* out_amax, unsigned out_epoch, void* stream) {
    return instnorm_bwd_t<float>(dy, x, mean, rstd, gamma, beta, dx, dgamma, dbeta, B, HW, C, relu, ws, ws_bytes, stream,
                                 (unsigned long long*)out_amax, out_epoch);
}
int dwc_layernorm_fwd_amax(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* inv, int B, int HW,
                           int C, float eps, int relu, void* ws, size_t ws_bytes, void* out_amax, unsigned out_epoch, void* stream) {
    return layernorm_fwd_t<float>(x, gamma, beta, y, mean, inv, B, HW, C, eps, relu, ws, ws_bytes, stream, (unsigned long long*)out_amax,
                                  out_epoch);
}
int dwc_layernorm_bwd_amax(const float* dy, const float* x, const float* mean, const float* inv, const float* gamma,
                           const float* beta, float* dx, float* dgamma, float* dbeta, int B, int HW, int C, float eps, int relu,
                           void* ws, size_t ws_bytes, void* out_amax, unsigned out_epoch, void* stream) {
    return layernorm_bwd_t<float>(dy, x, mean, inv, gamma, beta, dx, dgamma, dbeta, B, HW, C, eps, relu, ws, ws_bytes, stream,
                                  (unsigned long long*)out_amax, out_epoch);
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
