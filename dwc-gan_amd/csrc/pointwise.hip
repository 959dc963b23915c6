// HBM-bound pointwise / reduction kernels around the conv stack (gfx950, fp32, NHWC).
//   * activation backward + bias gradient      (autograd of reference networks.py:583-584, conv bias)
//   * bilinear x2 upsample / 2x2 mean          (reference networks_v2.py:154, networks.py:113)
//   * NCHW(3) <-> NHWC4 image boundary
//   * attention blend                           (reference solver.py:148,161,170,179-180,192,330-331)
//   * mean |a-b|                                (reference solver.py:113-114)
//   * Adam (+coupled L2) and the EMA lerp       (reference solver.py:62-68, utils.py:52-54)
// All use 16-byte accesses with the channel axis on the lanes and grid-stride loops.
#include <type_traits>

#include "dwc_common.h"

namespace {

int grid_for(size_t items, int cap = 4096) {
    size_t b = (items + 255) / 256;
    if (b > (size_t)cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

// ---- activation backward + bias gradient ------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void act_bwd_partial(const T* __restrict__ dy, const T* __restrict__ y,
                                                       T* __restrict__ g, float* __restrict__ part, int rows, int C,
                                                       int rows_per_chunk, int act, unsigned long long* amax = nullptr,
                                                       unsigned amax_ep = 0) {
    constexpr int V = VecOf<T>::V;                 // 16-byte accesses: 4 fp32 / 8 bf16 per lane
    __shared__ float sm[256 * V];
    const int cq = C / V;
    const int cqb = cq < 256 ? cq : 256;           // column groups handled by this block
    const int groups = 256 / cqb;
    const int col = blockIdx.y * 256 + threadIdx.x % cqb, rg = threadIdx.x / cqb;
    const int r0 = blockIdx.x * rows_per_chunk;
    const int r1 = min(rows, r0 + rows_per_chunk);
    float s[V];
    unsigned am = 0;                                // largest |g| this thread writes (absmax slot of g, see dwc_amax_wave_publish)
#pragma unroll
    for (int k = 0; k < V; ++k) s[k] = 0.f;
    auto one = [&](float (&d)[V], const float (&yy)[V]) {
        if (act != DWC_ACT_NONE) {
#pragma unroll
            for (int k = 0; k < V; ++k) d[k] *= dwc_act_grad(yy[k], act, col * V + k);
        }
#pragma unroll
        for (int k = 0; k < V; ++k) s[k] += d[k];
        if constexpr (std::is_same<T, float>::value) am = dwc_amax_fold<V>(am, d);
    };
    int r = r0 + rg;
    for (; r + groups < r1; r += 2 * groups) {      // two rows in flight per thread
        const size_t i0 = (size_t)r * cq + col, i1 = i0 + (size_t)groups * cq;
        float d0[V], d1[V], y0[V], y1[V];
        ldv(dy, i0, d0);
        ldv(dy, i1, d1);
        if (act != DWC_ACT_NONE) {
            ldv(y, i0, y0);
            ldv(y, i1, y1);
        }
        one(d0, y0);
        one(d1, y1);
        if (g) {
            stv(g, i0, d0);
            stv(g, i1, d1);
        }
    }
    for (; r < r1; r += groups) {
        const size_t i0 = (size_t)r * cq + col;
        float d0[V], y0[V];
        ldv(dy, i0, d0);
        if (act != DWC_ACT_NONE) ldv(y, i0, y0);
        one(d0, y0);
        if (g) stv(g, i0, d0);
    }
    if constexpr (std::is_same<T, float>::value) {
        __shared__ unsigned s_am[4];
        dwc_amax_block_publish(amax, amax_ep, am, s_am);
    }
    if (!part) return;
#pragma unroll
    for (int k = 0; k < V; ++k) sm[threadIdx.x * V + k] = s[k];
    __syncthreads();
    if (rg == 0) {
        for (int q = 1; q < groups; ++q)
#pragma unroll
            for (int k = 0; k < V; ++k) s[k] += sm[(q * cqb + threadIdx.x % cqb) * V + k];
        stf<V>(part, (size_t)blockIdx.x * C + col * V, s);
    }
}

// out[c] = sum_k part[k][c]; 64 channels x 16 chunk-lanes per block, fixed summation order
__global__ __launch_bounds__(1024) void colsum_final(const float* __restrict__ part, float* __restrict__ out, int chunks, int C) {
    __shared__ float sm[16][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (c < C) {
        int k = g;
        for (; k + 7 * 16 < chunks; k += 8 * 16) {      // eight loads in flight (the walk over up to 64 partials per thread was all latency)
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(k + 16 * u) * C + c];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < chunks; k += 16) s += part[(size_t)k * C + c];
    }
    sm[g][cl] = s;
    __syncthreads();
    if (g == 0 && c < C) {
#pragma unroll
        for (int k = 1; k < 16; ++k) s += sm[k][cl];
        out[c] = s;
    }
}

void act_plan(int rows, int* chunks, int* rpc) {
    int c = rows / 32;
    if (c > 1024) c = 1024;
    if (c < 1) c = 1;
    *rpc = (rows + c - 1) / c;
    *chunks = (rows + *rpc - 1) / *rpc;
}

size_t act_bwd_ws(int rows, int C) {
    int chunks, rpc;
    act_plan(rows, &chunks, &rpc);
    return (size_t)chunks * C * sizeof(float);
}

// ---- bilinear x2 (align_corners=False), torch's tap rule ------------------------------------
__device__ __forceinline__ void up_taps(int o, int n_in, int& i0, int& i1, float& l0, float& l1) {
    float src = ((float)o + 0.5f) * 0.5f - 0.5f;
    src = src < 0.f ? 0.f : src;
    i0 = (int)src;
    i1 = i0 + (i0 < n_in - 1 ? 1 : 0);
    l1 = src - (float)i0;
    l0 = 1.f - l1;
}

// One thread = one INPUT pixel column x V channels over UP_ROWS input rows (grid (W*cq / 256, ceil(H / UP_ROWS), B)): per input row it
// reads the pixel and its two horizontal neighbours (the neighbours are the next waves' own pixels: L1), forms the two horizontal
// blends, and writes the 2x2 output quad from the blends of rows i-1, i, i+1 kept in registers -- 3 loads per 4 stores, all 16-byte
// accesses, no per-element divisions.  (r05: the one-row form read the whole 3x3 neighbourhood per thread, 9 loads per 4 stores:
// bf16 128x32x32x256 205.7 us = 1.6 TB/s of algorithmic bytes, fp32 16x... 44.5 us.)
// Output row 2i reads rows (i-1: 0.25, i: 0.75), row 2i+1 reads (i: 0.75, i+1: 0.25); at the borders the missing neighbour is
// the pixel itself (torch clamps the source coordinate), which the clamped index reproduces exactly.
constexpr int UP_ROWS = 8;
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int H, int W, int cq) {
    constexpr int V = VecOf<T>::V;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= W * cq) return;
    const int ix = idx / cq, c = idx - ix * cq;
    const int iy0 = blockIdx.y * UP_ROWS;
    const size_t n = blockIdx.z;
    const int xm = max(ix - 1, 0), xp = min(ix + 1, W - 1);
    const size_t b = n * H * W;
    // horizontal blends of input row r: h[0] = 0.25*left + 0.75*mid, h[1] = 0.75*mid + 0.25*right
    auto blend = [&](int r, float (&h)[2][V]) {
        float l[V], m[V], q[V];
        const size_t rb = (b + (size_t)r * W) * cq + c;
        ldv(x, rb + (size_t)xm * cq, l);
        ldv(x, rb + (size_t)ix * cq, m);
        ldv(x, rb + (size_t)xp * cq, q);
#pragma unroll
        for (int k = 0; k < V; ++k) {
            h[0][k] = 0.75f * m[k] + 0.25f * l[k];
            h[1][k] = 0.75f * m[k] + 0.25f * q[k];
        }
    };
    float hm[2][V], hc[2][V], hp[2][V];
    blend(max(iy0 - 1, 0), hm);
    blend(iy0, hc);
#pragma unroll
    for (int r = 0; r < UP_ROWS; ++r) {
        const int iy = iy0 + r;
        if (iy >= H) break;
        blend(min(iy + 1, H - 1), hp);
        const size_t ob = (n * (2 * H) + 2 * iy) * (size_t)(2 * W) + 2 * ix;
#pragma unroll
        for (int d = 0; d < 2; ++d) {
            float o0[V], o1[V];
#pragma unroll
            for (int k = 0; k < V; ++k) {
                o0[k] = 0.75f * hc[d][k] + 0.25f * hm[d][k];      // output row 2*iy
                o1[k] = 0.75f * hc[d][k] + 0.25f * hp[d][k];      // output row 2*iy + 1
            }
            stv(y, (ob + d) * cq + c, o0);
            stv(y, (ob + (size_t)(2 * W) + d) * cq + c, o1);
        }
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int k = 0; k < V; ++k) hm[d][k] = hc[d][k], hc[d][k] = hp[d][k];
    }
}

// adjoint, gather form: input pixel i collects from output rows 2i-1 (0.25), 2i, 2i+1 (0.75 each), 2i+2 (0.25); rows outside the
// image do not exist and the clamped border taps fold onto rows 0 / 2H-1 (weight 1 there instead of 0.75).  Same walk as the
// forward: the horizontal gathers (4 loads) of output rows 2i+1 and 2i+2 are new per input row, those of 2i-1 and 2i are the
// previous row's -- 8 loads per store instead of 16.
template <typename T>
__global__ __launch_bounds__(256) void upsample2x_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int H, int W, int cq) {
    constexpr int V = VecOf<T>::V;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= W * cq) return;
    const int ix = idx / cq, c = idx - ix * cq;
    const int iy0 = blockIdx.y * UP_ROWS;
    const size_t n = blockIdx.z;
    float wx[4];
    wx[0] = ix > 0 ? 0.25f : 0.f;
    wx[1] = ix > 0 ? 0.75f : 1.f;
    wx[2] = ix < W - 1 ? 0.75f : 1.f;
    wx[3] = ix < W - 1 ? 0.25f : 0.f;
    int ox[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) ox[q] = min(max(2 * ix - 1 + q, 0), 2 * W - 1);
    const size_t base = n * (size_t)(2 * H) * (2 * W);
    auto gather = [&](int oy, float (&row)[V]) {                     // clamped rows carry weight 0
        const size_t rb = (base + (size_t)min(max(oy, 0), 2 * H - 1) * (2 * W)) * cq + c;
#pragma unroll
        for (int k = 0; k < V; ++k) row[k] = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[V];
            ldv(dy, rb + (size_t)ox[q] * cq, v);
#pragma unroll
            for (int k = 0; k < V; ++k) row[k] += wx[q] * v[k];
        }
    };
    float ha[V], hb[V], hc[V], hd[V];
    gather(2 * iy0 - 1, ha);
    gather(2 * iy0, hb);
#pragma unroll
    for (int r = 0; r < UP_ROWS; ++r) {
        const int iy = iy0 + r;
        if (iy >= H) break;
        gather(2 * iy + 1, hc);
        gather(2 * iy + 2, hd);
        const float wy0 = iy > 0 ? 0.25f : 0.f, wy1 = iy > 0 ? 0.75f : 1.f, wy2 = iy < H - 1 ? 0.75f : 1.f, wy3 = iy < H - 1 ? 0.25f : 0.f;
        float s[V];
#pragma unroll
        for (int k = 0; k < V; ++k) {
            s[k] = 0.f;
            s[k] += wy0 * ha[k];
            s[k] += wy1 * hb[k];
            s[k] += wy2 * hc[k];
            s[k] += wy3 * hd[k];
        }
        stv(dx, ((n * H + iy) * (size_t)W + ix) * cq + c, s);
#pragma unroll
        for (int k = 0; k < V; ++k) ha[k] = hc[k], hb[k] = hd[k];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int H, int W, int cq,
                                                           size_t total4) {
    const T* xs = x;
    const int Ho = H / 2, Wo = W / 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const int c = i % cq;
        size_t r = i / cq;
        const int ox = r % Wo;
        r /= Wo;
        const int oy = r % Ho;
        const size_t n = r / Ho;
        const size_t b = (n * H + 2 * oy) * W + 2 * ox;
        st4(y, i, ((ld4(xs, b * cq + c) + ld4(xs, (b + 1) * cq + c)) + (ld4(xs, (b + W) * cq + c) + ld4(xs, (b + W + 1) * cq + c))) * 0.25f);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void avgpool2_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int H, int W, int cq,
                                                           size_t total4) {
    const T* ds = dy;
    const int Ho = H / 2, Wo = W / 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const int c = i % cq;
        size_t r = i / cq;
        const int ix = r % W;
        r /= W;
        const int iy = r % H;
        const size_t n = r / H;
        st4(dx, i, ld4(ds, ((n * Ho + iy / 2) * Wo + ix / 2) * cq + c) * 0.25f);
    }
}

// ---- 2x2 max pooling (VGG16, reference networks.py:666,671,677) --------------------------------------
// Ties go to the first element in window scan order, as torch's max_pool2d does (after a ReLU all-zero windows tie).
template <typename T>
__global__ __launch_bounds__(256) void maxpool2_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int H, int W, int cq,
                                                           size_t total4) {
    const int Ho = H / 2, Wo = W / 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const int c = i % cq;
        size_t r = i / cq;
        const int ox = r % Wo;
        r /= Wo;
        const int oy = r % Ho;
        const size_t n = r / Ho;
        const size_t b = (n * H + 2 * oy) * W + 2 * ox;
        const f32x4 v0 = ld4(x, b * cq + c), v1 = ld4(x, (b + 1) * cq + c), v2 = ld4(x, (b + W) * cq + c), v3 = ld4(x, (b + W + 1) * cq + c);
        f32x4 m;
#pragma unroll
        for (int k = 0; k < 4; ++k) m[k] = fmaxf(fmaxf(v0[k], v1[k]), fmaxf(v2[k], v3[k]));
        st4(y, i, m);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool2_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                           T* __restrict__ dx, int H, int W, int cq, size_t total4) {
    const int Ho = H / 2, Wo = W / 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        const int c = i % cq;               // one thread per pooled element: it owns its whole 2x2 window of dx
        size_t r = i / cq;
        const int ox = r % Wo;
        r /= Wo;
        const int oy = r % Ho;
        const size_t n = r / Ho;
        const size_t b = (n * H + 2 * oy) * W + 2 * ox;
        const size_t at[4] = {b * cq + c, (b + 1) * cq + c, (b + W) * cq + c, (b + W + 1) * cq + c};
        f32x4 v[4], g[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = ld4(x, at[j]);
            g[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const f32x4 d = ld4(dy, i);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            int best = 0;
#pragma unroll
            for (int j = 1; j < 4; ++j)
                if (v[j][k] > v[best][k]) best = j;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j == best) g[j][k] = d[k];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) st4(dx, at[j], g[j]);
    }
}

// ---- image boundary ----------------------------------------------------------------------------
__global__ void pack_nhwc4_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int HW, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;  // pixel index over B*HW
    if (i >= total) return;
    const size_t n = i / HW, p = i % HW;
    f32x4 v = {0, 0, 0, 0};
    for (int c = 0; c < C; ++c) v[c] = x[(n * C + c) * HW + p];
    reinterpret_cast<f32x4*>(y)[i] = v;
}

__global__ void unpack_nhwc4_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int HW, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t n = i / HW, p = i % HW;
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    for (int c = 0; c < C; ++c) y[(n * C + c) * HW + p] = v[c];
}

// NCHW(<=3) fp32 -> NHWC8 bf16 (planes C..7 zero) and back: the bf16 path's image layout (16 bytes per pixel)
__global__ void pack_nhwc8_kernel(const float* __restrict__ x, dwc_bf16* __restrict__ y, int C, int HW, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t n = i / HW, p = i % HW;
    dwc_bf16x8 v;
#pragma unroll
    for (int c = 0; c < 8; ++c) v[c] = (dwc_bf16)0.f;
    for (int c = 0; c < C; ++c) v[c] = (dwc_bf16)x[(n * C + c) * HW + p];
    reinterpret_cast<dwc_bf16x8*>(y)[i] = v;
}

__global__ void unpack_nhwc8_kernel(const dwc_bf16* __restrict__ x, float* __restrict__ y, int C, int HW, size_t total) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const size_t n = i / HW, p = i % HW;
    const dwc_bf16x8 v = reinterpret_cast<const dwc_bf16x8*>(x)[i];
    for (int c = 0; c < C; ++c) y[(n * C + c) * HW + p] = (float)v[c];
}

__global__ void blend8_fwd_kernel(const dwc_bf16* __restrict__ heads, const dwc_bf16* __restrict__ real, dwc_bf16* __restrict__ out,
                                  size_t npix) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const dwc_bf16x8 h = reinterpret_cast<const dwc_bf16x8*>(heads)[i];
    const dwc_bf16x8 r = reinterpret_cast<const dwc_bf16x8*>(real)[i];
    const float a = (float)h[3], na = 1.f - a;
    dwc_bf16x8 o;
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = (dwc_bf16)0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) o[c] = (dwc_bf16)((float)h[c] * a + (float)r[c] * na);
    reinterpret_cast<dwc_bf16x8*>(out)[i] = o;
}

__global__ void blend8_bwd_kernel(const dwc_bf16* __restrict__ dout, const dwc_bf16* __restrict__ heads,
                                  const dwc_bf16* __restrict__ real, dwc_bf16* __restrict__ dheads, size_t npix) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const dwc_bf16x8 d = reinterpret_cast<const dwc_bf16x8*>(dout)[i];
    const dwc_bf16x8 h = reinterpret_cast<const dwc_bf16x8*>(heads)[i];
    const dwc_bf16x8 r = reinterpret_cast<const dwc_bf16x8*>(real)[i];
    dwc_bf16x8 o;
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = (dwc_bf16)0.f;
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        o[c] = (dwc_bf16)((float)d[c] * (float)h[3]);
        s += (float)d[c] * (float)h[c] - (float)d[c] * (float)r[c];
    }
    o[3] = (dwc_bf16)s;
    reinterpret_cast<dwc_bf16x8*>(dheads)[i] = o;
}

// ---- attention blend -----------------------------------------------------------------------------
__global__ void blend_fwd_kernel(const float* __restrict__ heads, const float* __restrict__ real, float* __restrict__ out, size_t npix) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const f32x4 h = reinterpret_cast<const f32x4*>(heads)[i];
    const f32x4 r = reinterpret_cast<const f32x4*>(real)[i];
    const float a = h[3], na = 1.f - h[3];
    f32x4 o;
    o[0] = h[0] * a + r[0] * na;
    o[1] = h[1] * a + r[1] * na;
    o[2] = h[2] * a + r[2] * na;
    o[3] = 0.f;
    reinterpret_cast<f32x4*>(out)[i] = o;
}

__global__ void blend_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ heads, const float* __restrict__ real,
                                 float* __restrict__ dheads, size_t npix) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    const f32x4 d = reinterpret_cast<const f32x4*>(dout)[i];
    const f32x4 h = reinterpret_cast<const f32x4*>(heads)[i];
    const f32x4 r = reinterpret_cast<const f32x4*>(real)[i];
    f32x4 o;
    o[0] = d[0] * h[3];
    o[1] = d[1] * h[3];
    o[2] = d[2] * h[3];
    o[3] = (d[0] * h[0] - d[0] * r[0]) + (d[1] * h[1] - d[1] * r[1]) + (d[2] * h[2] - d[2] * r[2]);
    reinterpret_cast<f32x4*>(dheads)[i] = o;
}

// ---- mean |a-b| ------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void l1_partial_kernel(const T* __restrict__ a, const T* __restrict__ b, float* __restrict__ part,
                                                         size_t n4, size_t n, int skip4) {
    __shared__ float sm[4];
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const f32x4 d = ld4(a, i) - ld4(b, i);
        if (skip4 == 8 && (i & 1)) continue;               // planes 4..7 of an 8-plane image
        s += (fabsf(d[0]) + fabsf(d[1])) + (fabsf(d[2]) + (skip4 ? 0.f : fabsf(d[3])));
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) s += fabsf((float)a[i] - (float)b[i]);
    s = dwc_block_sum_256(s, sm);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__global__ __launch_bounds__(256) void l1_final_kernel(const float* __restrict__ part, float* __restrict__ out, int blocks, float inv_n) {
    __shared__ float sm[4];
    float s = 0.f;
    for (int i = threadIdx.x; i < blocks; i += 256) s += part[i];
    s = dwc_block_sum_256(s, sm);
    if (threadIdx.x == 0) out[0] = s * inv_n;
}

template <typename T>
__global__ void l1_bwd_kernel(const T* __restrict__ a, const T* __restrict__ b, const float* __restrict__ dout, T* __restrict__ da,
                              T* __restrict__ db, size_t n, float inv_n, int skip4) {
    const float sc = dout[0] * inv_n;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float d = (float)a[i] - (float)b[i];
        float g = d > 0.f ? sc : (d < 0.f ? -sc : 0.f);
        if (skip4 && (i & (skip4 - 1)) >= 3) g = 0.f;
        if (da) da[i] = (T)g;
        if (db) db[i] = (T)(-g);
    }
}

// ---- adversarial loss tail of ONE discriminator scale ----------------------------------------------------------------
// reference networks.py:116-170 (calc_dis_loss / calc_gen_loss, LSGAN + BCE-with-logits attribute classification): the batch is
// `segs` segments of B samples (the batched passes [x_fake | x_fake1 | x_real] of the D step, [x_fake | x_fake1] of the G step);
//   out = sum_s  w_src[s] * mean_s((src - target[s])^2)  +  w_cls[s] * mean_s(BCEwithLogits(cls, labels))
// in ONE single-workgroup launch (a few thousand elements) instead of ~25 stock elementwise / reduction launches per scale
// and direction.  Sums are taken in a fixed order (bitwise reproducible).  labels: [B][ncls], shared by the segments.
__global__ __launch_bounds__(256) void adv_tail_fwd_kernel(const float* __restrict__ src, const float* __restrict__ cls,
                                                           const float* __restrict__ labels, float* __restrict__ out, int segs, int B,
                                                           int sps, int ncls, dwc_adv_spec sp) {
    __shared__ float sm[4];
    float total = 0.f;
    for (int s = 0; s < segs; ++s) {
        float a = 0.f, c = 0.f;
        if (sp.w_src[s] != 0.f) {
            const float* p = src + (size_t)s * B * sps;
            for (int i = threadIdx.x; i < B * sps; i += 256) {
                const float d = p[i] - sp.target[s];
                a += d * d;
            }
        }
        if (sp.w_cls[s] != 0.f) {
            const float* z = cls + (size_t)s * B * ncls;
            for (int i = threadIdx.x; i < B * ncls; i += 256) {
                const float v = z[i], t = labels[i];
                c += fmaxf(v, 0.f) - v * t + log1pf(expf(-fabsf(v)));          // torch's stable binary_cross_entropy_with_logits
            }
        }
        a = dwc_block_sum_256(a, sm);
        c = dwc_block_sum_256(c, sm);
        total += sp.w_src[s] * (a / (float)(B * sps)) + sp.w_cls[s] * (c / (float)(B * ncls));
    }
    if (threadIdx.x == 0) out[0] = total;
}

__global__ __launch_bounds__(256) void adv_tail_bwd_kernel(const float* __restrict__ src, const float* __restrict__ cls,
                                                           const float* __restrict__ labels, const float* __restrict__ dout,
                                                           float* __restrict__ dsrc, float* __restrict__ dcls, int segs, int B, int sps,
                                                           int ncls, dwc_adv_spec sp) {
    const float g = dout[0];
    const int nsrc = segs * B * sps, ncl = segs * B * ncls;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < nsrc + ncl; i += gridDim.x * 256) {
        if (i < nsrc) {
            const int s = i / (B * sps);
            dsrc[i] = g * sp.w_src[s] * 2.f * (src[i] - sp.target[s]) / (float)(B * sps);
        } else {
            const int j = i - nsrc, s = j / (B * ncls), r = j - s * (B * ncls);
            const float v = cls[j];
            dcls[j] = g * sp.w_cls[s] * (1.f / (1.f + expf(-v)) - labels[r]) / (float)(B * ncls);
        }
    }
}

// ---- optimiser -------------------------------------------------------------------------------------
__device__ __forceinline__ float lerp_torch(float start, float end, float w) {
    // torch.lerp's two-branch form (ATen/native/Lerp.h)
    return w < 0.5f ? start + w * (end - start) : end - (end - start) * (1.f - w);
}

__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                            float step_size, float beta1, float beta2, float eps, float wd, float bc2_sqrt) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float pi = p[i];
        const float gi = g[i] + wd * pi;
        const float mi = lerp_torch(m[i], gi, 1.f - beta1);
        const float vi = v[i] * beta2 + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = pi - step_size * (mi / denom);
    }
}

__global__ void ema_kernel(const float* __restrict__ p, float* __restrict__ e, size_t n, float beta) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        e[i] = lerp_torch(p[i], e[i], beta);
}

// Multi-tensor forms: ONE launch walks every parameter tensor of a network.  Workgroup b handles
// DWC_OPT_CHUNK consecutive elements of tensor chunk_tensor[b] starting at chunk_start[b]; tensors
// whose gradient pointer is NULL are skipped entirely (torch.optim.Adam semantics for parameters
// that received no gradient this step: no weight decay, no moment update, no step count).
__global__ __launch_bounds__(256) void adam_multi_kernel(const dwc_adam_tensor* __restrict__ tensors, const int* __restrict__ chunk_tensor,
                                                         const unsigned* __restrict__ chunk_start, float omb1, float beta2, float omb2,
                                                         float eps, float wd) {
    const dwc_adam_tensor d = tensors[chunk_tensor[blockIdx.x]];
    if (!d.g) return;
    const size_t i0 = chunk_start[blockIdx.x];
    const size_t i1 = i0 + DWC_OPT_CHUNK < d.n ? i0 + DWC_OPT_CHUNK : d.n;
    for (size_t i = i0 + threadIdx.x; i < i1; i += 256) {
        const float pi = d.p[i];
        const float gi = d.g[i] + wd * pi;
        const float mi = lerp_torch(d.m[i], gi, omb1);
        const float vi = d.v[i] * beta2 + omb2 * gi * gi;
        d.m[i] = mi;
        d.v[i] = vi;
        const float denom = sqrtf(vi) / d.bc2_sqrt + eps;
        d.p[i] = pi - d.step_size * (mi / denom);
    }
}

__global__ __launch_bounds__(256) void ema_multi_kernel(const dwc_ema_tensor* __restrict__ tensors, const int* __restrict__ chunk_tensor,
                                                        const unsigned* __restrict__ chunk_start, float beta) {
    const dwc_ema_tensor d = tensors[chunk_tensor[blockIdx.x]];
    const size_t i0 = chunk_start[blockIdx.x];
    const size_t i1 = i0 + DWC_OPT_CHUNK < d.n ? i0 + DWC_OPT_CHUNK : d.n;
    for (size_t i = i0 + threadIdx.x; i < i1; i += 256) d.ema[i] = lerp_torch(d.p[i], d.ema[i], beta);
}

}  // namespace

extern "C" {

size_t dwc_act_bwd_bias_ws_bytes(int rows, int C) {
    int chunks, rpc;
    act_plan(rows, &chunks, &rpc);
    return (size_t)chunks * C * sizeof(float);
}

}  // extern "C"

namespace {

template <typename T>
int act_bwd_bias_t(const T* dy, const T* y, T* g, float* db, int rows, int C, int act, void* ws, size_t ws_bytes, void* stream,
                   unsigned long long* amax = nullptr, unsigned amax_ep = 0) {
    constexpr int V = VecOf<T>::V;
    if (rows <= 0 || C <= 0 || (C % V)) return DWC_EINVAL;
    const int cq = C / V;
    if (cq < 256 ? (256 % cq) != 0 : (cq % 256) != 0) return DWC_EINVAL;
    if (act != DWC_ACT_NONE && !y) return DWC_EINVAL;
    if (db && (!ws || ws_bytes < act_bwd_ws(rows, C))) return DWC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    int chunks, rpc;
    act_plan(rows, &chunks, &rpc);
    hipLaunchKernelGGL(act_bwd_partial<T>, dim3(chunks, (cq + 255) / 256), dim3(256), 0, st, dy, y, g, db ? (float*)ws : nullptr,
                       rows, C, rpc, act, g ? amax : nullptr, amax_ep);
    DWC_LAUNCH_CHECK();
    if (db) {
        hipLaunchKernelGGL(colsum_final, dim3((C + 63) / 64), dim3(1024), 0, st, (const float*)ws, db, chunks, C);
        DWC_LAUNCH_CHECK();
    }
    return DWC_OK;
}

template <typename T>
int upsample2x_fwd_t(const T* x, T* y, int B, int H, int W, int C, void* stream) {
    constexpr int V = VecOf<T>::V;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % V) || H > 65535 || B > 65535) return DWC_EINVAL;
    hipLaunchKernelGGL(upsample2x_fwd_kernel<T>, dim3((W * (C / V) + 255) / 256, (H + UP_ROWS - 1) / UP_ROWS, B), dim3(256), 0, (hipStream_t)stream, x, y, H, W,
                       C / V);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

template <typename T>
int upsample2x_bwd_t(const T* dy, T* dx, int B, int H, int W, int C, void* stream) {
    constexpr int V = VecOf<T>::V;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C % V) || H > 65535 || B > 65535) return DWC_EINVAL;
    hipLaunchKernelGGL(upsample2x_bwd_kernel<T>, dim3((W * (C / V) + 255) / 256, (H + UP_ROWS - 1) / UP_ROWS, B), dim3(256), 0, (hipStream_t)stream, dy, dx, H, W,
                       C / V);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

template <typename T>
int avgpool2_fwd_t(const T* x, T* y, int B, int H, int W, int C, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || (C & 3)) return DWC_EINVAL;
    const size_t total4 = (size_t)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(avgpool2_fwd_kernel<T>, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, C / 4, total4);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

template <typename T>
int avgpool2_bwd_t(const T* dy, T* dx, int B, int H, int W, int C, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || (C & 3)) return DWC_EINVAL;
    const size_t total4 = (size_t)B * H * W * (C / 4);
    hipLaunchKernelGGL(avgpool2_bwd_kernel<T>, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, dy, dx, H, W, C / 4, total4);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

}  // namespace

extern "C" {

int dwc_act_bwd_bias(const float* dy, const float* y, float* g, float* db, int rows, int C, int act, void* ws, size_t ws_bytes,
                     void* stream) {
    return act_bwd_bias_t<float>(dy, y, g, db, rows, C, act, ws, ws_bytes, stream);
}
/* dwc_act_bwd_bias with the absmax slot of g (see dwc_instnorm_fwd_amax) */
int dwc_act_bwd_bias_amax(const float* dy, const float* y, float* g, float* db, int rows, int C, int act, void* ws, size_t ws_bytes,
                          void* g_amax, unsigned g_epoch, void* stream) {
    return act_bwd_bias_t<float>(dy, y, g, db, rows, C, act, ws, ws_bytes, stream, (unsigned long long*)g_amax, g_epoch);
}
int dwc_bf16_act_bwd_bias(const void* dy, const void* y, void* g, float* db, int rows, int C, int act, void* ws, size_t ws_bytes,
                          void* stream) {
    return act_bwd_bias_t<dwc_bf16>((const dwc_bf16*)dy, (const dwc_bf16*)y, (dwc_bf16*)g, db, rows, C, act, ws, ws_bytes, stream);
}
int dwc_upsample2x_fwd(const float* x, float* y, int B, int H, int W, int C, void* stream) { return upsample2x_fwd_t<float>(x, y, B, H, W, C, stream); }
int dwc_upsample2x_bwd(const float* dy, float* dx, int B, int H, int W, int C, void* stream) { return upsample2x_bwd_t<float>(dy, dx, B, H, W, C, stream); }
int dwc_avgpool2_fwd(const float* x, float* y, int B, int H, int W, int C, void* stream) { return avgpool2_fwd_t<float>(x, y, B, H, W, C, stream); }
int dwc_avgpool2_bwd(const float* dy, float* dx, int B, int H, int W, int C, void* stream) { return avgpool2_bwd_t<float>(dy, dx, B, H, W, C, stream); }
int dwc_bf16_upsample2x_fwd(const void* x, void* y, int B, int H, int W, int C, void* stream) {
    return upsample2x_fwd_t<dwc_bf16>((const dwc_bf16*)x, (dwc_bf16*)y, B, H, W, C, stream);
}
int dwc_bf16_upsample2x_bwd(const void* dy, void* dx, int B, int H, int W, int C, void* stream) {
    return upsample2x_bwd_t<dwc_bf16>((const dwc_bf16*)dy, (dwc_bf16*)dx, B, H, W, C, stream);
}
int dwc_bf16_avgpool2_fwd(const void* x, void* y, int B, int H, int W, int C, void* stream) {
    return avgpool2_fwd_t<dwc_bf16>((const dwc_bf16*)x, (dwc_bf16*)y, B, H, W, C, stream);
}
int dwc_bf16_avgpool2_bwd(const void* dy, void* dx, int B, int H, int W, int C, void* stream) {
    return avgpool2_bwd_t<dwc_bf16>((const dwc_bf16*)dy, (dwc_bf16*)dx, B, H, W, C, stream);
}

int dwc_maxpool2_fwd(const float* x, float* y, int B, int H, int W, int C, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || (C & 3)) return DWC_EINVAL;
    const size_t total4 = (size_t)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool2_fwd_kernel<float>, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, x, y, H, W, C / 4, total4);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

int dwc_bf16_maxpool2_fwd(const void* x, void* y, int B, int H, int W, int C, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || (C & 3)) return DWC_EINVAL;
    const size_t total4 = (size_t)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool2_fwd_kernel<dwc_bf16>, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, (const dwc_bf16*)x,
                       (dwc_bf16*)y, H, W, C / 4, total4);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

int dwc_bf16_maxpool2_bwd(const void* x, const void* dy, void* dx, int B, int H, int W, int C, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || (C & 3)) return DWC_EINVAL;
    const size_t total4 = (size_t)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool2_bwd_kernel<dwc_bf16>, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, (const dwc_bf16*)x,
                       (const dwc_bf16*)dy, (dwc_bf16*)dx, H, W, C / 4, total4);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

int dwc_maxpool2_bwd(const float* x, const float* dy, float* dx, int B, int H, int W, int C, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || (H & 1) || (W & 1) || C <= 0 || (C & 3)) return DWC_EINVAL;
    const size_t total4 = (size_t)B * (H / 2) * (W / 2) * (C / 4);
    hipLaunchKernelGGL(maxpool2_bwd_kernel<float>, dim3(grid_for(total4)), dim3(256), 0, (hipStream_t)stream, x, dy, dx, H, W, C / 4,
                       total4);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

int dwc_pack_nchw_to_nhwc4(const float* x, float* y, int B, int C, int H, int W, void* stream) {
    if (C < 1 || C > 4 || B <= 0 || H <= 0 || W <= 0) return DWC_EINVAL;
    const size_t total = (size_t)B * H * W;
    hipLaunchKernelGGL(pack_nhwc4_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, y, C, H * W, total);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

int dwc_unpack_nhwc4_to_nchw(const float* x, float* y, int B, int C, int H, int W, void* stream) {
    if (C < 1 || C > 4 || B <= 0 || H <= 0 || W <= 0) return DWC_EINVAL;
    const size_t total = (size_t)B * H * W;
    hipLaunchKernelGGL(unpack_nhwc4_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, y, C, H * W, total);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

int dwc_blend_fwd(const float* heads, const float* real, float* out, int npix, void* stream) {
    if (npix <= 0) return DWC_EINVAL;
    hipLaunchKernelGGL(blend_fwd_kernel, dim3((npix + 255) / 256), dim3(256), 0, (hipStream_t)stream, heads, real, out, (size_t)npix);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

int dwc_blend_bwd(const float* dout, const float* heads, const float* real, float* dheads, int npix, void* stream) {
    if (npix <= 0) return DWC_EINVAL;
    hipLaunchKernelGGL(blend_bwd_kernel, dim3((npix + 255) / 256), dim3(256), 0, (hipStream_t)stream, dout, heads, real, dheads,
                       (size_t)npix);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

size_t dwc_l1_ws_bytes(size_t n) { return (size_t)grid_for(n / 4 + 1, 1024) * sizeof(float); }

}  // extern "C"

namespace {

// skip4: 0 = mean over every element; 4 / 8 = the tensors are NHWC4 / NHWC8 images, mean over planes 0..2 only
template <typename T>
int l1_mean_fwd_t(const T* a, const T* b, float* out, size_t n, int skip4, void* ws, size_t ws_bytes, void* stream) {
    if (skip4 == 1) skip4 = 4;
    if (n == 0 || (skip4 != 0 && skip4 != 4 && skip4 != 8) || (skip4 && (n % skip4))) return DWC_EINVAL;
    if (!ws || ws_bytes < (size_t)grid_for(n / 4 + 1, 1024) * sizeof(float)) return DWC_EWORKSPACE;
    const int blocks = grid_for(n / 4 + 1, 1024);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(l1_partial_kernel<T>, dim3(blocks), dim3(256), 0, st, a, b, (float*)ws, n / 4, n, skip4);
    DWC_LAUNCH_CHECK();
    const double count = skip4 ? (double)n * 3.0 / skip4 : (double)n;
    hipLaunchKernelGGL(l1_final_kernel, dim3(1), dim3(256), 0, st, (const float*)ws, out, blocks, (float)(1.0 / count));
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

template <typename T>
int l1_mean_bwd_t(const T* a, const T* b, const float* dout, T* da, T* db, size_t n, int skip4, void* stream) {
    if (skip4 == 1) skip4 = 4;
    if (n == 0 || (skip4 != 0 && skip4 != 4 && skip4 != 8) || (skip4 && (n % skip4))) return DWC_EINVAL;
    const double count = skip4 ? (double)n * 3.0 / skip4 : (double)n;
    hipLaunchKernelGGL(l1_bwd_kernel<T>, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, a, b, dout, da, db, n,
                       (float)(1.0 / count), skip4);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

// ---- GMM style-space KL term (reference gmm.py:13-22; r06) ---------------------------------------------------------------
// sum_k mean_b sum_d 0.5 (log(sigma / e^lv) + (e^lv + (mu - c[b][k])^2) / sigma - 1) over [B][K][D] heads: the reference's (and rounds 1-5's)
// dozen elementwise launches forward and two dozen backward on 16 x 8 x 8 numbers.  One workgroup, fixed summation order.
__global__ __launch_bounds__(1024) void gmm_kl_sp_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv,
                                                             const float* __restrict__ centre, int centre_stride, int B, int K, int D,
                                                             float sigma, float* __restrict__ out) {
    __shared__ float red[1024];
    const int n = B * K * D;
    float s = 0.f;
    for (int e = threadIdx.x; e < n; e += 1024) {
        const int b = e / (K * D), k = (e / D) % K;
        const float c = centre[(size_t)b * centre_stride + k], var = expf(lv[e]), d = mu[e] - c;
        s += 0.5f * (logf(sigma / var) + (var + d * d) / sigma - 1.0f);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] / (float)B;
}

__global__ __launch_bounds__(256) void gmm_kl_sp_bwd_kernel(const float* __restrict__ mu, const float* __restrict__ lv,
                                                            const float* __restrict__ centre, int centre_stride, int B, int K, int D,
                                                            float sigma, const float* __restrict__ dout, float* __restrict__ dmu,
                                                            float* __restrict__ dlv) {
    const int n = B * K * D, e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n) return;
    const int b = e / (K * D), k = (e / D) % K;
    const float g = dout[0] / (float)B;
    if (dmu) dmu[e] = g * (mu[e] - centre[(size_t)b * centre_stride + k]) / sigma;
    if (dlv) dlv[e] = g * 0.5f * (expf(lv[e]) / sigma - 1.0f);
}

}  // namespace

extern "C" {

int dwc_gmm_kl_sp_fwd(const float* mu, const float* lv, const float* centre, int centre_stride, int B, int K, int D, float sigma,
                      float* out, void* stream) {
    if (!mu || !lv || !centre || !out || B <= 0 || K <= 0 || D <= 0 || centre_stride < K || !(sigma > 0.f) ||
        (long long)B * K * D > (1 << 24))
        return DWC_EINVAL;
    hipLaunchKernelGGL(gmm_kl_sp_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, mu, lv, centre, centre_stride, B, K, D, sigma, out);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}
int dwc_gmm_kl_sp_bwd(const float* mu, const float* lv, const float* centre, int centre_stride, int B, int K, int D, float sigma,
                      const float* dout, float* dmu, float* dlv, void* stream) {
    if (!mu || !lv || !centre || !dout || B <= 0 || K <= 0 || D <= 0 || centre_stride < K || !(sigma > 0.f) ||
        (long long)B * K * D > (1 << 24))
        return DWC_EINVAL;
    hipLaunchKernelGGL(gmm_kl_sp_bwd_kernel, dim3((B * K * D + 255) / 256), dim3(256), 0, (hipStream_t)stream, mu, lv, centre,
                       centre_stride, B, K, D, sigma, dout, dmu, dlv);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

int dwc_adv_tail_fwd(const float* src, const float* cls, const float* labels, float* out, int segs, int B, int src_per_sample, int ncls,
                     dwc_adv_spec spec, void* stream) {
    if (!src || !cls || !labels || !out || segs < 1 || segs > 4 || B <= 0 || src_per_sample <= 0 || ncls <= 0) return DWC_EINVAL;
    hipLaunchKernelGGL(adv_tail_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, src, cls, labels, out, segs, B, src_per_sample,
                       ncls, spec);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}
int dwc_adv_tail_bwd(const float* src, const float* cls, const float* labels, const float* dout, float* dsrc, float* dcls, int segs,
                     int B, int src_per_sample, int ncls, dwc_adv_spec spec, void* stream) {
    if (!src || !cls || !labels || !dout || !dsrc || !dcls || segs < 1 || segs > 4 || B <= 0 || src_per_sample <= 0 || ncls <= 0)
        return DWC_EINVAL;
    const int n = segs * B * (src_per_sample + ncls);
    hipLaunchKernelGGL(adv_tail_bwd_kernel, dim3((n + 255) / 256 < 64 ? (n + 255) / 256 : 64), dim3(256), 0, (hipStream_t)stream, src, cls,
                       labels, dout, dsrc, dcls, segs, B, src_per_sample, ncls, spec);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

int dwc_l1_mean_fwd(const float* a, const float* b, float* out, size_t n, int skip4, void* ws, size_t ws_bytes, void* stream) {
    return l1_mean_fwd_t<float>(a, b, out, n, skip4, ws, ws_bytes, stream);
}
int dwc_l1_mean_bwd(const float* a, const float* b, const float* dout, float* da, float* db, size_t n, int skip4, void* stream) {
    return l1_mean_bwd_t<float>(a, b, dout, da, db, n, skip4, stream);
}
int dwc_bf16_l1_mean_fwd(const void* a, const void* b, float* out, size_t n, int skip4, void* ws, size_t ws_bytes, void* stream) {
    return l1_mean_fwd_t<dwc_bf16>((const dwc_bf16*)a, (const dwc_bf16*)b, out, n, skip4, ws, ws_bytes, stream);
}
int dwc_bf16_l1_mean_bwd(const void* a, const void* b, const float* dout, void* da, void* db, size_t n, int skip4, void* stream) {
    return l1_mean_bwd_t<dwc_bf16>((const dwc_bf16*)a, (const dwc_bf16*)b, dout, (dwc_bf16*)da, (dwc_bf16*)db, n, skip4, stream);
}

int dwc_pack_nchw_to_nhwc8_bf16(const float* x, void* y, int B, int C, int H, int W, void* stream) {
    if (C < 1 || C > 8 || B <= 0 || H <= 0 || W <= 0) return DWC_EINVAL;
    const size_t total = (size_t)B * H * W;
    hipLaunchKernelGGL(pack_nhwc8_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, x, (dwc_bf16*)y, C, H * W, total);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}
int dwc_unpack_nhwc8_bf16_to_nchw(const void* x, float* y, int B, int C, int H, int W, void* stream) {
    if (C < 1 || C > 8 || B <= 0 || H <= 0 || W <= 0) return DWC_EINVAL;
    const size_t total = (size_t)B * H * W;
    hipLaunchKernelGGL(unpack_nhwc8_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const dwc_bf16*)x, y, C, H * W, total);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}
int dwc_bf16_blend_fwd(const void* heads, const void* real, void* out, int npix, void* stream) {
    if (npix <= 0) return DWC_EINVAL;
    hipLaunchKernelGGL(blend8_fwd_kernel, dim3((npix + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const dwc_bf16*)heads,
                       (const dwc_bf16*)real, (dwc_bf16*)out, (size_t)npix);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}
int dwc_bf16_blend_bwd(const void* dout, const void* heads, const void* real, void* dheads, int npix, void* stream) {
    if (npix <= 0) return DWC_EINVAL;
    hipLaunchKernelGGL(blend8_bwd_kernel, dim3((npix + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const dwc_bf16*)dout,
                       (const dwc_bf16*)heads, (const dwc_bf16*)real, (dwc_bf16*)dheads, (size_t)npix);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

int dwc_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                  float weight_decay, int step, void* stream) {
    if (n == 0 || step < 1) return DWC_EINVAL;
    const double bc1 = 1.0 - pow((double)beta1, step);
    const double bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, (float)((double)lr / bc1),
                       beta1, beta2, eps, weight_decay, (float)sqrt(bc2));
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

int dwc_adam_multi(const dwc_adam_tensor* tensors_dev, const int* chunk_tensor_dev, const unsigned* chunk_start_dev, int n_chunks,
                   double beta1, double beta2, double eps, double weight_decay, void* stream) {
    if (n_chunks <= 0 || !tensors_dev || !chunk_tensor_dev || !chunk_start_dev) return DWC_EINVAL;
    // hyper-parameters arrive as doubles and 1-beta is formed in double before the cast, as torch.optim.Adam does
    hipLaunchKernelGGL(adam_multi_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, tensors_dev, chunk_tensor_dev,
                       chunk_start_dev, (float)(1.0 - beta1), (float)beta2, (float)(1.0 - beta2), (float)eps, (float)weight_decay);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

int dwc_ema_multi(const dwc_ema_tensor* tensors_dev, const int* chunk_tensor_dev, const unsigned* chunk_start_dev, int n_chunks,
                  float beta, void* stream) {
    if (n_chunks <= 0 || !tensors_dev || !chunk_tensor_dev || !chunk_start_dev) return DWC_EINVAL;
    hipLaunchKernelGGL(ema_multi_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, tensors_dev, chunk_tensor_dev,
                       chunk_start_dev, beta);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

int dwc_ema_lerp(const float* p, float* ema, size_t n, float beta, void* stream) {
    if (n == 0) return DWC_EINVAL;
    hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, p, ema, n, beta);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

}  // extern "C"
