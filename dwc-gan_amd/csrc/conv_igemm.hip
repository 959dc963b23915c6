// Implicit-GEMM convolution on the fp32 matrix cores of gfx950 (v_mfma_f32_32x32x2_f32).
//
// Replaces nn.ReflectionPad2d + nn.Conv2d (+bias, +activation) and their autograd
// (reference networks.py:579-585) — the padded copy is never materialised: the reflect
// (or, for the data gradient, zero) boundary rule is applied to the gather index while the
// input tile is staged into LDS.
//
// GEMM view (forward):  Y[m][n] = sum_k A[m][k] * Wt[k][n]
//     m = (image, oh, ow) output pixel          M = B*Ho*Wo
//     n = output channel                        N = Cout
//     k = (kh, kw, ci), ci fastest              K = KH*KW*Cin     (NHWC makes ci contiguous)
// The data gradient is the same kernel with another index rule (see dwc_conv2d_bwd_data);
// the weight gradient contracts over m instead (conv_wgrad_kernel).
//
// Tiling: 256 threads = 4 waves, block tile 128(M) x BN(N) x 32(K); each wave owns TMxTN
// 32x32 accumulator tiles (16 VGPRs each).  Operands are staged global -> registers -> LDS
// (16-byte vectors along the channel axis) with the next K-slab's loads issued before the
// MFMAs of the current one.
#include "dwc_common.h"

namespace {

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int A_LD = BK + 4;  // row stride (floats) of the A tile: 36 -> conflict-free b128 reads

struct Gather {        // how GEMM row m / column k address the source tensor
    const float* src;  // [B][SH][SW][SC]
    int SH, SW, SC, logSC;
    int OH, OW;        // pixel grid enumerated by m (per image)
    int KH, KW, kw_magic;
    int mul, kstep, off_h, off_w;  // src_h = oh*mul + kh*kstep + off_h
    int reflect;       // 1: reflect at the border, 0: zero outside
    int M, K;
};

struct Scatter {       // where GEMM row m lands in the destination tensor
    float* dst;        // [B][OHf][OWf][N]
    int N;
    int OHf, OWf, os;  // dst pixel = (oh*os + oph, ow*os + opw)
};

__device__ __forceinline__ int reflect_idx(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

template <int BN, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void conv_gemm_kernel(Gather g, const float* __restrict__ wmat, size_t w_class_stride,
                                                        Scatter o, const float* __restrict__ bias, int act,
                                                        int tiles_n) {
    static_assert(WM * WN == 4 && WM * TM * 32 == BM && WN * TN * 32 == BN, "tile shape");
    __shared__ __attribute__((aligned(16))) float sA[BM * A_LD];
    __shared__ __attribute__((aligned(16))) float sB[BK * BN];

    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;

    // XCD-aware tile order: blocks that share an XCD (id % 8) walk neighbouring tiles
    int bid = blockIdx.x;
    {
        const int nb = gridDim.x;
        if (nb >= 16) {
            const int q = nb >> 3, r = nb & 7, x = bid & 7, y = bid >> 3;
            bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
        }
    }
    const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int cls = blockIdx.z;  // stride-2 data gradient: output parity class
    wmat += (size_t)cls * w_class_stride;
    const int oph = cls >> 1, opw = cls & 1;

    // ---- per-thread gather rows (fixed for the whole K loop) ----
    const int arow = t >> 3;        // + 32*i
    const int acol = (t & 7) * 4;   // k offset inside the slab
    int a_bh[4], a_bw[4], a_img[4];
    bool a_ok[4];
    const int ohw = g.OH * g.OW;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + arow + 32 * i;
        a_ok[i] = m < g.M;
        const int mm = a_ok[i] ? m : 0;
        const int n = mm / ohw;
        const int rem = mm - n * ohw;
        const int oh = rem / g.OW, ow = rem - oh * g.OW;
        a_bh[i] = oh * g.mul + g.off_h;
        a_bw[i] = ow * g.mul + g.off_w;
        a_img[i] = n * g.SH * g.SW;
    }
    // B tile: rows k, BN/4 float4 per row
    constexpr int B_F4 = BN / 4;
    constexpr int B_ROWS_PER_PASS = 256 / B_F4;
    constexpr int B_PASSES = BK / B_ROWS_PER_PASS;
    const int brow = t / B_F4, bcol = (t % B_F4) * 4;

    f32x4 ra[4], rb[B_PASSES];
    const int nk = (g.K + BK - 1) / BK;

    auto load_slab = [&](int kt) {
        const int kg = kt * BK + acol;
        const bool kok = kg < g.K;
        const int tap = kg >> g.logSC;
        const int ci = kg & (g.SC - 1);
        const int kh = (tap * g.kw_magic) >> 16;
        const int kw = tap - kh * g.KW;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            int h = a_bh[i] + kh * g.kstep;
            int w = a_bw[i] + kw * g.kstep;
            bool ok = a_ok[i] && kok;
            if (g.reflect) {
                h = reflect_idx(h, g.SH);
                w = reflect_idx(w, g.SW);
            } else {
                ok = ok && h >= 0 && h < g.SH && w >= 0 && w < g.SW;
            }
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) v = *reinterpret_cast<const f32x4*>(g.src + (((size_t)(a_img[i] + h * g.SW + w)) << g.logSC) + ci);
            ra[i] = v;
        }
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p) {
            const int k = kt * BK + brow + p * B_ROWS_PER_PASS;
            const int n = n0 + bcol;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k < g.K && n < o.N) v = *reinterpret_cast<const f32x4*>(wmat + (size_t)k * o.N + n);
            rb[p] = v;
        }
    };
    auto store_slab = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(&sA[(arow + 32 * i) * A_LD + acol]) = ra[i];
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p)
            *reinterpret_cast<f32x4*>(&sB[(brow + p * B_ROWS_PER_PASS) * BN + bcol]) = rb[p];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    load_slab(0);
    store_slab();
    __syncthreads();

    for (int kt = 0; kt < nk; ++kt) {
        if (kt + 1 < nk) load_slab(kt + 1);
        // 16 MFMA k-steps; lane half `hi` of step (q,j) contracts k = 8q + 4hi + j for both operands
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 a4[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i)
                a4[i] = *reinterpret_cast<const f32x4*>(&sA[((wm * TM + i) * 32 + l31) * A_LD + 8 * q + 4 * hi]);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float b[TN];
#pragma unroll
                for (int n = 0; n < TN; ++n) b[n] = sB[(8 * q + 4 * hi + j) * BN + (wn * TN + n) * 32 + l31];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a4[i][j], b[n], acc[i][n], 0, 0, 0);
            }
        }
        __syncthreads();
        if (kt + 1 < nk) {
            store_slab();
            __syncthreads();
        }
    }

    // ---- epilogue: bias + activation, scatter rows ----
#pragma unroll
    for (int i = 0; i < TM; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            const int m = m0 + row;
            if (m >= g.M) continue;
            const int n_img = m / ohw;
            const int rem = m - n_img * ohw;
            const int oh = rem / g.OW, ow = rem - oh * g.OW;
            const size_t prow = ((size_t)n_img * o.OHf + (oh * o.os + oph)) * o.OWf + (ow * o.os + opw);
#pragma unroll
            for (int n = 0; n < TN; ++n) {
                const int col = n0 + (wn * TN + n) * 32 + l31;
                if (col < o.N) {
                    float v = acc[i][n][r];
                    if (bias) v += bias[col];
                    o.dst[prow * o.N + col] = dwc_act_apply(v, act, col);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// weight gradient: dW[k][n] = sum_m A[m][k] * dY[m][n], split over m ("split-K") into slabs
// ------------------------------------------------------------------------------------------
template <int BN, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(Gather g, const float* __restrict__ dy, int N, float* __restrict__ slab,
                                                         int m_chunk) {
    static_assert(WM * WN == 4 && WM * TM * 32 == 128 && WN * TN * 32 == BN, "tile shape");
    __shared__ __attribute__((aligned(16))) float sA[32 * 128];  // [m][k]
    __shared__ __attribute__((aligned(16))) float sB[32 * BN];   // [m][n]
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    const int k0 = blockIdx.x * 128, n0 = blockIdx.y * BN;
    const int m_begin = blockIdx.z * m_chunk;
    const int m_end = min(g.M, m_begin + m_chunk);

    // A^T tile: thread owns one 4-wide k group (fixed tap / channel) and rows (t>>5)+8i
    const int kg = k0 + (t & 31) * 4;
    const bool kok = kg < g.K;
    const int tap = kg >> g.logSC;
    const int ci = kg & (g.SC - 1);
    const int kh = (tap * g.kw_magic) >> 16;
    const int kw = tap - kh * g.KW;
    const int dh = kh * g.kstep + g.off_h, dw = kw * g.kstep + g.off_w;
    const int arow = t >> 5;
    constexpr int B_F4 = BN / 4;
    constexpr int B_ROWS_PER_PASS = 256 / B_F4;
    constexpr int B_PASSES = 32 / B_ROWS_PER_PASS;
    const int brow = t / B_F4, bcol = (t % B_F4) * 4;
    const int ohw = g.OH * g.OW;

    f32x4 ra[4], rb[B_PASSES];
    auto load_slab = [&](int mb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = mb + arow + 8 * i;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (kok && m < m_end) {
                const int n = m / ohw;
                const int rem = m - n * ohw;
                const int oh = rem / g.OW, ow = rem - oh * g.OW;
                const int h = reflect_idx(oh * g.mul + dh, g.SH);
                const int w = reflect_idx(ow * g.mul + dw, g.SW);
                v = *reinterpret_cast<const f32x4*>(g.src + (((size_t)((n * g.SH + h) * g.SW + w)) << g.logSC) + ci);
            }
            ra[i] = v;
        }
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p) {
            const int m = mb + brow + p * B_ROWS_PER_PASS;
            const int n = n0 + bcol;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (m < m_end && n < N) v = *reinterpret_cast<const f32x4*>(dy + (size_t)m * N + n);
            rb[p] = v;
        }
    };
    auto store_slab = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<f32x4*>(&sA[(arow + 8 * i) * 128 + (t & 31) * 4]) = ra[i];
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p)
            *reinterpret_cast<f32x4*>(&sB[(brow + p * B_ROWS_PER_PASS) * BN + bcol]) = rb[p];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (m_begin < m_end) {
        load_slab(m_begin);
        store_slab();
        __syncthreads();
        for (int mb = m_begin; mb < m_end; mb += 32) {
            if (mb + 32 < m_end) load_slab(mb + 32);
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int ml = 2 * s + hi;
                float a[TM], b[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) a[i] = sA[ml * 128 + (wm * TM + i) * 32 + l31];
#pragma unroll
                for (int n = 0; n < TN; ++n) b[n] = sB[ml * BN + (wn * TN + n) * 32 + l31];
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
                        acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[n], acc[i][n], 0, 0, 0);
            }
            __syncthreads();
            if (mb + 32 < m_end) {
                store_slab();
                __syncthreads();
            }
        }
    }
    float* out = slab + (size_t)blockIdx.z * g.K * N;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = k0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if (k >= g.K) continue;
#pragma unroll
            for (int n = 0; n < TN; ++n) {
                const int col = n0 + (wn * TN + n) * 32 + l31;
                if (col < N) out[(size_t)k * N + col] = acc[i][n][r];
            }
        }
}

// slab[s][(kh,kw,ci)][co] summed over s -> dw[co][ci][kh][kw] (state_dict layout), real channels only
__global__ void wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int splits, int K, int N, int Cin,
                                    int KHW, int cin_real, int cout_real) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)K * N) return;
    const int co = idx % N;
    const int k = idx / N;
    const int ci = k % Cin, tap = k / Cin;
    if (co >= cout_real || ci >= cin_real) return;
    float s = 0.f;
    for (int z = 0; z < splits; ++z) s += slab[(size_t)z * K * N + idx];
    dw[((size_t)co * cin_real + ci) * KHW + tap] = s;
}

// reflect-pad adjoint: fold the padded gradient image back onto the un-padded one
__global__ void fold_reflect_kernel(const float* __restrict__ gp, float* __restrict__ dx, int B, int H, int W, int C4, int pad) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * H * W * C4;
    if (idx >= total) return;
    const int c = idx % C4;
    size_t r = idx / C4;
    const int w = r % W;
    r /= W;
    const int h = r % H;
    const int n = r / H;
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    int hs[3], ws[3], nh = 0, nw = 0;
    hs[nh++] = h + pad;
    if (h >= 1 && h <= pad) hs[nh++] = pad - h;
    if (h >= H - 1 - pad && h <= H - 2) hs[nh++] = pad + 2 * (H - 1) - h;
    ws[nw++] = w + pad;
    if (w >= 1 && w <= pad) ws[nw++] = pad - w;
    if (w >= W - 1 - pad && w <= W - 2) ws[nw++] = pad + 2 * (W - 1) - w;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    const f32x4* g4 = reinterpret_cast<const f32x4*>(gp);
    for (int a = 0; a < nh; ++a)
        for (int b = 0; b < nw; ++b) s += g4[((size_t)(n * Hp + hs[a]) * Wp + ws[b]) * C4 + c];
    reinterpret_cast<f32x4*>(dx)[idx] = s;
}

__global__ void weight_to_hwio_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int KH, int KW,
                                      int cout_pad, int cin_pad) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)KH * KW * cin_pad * cout_pad;
    if (idx >= total) return;
    const int co = idx % cout_pad;
    size_t r = idx / cout_pad;
    const int ci = r % cin_pad;
    const int tap = r / cin_pad;
    float v = 0.f;
    if (co < Cout && ci < Cin) v = w[((size_t)co * Cin + ci) * KH * KW + tap];
    out[idx] = v;
}

// stride 1: out[kh'][kw'][co][ci] = W[co][ci][KH-1-kh'][KW-1-kw']
// stride 2: out[ph][pw][th][tw][co][ci] = W[co][ci][ph+2th][pw+2tw]   (KH=KW=4)
__global__ void weight_to_dgrad_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int KH, int KW,
                                       int stride, int cout_pad, int cin_pad) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)KH * KW * cin_pad * cout_pad;
    if (idx >= total) return;
    const int ci = idx % cin_pad;
    size_t r = idx / cin_pad;
    const int co = r % cout_pad;
    const int tapo = r / cout_pad;
    int kh, kw;
    if (stride == 1) {
        kh = KH - 1 - tapo / KW;
        kw = KW - 1 - tapo % KW;
    } else {
        const int tw = tapo & 1, th = (tapo >> 1) & 1, pw = (tapo >> 2) & 1, ph = (tapo >> 3) & 1;
        kh = ph + 2 * th;
        kw = pw + 2 * tw;
    }
    float v = 0.f;
    if (co < Cout && ci < Cin) v = w[((size_t)co * Cin + ci) * KH * KW + kh * KW + kw];
    out[idx] = v;
}

int kw_magic_for(int KW, int max_tap) {
    const int magic = (65536 + KW - 1) / KW;
    for (int tp = 0; tp <= max_tap; ++tp)
        if (((tp * magic) >> 16) != tp / KW) return -1;
    return magic;
}

int launch_gemm(const Gather& g, const float* w, size_t w_class_stride, int classes, const Scatter& o, const float* bias,
                int act, hipStream_t st) {
    const int tiles_m = (g.M + BM - 1) / BM;
    if (o.N > 64) {
        const int tn = (o.N + 127) / 128;
        hipLaunchKernelGGL((conv_gemm_kernel<128, 2, 2, 2, 2>), dim3(tiles_m * tn, 1, classes), dim3(256), 0, st, g, w,
                           w_class_stride, o, bias, act, tn);
    } else if (o.N > 32) {
        hipLaunchKernelGGL((conv_gemm_kernel<64, 2, 2, 2, 1>), dim3(tiles_m, 1, classes), dim3(256), 0, st, g, w,
                           w_class_stride, o, bias, act, 1);
    } else {
        hipLaunchKernelGGL((conv_gemm_kernel<32, 4, 1, 1, 1>), dim3(tiles_m, 1, classes), dim3(256), 0, st, g, w,
                           w_class_stride, o, bias, act, 1);
    }
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

bool conv_args_ok(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    if (B <= 0 || H <= 0 || W <= 0 || KH <= 0 || KW <= 0) return false;
    if (dwc_ilog2_exact(Cin) < 2 || (Cout & 3) || Cout <= 0) return false;
    if (stride != 1 && stride != 2) return false;
    if (pad < 0 || pad >= H || pad >= W) return false;  // reflect needs pad < size
    if (H + 2 * pad < KH || W + 2 * pad < KW) return false;
    return true;
}

void wgrad_plan(int M, int K, int N, int* splits, int* chunk) {
    const int tiles = ((K + 127) / 128) * ((N + (N > 64 ? 127 : (N > 32 ? 63 : 31))) / (N > 64 ? 128 : (N > 32 ? 64 : 32)));
    int want = (1024 + tiles - 1) / tiles;
    if (want < 1) want = 1;
    int c = (M + want - 1) / want;
    if (c < 256) c = 256;
    c = (c + 31) / 32 * 32;
    *chunk = c;
    *splits = (M + c - 1) / c;
}

}  // namespace

extern "C" {

int dwc_version(void) { return 1; }

int dwc_weight_oihw_to_hwio(const float* w, float* out, int Cout, int Cin, int KH, int KW, int cout_pad, int cin_pad,
                            void* stream) {
    if (cout_pad < Cout || cin_pad < Cin) return DWC_EINVAL;
    const size_t total = (size_t)KH * KW * cin_pad * cout_pad;
    hipLaunchKernelGGL(weight_to_hwio_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, out, Cout, Cin,
                       KH, KW, cout_pad, cin_pad);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

int dwc_weight_oihw_to_dgrad(const float* w, float* out, int Cout, int Cin, int KH, int KW, int stride, int cout_pad,
                             int cin_pad, void* stream) {
    if (cout_pad < Cout || cin_pad < Cin) return DWC_EINVAL;
    if (stride == 2 && !(KH == 4 && KW == 4)) return DWC_EINVAL;
    if (stride != 1 && stride != 2) return DWC_EINVAL;
    const size_t total = (size_t)KH * KW * cin_pad * cout_pad;
    hipLaunchKernelGGL(weight_to_dgrad_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, out, Cout, Cin,
                       KH, KW, stride, cout_pad, cin_pad);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

int dwc_conv2d_fwd(const float* x, const float* w_hwio, const float* bias, float* y, int B, int H, int W, int Cin, int Cout,
                   int KH, int KW, int stride, int pad, int act, void* stream) {
    if (!conv_args_ok(B, H, W, Cin, Cout, KH, KW, stride, pad)) return DWC_EINVAL;
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    Gather g;
    g.src = x; g.SH = H; g.SW = W; g.SC = Cin; g.logSC = dwc_ilog2_exact(Cin);
    g.OH = Ho; g.OW = Wo; g.KH = KH; g.KW = KW;
    g.kw_magic = kw_magic_for(KW, KH * KW + 64);
    if (g.kw_magic < 0) return DWC_EINVAL;
    g.mul = stride; g.kstep = 1; g.off_h = -pad; g.off_w = -pad; g.reflect = 1;
    g.M = B * Ho * Wo; g.K = KH * KW * Cin;
    Scatter o;
    o.dst = y; o.N = Cout; o.OHf = Ho; o.OWf = Wo; o.os = 1;
    return launch_gemm(g, w_hwio, 0, 1, o, bias, act, (hipStream_t)stream);
}

int dwc_reflect_pad_adjoint(const float* dxp, float* dx, int B, int H, int W, int C, int pad, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || pad < 0 || pad >= H || pad >= W) return DWC_EINVAL;
    const size_t total = (size_t)B * H * W * (C / 4);
    hipLaunchKernelGGL(fold_reflect_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, dxp, dx, B, H, W, C / 4,
                       pad);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

int dwc_conv2d_bwd_data(const float* dy, const float* w_dgrad, float* dxp, int B, int H, int W, int Cin, int Cout, int KH,
                        int KW, int stride, int pad, void* stream) {
    // here the gathered tensor is dy (Cout channels) and the produced one is dx (Cin channels)
    if (!conv_args_ok(B, H, W, Cout, Cin, KH, KW, stride, pad)) return DWC_EINVAL;
    if (dwc_ilog2_exact(Cout) < 2 || (Cin & 3)) return DWC_EINVAL;
    if (stride == 2 && !(KH == 4 && KW == 4 && pad == 1 && !(H & 1) && !(W & 1))) return DWC_EINVAL;
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    float* target = dxp;
    hipStream_t st = (hipStream_t)stream;
    Gather g;
    g.src = dy; g.SH = Ho; g.SW = Wo; g.SC = Cout; g.logSC = dwc_ilog2_exact(Cout);
    g.reflect = 0;
    Scatter o;
    o.dst = target; o.N = Cin; o.OHf = Hp; o.OWf = Wp;
    int rc;
    if (stride == 1) {
        g.OH = Hp; g.OW = Wp; g.KH = KH; g.KW = KW;
        g.kw_magic = kw_magic_for(KW, KH * KW + 64);
        if (g.kw_magic < 0) return DWC_EINVAL;
        g.mul = 1; g.kstep = 1; g.off_h = -(KH - 1); g.off_w = -(KW - 1);
        g.M = B * Hp * Wp; g.K = KH * KW * Cout;
        o.os = 1;
        rc = launch_gemm(g, w_dgrad, 0, 1, o, nullptr, DWC_ACT_NONE, st);
    } else {
        g.OH = Hp / 2; g.OW = Wp / 2; g.KH = 2; g.KW = 2;
        g.kw_magic = kw_magic_for(2, 64);
        g.mul = 1; g.kstep = -1; g.off_h = 0; g.off_w = 0;
        g.M = B * (Hp / 2) * (Wp / 2); g.K = 4 * Cout;
        o.os = 2;
        rc = launch_gemm(g, w_dgrad, (size_t)4 * Cout * Cin, 4, o, nullptr, DWC_ACT_NONE, st);
    }
    return rc;
}

size_t dwc_conv2d_bwd_weight_ws_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    int splits, chunk;
    wgrad_plan(B * Ho * Wo, KH * KW * Cin, Cout, &splits, &chunk);
    return (size_t)splits * KH * KW * Cin * Cout * sizeof(float);
}

int dwc_conv2d_bwd_weight(const float* x, const float* dy, float* dw_oihw, int B, int H, int W, int Cin, int Cout, int KH,
                          int KW, int stride, int pad, int cin_real, int cout_real, void* ws, size_t ws_bytes, void* stream) {
    if (!conv_args_ok(B, H, W, Cin, Cout, KH, KW, stride, pad)) return DWC_EINVAL;
    if (cin_real > Cin || cout_real > Cout) return DWC_EINVAL;
    if (!ws || ws_bytes < dwc_conv2d_bwd_weight_ws_bytes(B, H, W, Cin, Cout, KH, KW, stride, pad)) return DWC_EWORKSPACE;
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    hipStream_t st = (hipStream_t)stream;
    Gather g;
    g.src = x; g.SH = H; g.SW = W; g.SC = Cin; g.logSC = dwc_ilog2_exact(Cin);
    g.OH = Ho; g.OW = Wo; g.KH = KH; g.KW = KW;
    g.kw_magic = kw_magic_for(KW, KH * KW + 64);
    if (g.kw_magic < 0) return DWC_EINVAL;
    g.mul = stride; g.kstep = 1; g.off_h = -pad; g.off_w = -pad; g.reflect = 1;
    g.M = B * Ho * Wo; g.K = KH * KW * Cin;
    int splits, chunk;
    wgrad_plan(g.M, g.K, Cout, &splits, &chunk);
    float* slab = (float*)ws;
    const int tk = (g.K + 127) / 128;
    if (Cout > 64) {
        hipLaunchKernelGGL((conv_wgrad_kernel<128, 2, 2, 2, 2>), dim3(tk, (Cout + 127) / 128, splits), dim3(256), 0, st, g, dy,
                           Cout, slab, chunk);
    } else if (Cout > 32) {
        hipLaunchKernelGGL((conv_wgrad_kernel<64, 2, 2, 2, 1>), dim3(tk, 1, splits), dim3(256), 0, st, g, dy, Cout, slab, chunk);
    } else {
        hipLaunchKernelGGL((conv_wgrad_kernel<32, 4, 1, 1, 1>), dim3(tk, 1, splits), dim3(256), 0, st, g, dy, Cout, slab, chunk);
    }
    DWC_LAUNCH_CHECK();
    const size_t total = (size_t)g.K * Cout;
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, slab, dw_oihw, splits, g.K, Cout, Cin,
                       KH * KW, cin_real, cout_real);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

}  // extern "C"
