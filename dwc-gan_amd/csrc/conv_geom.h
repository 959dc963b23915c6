// Geometry, planning and launch descriptors shared by the fp32 (conv_igemm.hip) and bf16 (conv_bf16.hip) implicit-GEMM
// convolution kernels: how a GEMM row / column addresses the NHWC source tensor (Gather), where a row lands in the
// destination (Scatter), tile / split selection and the host-side builders of those descriptors for the forward,
// data-gradient and weight-gradient formulations (reference networks.py:579-585 and its autograd).
// Element types are opaque here (void*): `bk` is the K-slab depth in ELEMENTS of the caller's kernels (32 fp32 / 64 bf16,
// both 128 bytes) and `min_log_c` the log2 of the smallest channel count one 16-byte staging chunk may cover.
#pragma once
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dwc_common.h"

namespace {

struct Gather {        // how GEMM row m / column k address the source tensor
    const void* src;   // [B][SH][SW][SC]
    int SH, SW, SC, logSC;
    int OH, OW;        // pixel grid enumerated by m (per image)
    int logOW, logOHW; // log2 of OW and OH*OW when both are powers of two, else -1
    int KH, KW, kw_magic;
    int mul_h, mul_w, kstep, off_h, off_w;  // src_h = oh*mul_h + kh*kstep + off_h (same for w)
    int reflect;       // 1: reflect at the border, 0: zero outside
    int M, K;
    int tap_t;         // 1: taps enumerated (kw, kh) instead of (kh, kw) — weights prepared from the transposed filter
};

struct Scatter {       // where GEMM row m lands in the destination tensor
    void* dst;         // [B][OHf][OWf][N]
    int N;
    int OHf, OWf, os;  // dst pixel = (oh*os + oph, ow*os + opw)
    // crop > 0 (data gradients through a reflect pad): dst is the gradient of the PADDED image, of which only the border ring is
    // wanted there -- a pixel whose cropped coordinate (py - crop, px - crop) lies inside IH x IW goes straight to `inner`
    // ([B][IH][IW][N], the gradient of the unpadded tensor), the ring pixels to dst; a band fold adds them onto `inner` afterwards
    int crop = 0, IH = 0, IW = 0;
    void* inner = nullptr;
};

__device__ __forceinline__ int reflect_idx(int i, int n) {
    i = i < 0 ? -i : i;
    return i >= n ? 2 * (n - 1) - i : i;
}

// Up to four small products with their own geometry, K range and weight matrix in ONE launch (blockIdx.z picks the
// strip): the border ring of a data gradient, see dwc_conv2d_bwd_data_same.
struct Strip {
    Gather g;
    Scatter o;
    const void* w;
    int kt0, kt1, tiles, tiles_n;
    int oph, opw;           // parity class offsets of the scatter (stride-2 gradients: dst pixel = (oh*os + oph, ow*os + opw)), else 0
};
struct StripSet {
    Strip s[8];             // (the stride-1 ring uses 4, the stride-2 ring 8: gridDim.z of the launch)
    int kt_per_part;        // the K range of every strip is cut into gridDim.y parts (summed by fold_ring_kernel)
    size_t part_stride;     // elements between the ring buffers of consecutive parts
};

// Sum of the weight-gradient slabs slab[s][(tap*Cin + ci)][co] over s -> dw[co][ci][tap] for SMALL filters cut into MANY pixel
// splits (the 7x7 / 4x4 stems on 4-plane images: 12 544 / 4 096 elements, up to a few hundred splits): the element-wise reduce kernels
// give every element one thread that walks all slabs (89 / 114 us per call on c1, r03 kernel trace).  Here a workgroup owns 32
// elements, thread (g, e) sums the g-th eighth of the slabs for element e with eight loads in flight, and the eight partial sums are
// added in order through LDS: a fixed order, independent of the launch.
__global__ __launch_bounds__(256) void wgrad_reduce_wide_kernel(const float* __restrict__ slab, float* __restrict__ dw, int splits, int K, int N,
                                                                int Cin, int KHW, int cin_real, int cout_real) {
    __shared__ float part[8][32];
    const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
    const size_t total = (size_t)K * N;
    const size_t idx = (size_t)blockIdx.x * 32 + e;
    const int per = (splits + 7) / 8;
    const int z0 = g * per, z1 = min(splits, z0 + per);
    float s = 0.f;
    if (idx < total) {
        const float* p = slab + idx;
        int z = z0;
        for (; z + 8 <= z1; z += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(z + u) * total];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; z < z1; ++z) s += p[(size_t)z * total];
    }
    part[g][e] = s;
    __syncthreads();
    if (g != 0 || idx >= total) return;
#pragma unroll
    for (int q = 1; q < 8; ++q) s += part[q][e];
    const int co = idx % N;
    const int k = idx / N;
    const int ci = k % Cin, tap = k / Cin;
    if (co >= cout_real || ci >= cin_real) return;
    dw[((size_t)co * cin_real + ci) * KHW + tap] = s;
}

// (host) true when the wide form was launched: few elements, many splits
inline bool wgrad_reduce_wide(const float* slab, float* dw, int splits, int K, int N, int Cin, int KHW, int cin_real, int cout_real,
                              hipStream_t st) {
    const size_t total = (size_t)K * N;
    if (splits < 16 || total > 262144) return false;
    hipLaunchKernelGGL(wgrad_reduce_wide_kernel, dim3((unsigned)((total + 31) / 32)), dim3(256), 0, st, slab, dw, splits, K, N, Cin, KHW,
                       cin_real, cout_real);
    return true;
}

int kw_magic_for(int KW, int max_tap) {
    const int magic = (65536 + KW - 1) / KW;
    for (int tp = 0; tp <= max_tap; ++tp)
        if (((tp * magic) >> 16) != tp / KW) return -1;
    return magic;
}

// ---- tile / split selection ------------------------------------------------------------------
constexpr int NUM_CU = 256;
struct Plan {
    int bm, bn, splits, kt_per_split;
};

// Cost model (relative time of the busiest CU).  A CU holds up to `resident` workgroups of a
// tile shape (LDS / VGPR limits of the direct-to-LDS kernel); it receives n = ceil(blocks/256)
// of them and works through them `resident` at a time.  A full group runs at the tile's measured
// MFMA-rate factor f (relative to 128x128 at 2 WG/CU, r01 kernel_bench: 130 / 108 / ~95 TFLOP/s);
// a trailing partial group runs at reduced efficiency because fewer waves per SIMD are left to
// cover barrier / LDS latency (one WG alone: 0.62).  This is what makes 768 tiles of 128x128
// (1.5 groups) lose to 1536 tiles of 128x64 (exactly 2 groups) for the 3B-batched 3x3 convs.
Plan plan_gemm(int M, int N, int K, int classes, int bk = 32) {
    struct Cand { int bm, bn, resident; float f; };
    static const Cand all[] = {{128, 128, 2, 1.0f}, {128, 64, 3, 0.85f}, {64, 64, 5, 0.72f}, {128, 32, 4, 0.5f}};
    const int nk = (K + bk - 1) / bk;
    Plan best = {128, 32, 1, nk};
    float best_cost = 3.0e38f;
    auto valid = [N](const Cand& c) {
        if (N <= 32) return c.bn == 32;
        if (N <= 64) return c.bn == 64;
        return c.bn != 32;
    };
    for (const Cand& c : all) {
        if (!valid(c)) continue;
        const long blocks = (long)((M + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn) * classes;
        const long n = (blocks + NUM_CU - 1) / NUM_CU;
        const long full = n / c.resident, rem = n % c.resident;
        const float tile = (float)c.bm * c.bn / c.f;
        float cost = (float)full * c.resident * tile;
        if (rem) cost += (float)rem * tile / (rem == 1 ? 0.62f : 0.9f);
        if (cost < best_cost) {
            best_cost = cost;
            best = {c.bm, c.bn, 1, nk};
        }
    }
    // split-K: too few output tiles to fill the chip but a long contraction
    const long blocks = (long)((M + best.bm - 1) / best.bm) * ((N + best.bn - 1) / best.bn) * classes;
    // (r05: the threshold was NUM_CU / 2 -- 128 tiles of a 4096-deep contraction ran 128 slabs each, one barrier per slab: c1's
    // discriminator tail launches 32.9 -> 28.5 us with up to 2 * NUM_CU tiles split, +28 reduce launches; net -0.2 ms per step)
    if (blocks < 2 * NUM_CU && nk >= 8) {
        int s = (int)((2 * NUM_CU + blocks - 1) / blocks);
        if (s > nk / 4) s = nk / 4;
        if (s > 32) s = 32;
        if (s >= 2) {
            best.kt_per_split = (nk + s - 1) / s;
            best.splits = (nk + best.kt_per_split - 1) / best.kt_per_split;
        }
    }
    return best;
}

size_t gemm_ws_bytes(int M, int N, int K, int classes, size_t dst_elems, int bk = 32) {
    const Plan p = plan_gemm(M, N, K, classes, bk);
    return p.splits > 1 ? (size_t)p.splits * dst_elems * sizeof(float) : 0;
}

bool conv_args_ok(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, int min_log_c = 2) {
    if (B <= 0 || H <= 0 || W <= 0 || KH <= 0 || KW <= 0) return false;
    if (dwc_ilog2_exact(Cin) < min_log_c || (Cout & 3) || Cout <= 0) return false;
    if (stride != 1 && stride != 2) return false;
    if (pad < 0 || pad >= H || pad >= W) return false;  // reflect needs pad < size
    if (H + 2 * pad < KH || W + 2 * pad < KW) return false;
    return true;
}

// Split of the pixel contraction of the weight gradient into `splits` ranges of `chunk` rows (a multiple of the
// 32-row slab).  Every workgroup of a launch does the same amount of work, so the launch takes
// rounds * (slabs per workgroup + prologue/epilogue) where a round is one full set of resident workgroups: the
// split count is chosen to fill 1..4 rounds EXACTLY rather than to reach a fixed number of workgroups (29 ranges x
// 36 tiles = 1044 workgroups is two rounds plus a third for the last 20).
void wgrad_plan(int M, int K, int N, int* splits, int* chunk, int classes = 1, int slab_rows = 32) {
    const int bn = N > 64 ? 128 : (N > 32 ? 64 : 32);
    const long resident = bn == 128 ? 2 : (bn == 64 ? 3 : 4);       // workgroups per CU by LDS footprint
    const long tiles = (long)((K + 127) / 128) * ((N + bn - 1) / bn) * classes;
    const long slots = NUM_CU * resident;
    const long slabs = (M + slab_rows - 1) / slab_rows;
    long best_cost = -1, best_c = slabs;
    for (long r = 1; r <= 4; ++r) {
        long s = r * slots / tiles;
        if (s < 1) s = 1;
        if (s > slabs / 8) s = slabs / 8 > 0 ? slabs / 8 : 1;       // at least 8 slabs per workgroup
        const long c = (slabs + s - 1) / s;
        const long s_eff = (slabs + c - 1) / c;
        const long rounds = (tiles * s_eff + slots - 1) / slots;
        const long cost = rounds * (c + 3);
        if (best_cost < 0 || cost < best_cost) {
            best_cost = cost;
            best_c = c;
        }
    }
    *chunk = (int)best_c * slab_rows;
    *splits = (int)((slabs + best_c - 1) / best_c);
}

struct FwdGeom {
    Gather g;
    Scatter o;
    size_t dst_elems;
};

// general forward geometry: separate strides / reflect pads per axis (the plain entry points pass
// the same value twice; the "wide" 8-pixels-per-row form of the image heads uses stride_w = 8)
bool fwd_geom_ex(const void* x, void* y, int B, int H, int W, int Cin, int Cout, int KH, int KW, int sh, int sw, int ph,
                 int pw, FwdGeom* f, int min_log_c = 2) {
    if (B <= 0 || H <= 0 || W <= 0 || KH <= 0 || KW <= 0 || sh < 1 || sw < 1) return false;
    if (dwc_ilog2_exact(Cin) < min_log_c || (Cout & 3) || Cout <= 0) return false;
    if (ph < 0 || pw < 0 || ph >= H || pw >= W || H + 2 * ph < KH || W + 2 * pw < KW) return false;
    const int Ho = (H + 2 * ph - KH) / sh + 1, Wo = (W + 2 * pw - KW) / sw + 1;
    // the furthest gathered index must stay inside the single reflection: i <= 2*(n-1)
    if ((Ho - 1) * sh - ph + KH - 1 > 2 * (H - 1) || (Wo - 1) * sw - pw + KW - 1 > 2 * (W - 1)) return false;
    Gather& g = f->g;
    g.tap_t = 0;
    g.src = x; g.SH = H; g.SW = W; g.SC = Cin; g.logSC = dwc_ilog2_exact(Cin);
    g.OH = Ho; g.OW = Wo; g.KH = KH; g.KW = KW;
    g.kw_magic = kw_magic_for(KW, KH * KW + 64);
    if (g.kw_magic < 0) return false;
    g.mul_h = sh; g.mul_w = sw; g.kstep = 1; g.off_h = -ph; g.off_w = -pw; g.reflect = 1;
    g.M = B * Ho * Wo; g.K = KH * KW * Cin;
    g.logOW = dwc_ilog2_exact(Wo); g.logOHW = dwc_ilog2_exact(Ho * Wo);
    if (g.logOW < 0 || g.logOHW < 0) g.logOW = g.logOHW = -1;
    f->o.dst = y; f->o.N = Cout; f->o.OHf = Ho; f->o.OWf = Wo; f->o.os = 1;
    f->dst_elems = (size_t)g.M * Cout;
    return true;
}

bool fwd_geom(const void* x, void* y, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad, FwdGeom* f,
              int min_log_c = 2) {
    if (!conv_args_ok(B, H, W, Cin, Cout, KH, KW, stride, pad, min_log_c)) return false;
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    Gather& g = f->g;
    g.tap_t = 0;
    g.src = x; g.SH = H; g.SW = W; g.SC = Cin; g.logSC = dwc_ilog2_exact(Cin);
    g.OH = Ho; g.OW = Wo; g.KH = KH; g.KW = KW;
    g.kw_magic = kw_magic_for(KW, KH * KW + 64);
    if (g.kw_magic < 0) return false;
    g.mul_h = g.mul_w = stride; g.kstep = 1; g.off_h = -pad; g.off_w = -pad; g.reflect = 1;
    g.M = B * Ho * Wo; g.K = KH * KW * Cin;
    g.logOW = dwc_ilog2_exact(Wo); g.logOHW = dwc_ilog2_exact(Ho * Wo);
    if (g.logOW < 0 || g.logOHW < 0) g.logOW = g.logOHW = -1;
    f->o.dst = y; f->o.N = Cout; f->o.OHf = Ho; f->o.OWf = Wo; f->o.os = 1;
    f->dst_elems = (size_t)g.M * Cout;
    return true;
}

struct BwdGeom {
    Gather g;
    Scatter o;
    size_t dst_elems, wcs;
    int classes;
};

bool bwd_geom(const void* dy, void* dxp, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
              BwdGeom* f, int bk = 32, int min_log_c = 2) {
    // here the gathered tensor is dy (Cout channels) and the produced one is dx (Cin channels)
    if (!conv_args_ok(B, H, W, Cout, Cin, KH, KW, stride, pad, min_log_c)) return false;
    if (dwc_ilog2_exact(Cout) < min_log_c || (Cin & 3)) return false;
    if (stride == 2 && !(KH == 4 && KW == 4 && pad == 1 && !(H & 1) && !(W & 1))) return false;
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    const int Hp = H + 2 * pad, Wp = W + 2 * pad;
    Gather& g = f->g;
    g.tap_t = 0;
    g.src = dy; g.SH = Ho; g.SW = Wo; g.SC = Cout; g.logSC = dwc_ilog2_exact(Cout);
    g.reflect = 0; g.logOW = g.logOHW = -1;
    f->o.dst = dxp; f->o.N = Cin; f->o.OHf = Hp; f->o.OWf = Wp;
    f->dst_elems = (size_t)B * Hp * Wp * Cin;
    if (stride == 1) {
        g.OH = Hp; g.OW = Wp; g.KH = KH; g.KW = KW;
        g.kw_magic = kw_magic_for(KW, KH * KW + 64);
        if (g.kw_magic < 0) return false;
        g.mul_h = g.mul_w = 1; g.kstep = 1; g.off_h = -(KH - 1); g.off_w = -(KW - 1);
        g.M = B * Hp * Wp; g.K = KH * KW * Cout;
        f->o.os = 1; f->classes = 1; f->wcs = 0;
    } else {
        g.OH = Hp / 2; g.OW = Wp / 2; g.KH = 2; g.KW = 2;
        g.kw_magic = kw_magic_for(2, 64);
        g.mul_h = g.mul_w = 1; g.kstep = -1; g.off_h = 0; g.off_w = 0;
        g.M = B * (Hp / 2) * (Wp / 2); g.K = 4 * Cout;
        f->o.os = 2; f->classes = 4; f->wcs = (size_t)Cin * ((4 * Cout + bk - 1) / bk * bk);
    }
    return true;
}

bool zeropad_dgrad_geom(const void* dy, void* dx, int B, int H, int W, int Cin, int Cout, int KH, int KW, int pad,
                               FwdGeom* f) {
    if (!conv_args_ok(B, H, W, Cout, Cin, KH, KW, 1, pad)) return false;
    if (dwc_ilog2_exact(Cout) < 2 || (Cin & 3) || KH != KW || 2 * pad != KH - 1) return false;
    Gather& g = f->g;
    g.tap_t = 0;
    g.src = dy; g.SH = H; g.SW = W; g.SC = Cout; g.logSC = dwc_ilog2_exact(Cout);
    g.OH = H; g.OW = W; g.KH = KH; g.KW = KW;
    g.kw_magic = kw_magic_for(KW, KH * KW + 64);
    if (g.kw_magic < 0) return false;
    g.mul_h = g.mul_w = 1; g.kstep = 1; g.off_h = -pad; g.off_w = -pad; g.reflect = 0;
    g.M = B * H * W; g.K = KH * KW * Cout;
    g.logOW = dwc_ilog2_exact(W); g.logOHW = dwc_ilog2_exact(H * W);
    if (g.logOW < 0 || g.logOHW < 0) g.logOW = g.logOHW = -1;
    f->o.dst = dx; f->o.N = Cin; f->o.OHf = H; f->o.OWf = W; f->o.os = 1;
    f->dst_elems = (size_t)g.M * Cin;
    return true;
}

// Data gradient of a stride-1 "same" reflect-padded convolution (2*pad == K-1, square filter) WITHOUT building the whole
// padded gradient image: the interior is a zero-padded correlation of dY with the flipped filter over the H x W grid
// (power-of-two geometry, no wasted rows), written straight into dx; the border ring of the padded image -- the only
// part the reflect adjoint needs besides -- is four thin strips, each restricted to the filter rows (columns) that can
// reach real dY pixels, computed by one extra launch and folded onto dx.  Ring work is pad*(KH+KW)/(KH*KW) of the
// 2*pad*(H+W+2*pad)/(H*W) a full padded image would add: 4 % instead of 13 % for 3x3 on 32x32.
struct SameDgrad {
    Gather g;
    Scatter o;
    StripSet ss;
    size_t ring_elems[4], ring_total, dst_elems;
    int max_tiles, parts;
};

bool same_dgrad_geom(const void* dy, const void* w_dg, const void* w_dg_t, void* dx, float* ring, int B, int H, int W,
                     int Cin, int Cout, int KH, int KW, int pad, SameDgrad* f, int bk = 32, int bm = 64) {
    if (!conv_args_ok(B, H, W, Cout, Cin, KH, KW, 1, pad)) return false;
    if (Cout < bk || dwc_ilog2_exact(Cout) < 5 || (Cin & 3) || pad <= 0 || KH != KW || 2 * pad != KH - 1) return false;
    if (H < 2 * pad + 2 || W < 2 * pad + 2) return false;    // the two border bands of an axis must not overlap
    const int Wp = W + 2 * pad;
    const int magic = kw_magic_for(KW, KH * KW + 64);
    if (magic < 0) return false;
    auto base = [&](Gather& g, int OH, int OW, int off_h, int off_w, int tap_t) {
        g.src = dy; g.SH = H; g.SW = W; g.SC = Cout; g.logSC = dwc_ilog2_exact(Cout);
        g.OH = OH; g.OW = OW; g.KH = KH; g.KW = KW; g.kw_magic = magic;
        g.mul_h = g.mul_w = 1; g.kstep = 1; g.off_h = off_h; g.off_w = off_w; g.reflect = 0;
        g.M = B * OH * OW; g.K = KH * KW * Cout; g.tap_t = tap_t;
        g.logOW = dwc_ilog2_exact(OW); g.logOHW = dwc_ilog2_exact(OH * OW);
        if (g.logOW < 0 || g.logOHW < 0) g.logOW = g.logOHW = -1;
    };
    // interior: padded coordinate (i+pad, j+pad) -> source offset -(K-1)+pad = -pad
    base(f->g, H, W, -pad, -pad, 0);
    f->o.dst = dx; f->o.N = Cin; f->o.OHf = H; f->o.OWf = W; f->o.os = 1;
    f->dst_elems = (size_t)B * H * W * Cin;
    // ring strips (padded coordinates): top rows [0,pad), bottom rows [pad+H, Hp), left/right columns of the rows between
    const int spt = Cout / bk;                               // K-slabs per filter tap (Cout is a power of two >= bk)
    const int geo[4][4] = {{pad, Wp, -(KH - 1), -(KW - 1)},
                           {pad, Wp, -(KH - 1) + pad + H, -(KW - 1)},
                           {H, pad, -(KH - 1) + pad, -(KW - 1)},
                           {H, pad, -(KH - 1) + pad, -(KW - 1) + pad + W}};
    float* p = ring;
    f->max_tiles = 0;
    for (int z = 0; z < 4; ++z) {
        Strip& st = f->ss.s[z];
        base(st.g, geo[z][0], geo[z][1], geo[z][2], geo[z][3], z >= 2);
        st.o.dst = p; st.o.N = Cin; st.o.OHf = geo[z][0]; st.o.OWf = geo[z][1]; st.o.os = 1;
        f->ring_elems[z] = (size_t)st.g.M * Cin;
        p += f->ring_elems[z];
        st.w = z < 2 ? w_dg : w_dg_t;
        // strips 0 / 2 sit before the image: only the LAST pad filter rows (columns) reach real pixels; 1 / 3 the first
        const int first = (z & 1) ? 0 : KH - pad;
        st.kt0 = first * KW * spt;
        st.kt1 = ((z & 1) ? pad : KH) * KW * spt;
        st.tiles_n = (Cin + 63) / 64;
        st.tiles = ((st.g.M + bm - 1) / bm) * st.tiles_n;      // bm x 64 tiles (strip_bm)
        st.oph = st.opw = 0;
        if (st.tiles > f->max_tiles) f->max_tiles = st.tiles;
    }
    f->ring_total = (size_t)(p - ring);
    // every strip contracts over pad filter rows (columns): cut that range into parts of >= 8 slabs so that the launch
    // has enough workgroups in flight to hide the load latency of these short K loops
    const int range = pad * KW * spt;
    f->parts = range >= 32 ? 4 : (range >= 24 ? 3 : (range >= 16 ? 2 : 1));
    // (r04) ... unless the strips alone already fill the machine (large batches): every extra part is another fp32 copy of the
    // ring to write and to sum in fold_ring_kernel
    if (4 * f->max_tiles * (bm / 64) >= 1024) f->parts = 1;
    else if (4 * f->max_tiles * (bm / 64) >= 512 && f->parts > 2) f->parts = 2;
    f->ss.kt_per_part = (range + f->parts - 1) / f->parts;
    f->ss.part_stride = f->ring_total;
    return true;
}

// Border ring of the PADDED gradient image of a 4x4 stride-2 reflect-pad-1 convolution (reference networks.py:90,94,437,
// networks_v2.py:107-111), for data gradients whose interior comes from a halo-tiled kernel (dwc_x3_conv2d_s2_bwd_data,
// dwc_bf16_conv2d_s2_halo_bwd_data): padded pixel (P, Q) = (2 oh + cy, 2 ow + cx) of parity class (cy, cx) is a 2x2-tap
// zero-padded correlation over dY (src = o - tap, class weight matrix cy*2 + cx of the stride-2 dgrad layout).  The ring --
// rows P = 0 and P = H+1 (all Q), columns Q = 0 and Q = W+1 (P = 1..H) -- is eight thin strips (2 column / row classes per line),
// written at their place in the padded scratch image `dxp` ([B][H+2][W+2][Cin]); fold_band_kernel then adds them onto dx.
struct S2Ring {
    StripSet ss;
    int max_tiles;
};

inline bool s2_ring_geom(const void* dy, const void* w_dgrad, void* dxp, size_t elem_bytes, int B, int H, int W, int Cin, int Cout,
                         S2Ring* f, int bk = 32, int min_log_c = 2, int bm = 64) {
    if (B <= 0 || H < 4 || W < 4 || (H & 1) || (W & 1) || (Cin & 3) || dwc_ilog2_exact(Cout) < min_log_c) return false;
    const int H2 = H / 2, W2 = W / 2, Hp = H + 2, Wp = W + 2;
    const int Kp = (4 * Cout + bk - 1) / bk * bk;
    const size_t wcs = (size_t)Cin * Kp;                   // elements per class matrix (bwd_geom)
    const int magic = kw_magic_for(2, 64);
    f->max_tiles = 0;
    f->ss.kt_per_part = Kp / bk;
    f->ss.part_stride = 0;
    for (int z = 0; z < 8; ++z) {
        const int line = z >> 1, c = z & 1;                // line: 0 top, 1 bottom, 2 left, 3 right; c: the class along the line
        const int cy = line == 0 ? 0 : (line == 1 ? 1 : c), cx = line == 2 ? 0 : (line == 3 ? 1 : c);
        Strip& st = f->ss.s[z];
        Gather& g = st.g;
        g.src = dy; g.SH = H2; g.SW = W2; g.SC = Cout; g.logSC = dwc_ilog2_exact(Cout);
        g.KH = 2; g.KW = 2; g.kw_magic = magic; g.mul_h = g.mul_w = 1; g.kstep = -1; g.reflect = 0; g.tap_t = 0;
        g.logOW = g.logOHW = -1;
        g.K = 4 * Cout;
        size_t shift = 0;                                  // pixels the strip's first row / column lies behind the image origin of dxp
        if (line < 2) {                                    // rows P = 0 (oh = 0, cy = 0) / P = H+1 (oh = H/2, cy = 1), every column of class cx
            g.OH = 1; g.OW = W2 + 1; g.off_w = 0;
            g.off_h = line == 0 ? 0 : H2;
            shift = line == 0 ? 0 : (size_t)H * Wp;        // 2 * (H/2) rows
        } else {                                           // columns Q = 0 (ow = 0, cx = 0) / Q = W+1 (ow = W/2, cx = 1), rows P = 1..H of class cy
            g.OH = H2; g.OW = 1; g.off_h = cy == 0 ? 1 : 0;      // cy = 0: P = 2 oh, oh = 1..H/2; cy = 1: P = 2 oh + 1, oh = 0..H/2-1
            g.off_w = line == 2 ? 0 : W2;
            shift = (cy == 0 ? (size_t)2 * Wp : 0) + (line == 2 ? 0 : (size_t)W);
        }
        g.M = B * g.OH * g.OW;
        st.o.dst = dxp ? (void*)((char*)dxp + shift * Cin * elem_bytes) : nullptr;
        st.o.N = Cin; st.o.OHf = Hp; st.o.OWf = Wp; st.o.os = 2;
        st.o.crop = 0; st.o.IH = st.o.IW = 0; st.o.inner = nullptr;
        st.oph = cy; st.opw = cx;
        st.w = w_dgrad ? (const void*)((const char*)w_dgrad + (size_t)(cy * 2 + cx) * wcs * elem_bytes) : nullptr;
        st.kt0 = 0; st.kt1 = Kp / bk;
        st.tiles_n = (Cin + 63) / 64;
        st.tiles = ((g.M + bm - 1) / bm) * st.tiles_n;
        if (st.tiles > f->max_tiles) f->max_tiles = st.tiles;
    }
    return true;
}

// Row-tile height of the ring-strip launches (64-column tiles either way).  bf16: 128 rows per workgroup -- two 32x32 accumulators
// per wave sharing every weight fragment -- once the strips hold enough rows to fill the chip with tiles of that size (measured,
// benchmarks/ring_bench.py: -10..15 % at batch 128, equal at 384), 64 below that.  fp32 (split products): always 64 -- the
// 128-row form has to halve its accumulator tile per MFMA group and lost 10..30 % at every batch.
// `rows` = GEMM rows of the longest strip, `strips` x `tiles_n` x `parts` workgroups per row tile.
inline int strip_bm(long rows, int tiles_n, int strips, int parts, bool half) {
    if (!half) return 64;
    return ((rows + 127) / 128) * tiles_n * strips * parts >= 2 * NUM_CU ? 128 : 64;
}

// Data gradient w.r.t. an NHWC4 IMAGE (stem convolutions, Cin = 4): N = 4 would fill 1/8 of a 32-wide MFMA tile, so
// 8 horizontally adjacent pixels x 4 channels are produced as 32 columns of a KH x (KW+7), stride-(1,8) filter bank
// (copy p = the flipped filter shifted right by p taps) applied to dY with the zero rule; the padded gradient image
// lands in `ws` with a row pitch of ceil((W+2*pad)/8)*8 pixels and is folded onto dx (reflect-pad adjoint).
bool image_dgrad_geom(const void* dy, void* dxp, int B, int H, int W, int Cout, int KH, int KW, int pad, FwdGeom* f, int px = 8) {
    // px horizontally adjacent pixels x (32 / px) planes form the 32 GEMM columns (8 x 4 for NHWC4, 4 x 8 for NHWC8)
    if (B <= 0 || H <= 0 || W <= 0 || KH <= 0 || KW <= 0 || pad <= 0 || 2 * pad != KH - 1 || 2 * pad != KW - 1) return false;
    if (dwc_ilog2_exact(Cout) < 5 || pad >= H || pad >= W || (px != 8 && px != 4)) return false;
    const int Hp = H + 2 * pad, Wg = (W + 2 * pad + px - 1) / px;
    Gather& g = f->g;
    g.tap_t = 0;
    g.src = dy; g.SH = H; g.SW = W; g.SC = Cout; g.logSC = dwc_ilog2_exact(Cout);
    g.OH = Hp; g.OW = Wg; g.KH = KH; g.KW = KW + px - 1;
    g.kw_magic = kw_magic_for(KW + px - 1, KH * (KW + px - 1) + 64);
    if (g.kw_magic < 0) return false;
    g.mul_h = 1; g.mul_w = px; g.kstep = 1; g.off_h = -(KH - 1); g.off_w = -(KW - 1); g.reflect = 0;
    g.M = B * Hp * Wg; g.K = KH * (KW + px - 1) * Cout;
    g.logOW = g.logOHW = -1;
    f->o.dst = dxp; f->o.N = 32; f->o.OHf = Hp; f->o.OWf = Wg; f->o.os = 1;
    f->dst_elems = (size_t)g.M * 32;
    return true;
}

}  // namespace
