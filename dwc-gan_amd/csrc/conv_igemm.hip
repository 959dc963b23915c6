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
// Tiling: 256 threads = 4 waves, block tile BM x BN x 32(K); each wave owns TMxTN 32x32
// accumulator tiles (16 VGPRs each).  Operands are staged global -> LDS directly (16-byte
// vectors along the channel axis) into a double-buffered LDS tile: the next K-slab's loads are
// issued before the MFMAs of the current one and there is ONE barrier per slab.
// Tile shape is picked per problem so that every CU holds >= 2 workgroups whenever the grid
// allows it (one wave per SIMD cannot hide its own LDS/barrier latency), and products with few
// output tiles but a long K (the discriminator tails, M = B*16 rows) are split along K into
// partial images that a small epilogue kernel sums (+bias, +activation) in a fixed order.
#include <type_traits>

#include <atomic>

#include "conv_geom.h"

namespace {

constexpr int BK = 32;


// ------------------------------------------------------------------------------------------
// Forward / data-gradient GEMM.  Staging is DIRECT global->LDS (global_load_lds_dwordx4): no
// staging registers, no ds_write pass.  A wave instruction deposits 64 x 16 B contiguously, i.e.
// 8 rows x 128 B of an UNPADDED [row][32] tile; bank conflicts of the ds_read_b128 fragment reads
// are avoided by an XOR swizzle of the 16-byte chunk index with ((row >> 1) & 7), applied on the
// SOURCE address (which K-chunk a lane fetches) and again on the read.  Out-of-bounds taps of the
// zero-padded data gradient fetch from a zero page.  (Measured against the register-staged
// variant this replaced, r01: 5x5 256->128 @64^2 113 -> 130 TFLOP/s, 3x3 256->256 @32^2 94 -> 98.)
// ------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) float g_zero_page[64];

#ifdef DWC_CLOCK_PROBE   // development builds only (make PROBE=1): average shader clock seen by the GEMM kernels
__device__ unsigned long long g_clock_probe[2];
struct ClockProbe {
    long long c0, r0;
    __device__ ClockProbe() : c0(clock64()), r0(wall_clock64()) {}
    __device__ ~ClockProbe() {
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) {
            atomicAdd(&g_clock_probe[0], (unsigned long long)(clock64() - c0));
            atomicAdd(&g_clock_probe[1], (unsigned long long)(wall_clock64() - r0));
        }
    }
};
#define DWC_PROBE() ClockProbe clock_probe_
#else
#define DWC_PROBE()
#endif

// Workgroup barrier that also retires this wave's outstanding direct-to-LDS loads.  hipcc's
// __syncthreads() already drains vmcnt when an LDS-DMA is in flight (checked in the ISA: every
// s_barrier is preceded by s_waitcnt vmcnt(0) lgkmcnt(0)); the explicit wait makes the kernels
// independent of that compiler behaviour.
__device__ __forceinline__ void lds_dma_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

// One output tile (or, when part_stride != 0, its contribution from K-slabs [kt0, kt1_in)).  `bid`/`nb`: this
// workgroup's linear tile index and the number of tiles of the launch; (oph, opw): parity class of a stride-2 gradient.
// X3 (r04): the SAME kernel -- staging, geometry, prepared fp32 weights, epilogue -- with the inner product taken as exact
// three-way bf16 split products on the bf16 matrix cores (conv_halo_x3.hip explains the arithmetic: a = a0 + a1 + a2 with 8
// significand bits each, six of the nine partial products, the rest below one fp32 rounding; leading product and corrections
// in separate accumulators).  The fp32 fragments are read from the unchanged LDS image and split in registers in front of the
// MFMAs: ~44 vector-ALU instructions per 32x16 fragment, which two workgroups per CU hide behind each other's 48 bf16 MFMAs
// per slab.  For the layers the halo form cannot take (small discriminator layers, 7x7 image layers, ring strips, 1x1).
typedef __bf16 gx3_bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned gx3_u32x4 __attribute__((ext_vector_type(4)));

// 8 fp32 (two 16-byte chunks of a row) -> three bf16x8 planes, exact: v = p0 + p1 + p2 (truncate, subtract, truncate, subtract)
__device__ __forceinline__ void gx3_split8(const f32x4& a, const f32x4& b, gx3_bf16x8& p0, gx3_bf16x8& p1, gx3_bf16x8& p2) {
    unsigned h[8], m[8], l[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const float v = k < 4 ? a[k] : b[k - 4];
        const unsigned hb = __float_as_uint(v) & 0xffff0000u;
        const float r = v - __uint_as_float(hb);
        const unsigned mb = __float_as_uint(r) & 0xffff0000u;
        const float q = r - __uint_as_float(mb);
        h[k] = hb; m[k] = mb; l[k] = __float_as_uint(q);
    }
    gx3_u32x4 u0, u1, u2;
#pragma unroll
    for (int k = 0; k < 4; ++k) {                      // bytes 2,3 of the even element below bytes 2,3 of the odd one
        u0[k] = __builtin_amdgcn_perm(h[2 * k + 1], h[2 * k], 0x07060302u);
        u1[k] = __builtin_amdgcn_perm(m[2 * k + 1], m[2 * k], 0x07060302u);
        u2[k] = __builtin_amdgcn_perm(l[2 * k + 1], l[2 * k], 0x07060302u);
    }
    p0 = __builtin_bit_cast(gx3_bf16x8, u0);
    p1 = __builtin_bit_cast(gx3_bf16x8, u1);
    p2 = __builtin_bit_cast(gx3_bf16x8, u2);
}

template <int BM, int BN, int WM, int WN, int TM, int TN, bool X3 = false>
__device__ __forceinline__ void conv_gemm_body(const Gather& g, const float* __restrict__ wmat, const Scatter& o,
                                               const float* __restrict__ bias, int act, int tiles_n, int kt0, int kt1_in,
                                               size_t part_offset, bool partial, int oph, int opw, int bid, int nb) {
    static_assert(WM * WN == 4 && WM * TM * 32 == BM && WN * TN * 32 == BN, "tile shape");
    constexpr int A_PASSES = BM / 32, B_PASSES = BN / 32;
    constexpr int A_TILE = BM * BK, B_TILE = BN * BK;
    __shared__ __attribute__((aligned(16))) float smem[2 * (A_TILE + B_TILE)];
    float* sA = smem;
    float* sB = smem + 2 * A_TILE;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;

    if (nb >= 16) {     // XCD-aware remap: the 32 CUs of one XCD walk neighbouring tiles
        const int q = nb >> 3, r = nb & 7, x = bid & 7, y = bid >> 3;
        bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
    }
    const int tile_m = bid / tiles_n, tile_n = bid % tiles_n;
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    const int arow = t >> 3;                              // tile row this lane stages (+32 per pass)
    const int acol = (((t & 7) ^ ((arow >> 1) & 7))) * 4;  // LOGICAL k offset fetched into physical chunk t&7
    int a_bh[A_PASSES], a_bw[A_PASSES], a_img[A_PASSES];
    const int ohw = g.OH * g.OW;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
        const int mm = min(m0 + arow + 32 * i, g.M - 1);
        const int n = mm / ohw;
        const int rem = mm - n * ohw;
        const int oh = rem / g.OW, ow = rem - oh * g.OW;
        a_bh[i] = oh * g.mul_h + g.off_h;
        a_bw[i] = ow * g.mul_w + g.off_w;
        a_img[i] = n * g.SH * g.SW;
    }
    const int Kp = (g.K + BK - 1) / BK * BK;
    const float* b_ptr[B_PASSES];
#pragma unroll
    for (int p = 0; p < B_PASSES; ++p) b_ptr[p] = wmat + (size_t)min(n0 + arow + 32 * p, o.N - 1) * Kp + acol;

    const int nk_all = Kp / BK;
    const int kt1 = min(nk_all, kt1_in);
    const int n_taps = g.KH * g.KW;

    int cur_tap = -1;
    const float* a_src[A_PASSES];     // per-row source of the current tap (zero page when out of bounds)
    const bool tap_uniform = g.SC >= BK;
    auto row_sources = [&](int tap, int ci) {
        int kh = (tap * g.kw_magic) >> 16;
        int kw = tap - kh * g.KW;
        if (g.tap_t) {          // square filters only: the same decode gives (kw, kh)
            const int x = kh;
            kh = kw;
            kw = x;
        }
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            int h = a_bh[i] + kh * g.kstep;
            int w = a_bw[i] + kw * g.kstep;
            bool inb = true;
            if (g.reflect) {
                h = reflect_idx(h, g.SH);
                w = reflect_idx(w, g.SW);
            } else {
                inb = h >= 0 && h < g.SH && w >= 0 && w < g.SW;
                h = min(max(h, 0), g.SH - 1);
                w = min(max(w, 0), g.SW - 1);
            }
            const float* p = (const float*)g.src + ((a_img[i] + h * g.SW + w) << g.logSC) + ci;
            a_src[i] = inb ? p : g_zero_page;   // (no channel offset is ever added to the zero page, see stage_slab)
        }
    };
    // issue the direct loads of slab kt into LDS buffer `buf`
    auto stage_slab = [&](int kt, int buf) {
        int coff;   // channel offset added to a_src (0 when the source is the zero page)
        if (tap_uniform) {
            const int kg0 = kt * BK;
            const int tap = min(kg0 >> g.logSC, n_taps - 1);
            if (tap != cur_tap) {
                row_sources(tap, acol);
                cur_tap = tap;
            }
            coff = kg0 & (g.SC - 1);
        } else {
            const int kg = kt * BK + acol;
            row_sources(min(kg >> g.logSC, n_taps - 1), kg & (g.SC - 1));
            coff = 0;
        }
        float* la = sA + buf * A_TILE + wave * (8 * BK);
        float* lb = sB + buf * B_TILE + wave * (8 * BK);
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            const float* s = a_src[i];
            if (s != g_zero_page) s += coff;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                             (__attribute__((address_space(3))) void*)(la + i * 32 * BK), 16, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_ptr[p] + kt * BK),
                                             (__attribute__((address_space(3))) void*)(lb + p * 32 * BK), 16, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // fragment reads: logical chunk 2q+hi of row r sits in physical chunk (2q+hi) ^ ((r>>1)&7)
    const int fsw = (l31 >> 1) & 7;
    int frag_off[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) frag_off[q] = ((2 * q + hi) ^ fsw) * 4;
    const int a_row = (wm * TM * 32 + l31) * BK;
    const int b_row = (wn * TN * 32 + l31) * BK;
    f32x4 fa[2][TM], fb[2][TN];
    auto load_frags = [&](int set, int buf, int q) {
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

    if constexpr (X3) {
        static_assert(TM * TN <= 2, "split accumulators: at most two 32x32 tiles per wave");
        f32x16 lo[TM][TN];                               // the five correction products, merged once at the end
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) lo[i][j][r] = 0.f;
        // k-step s (16 deep) of a slab: lane half hi takes k = 16 s + 8 hi .. + 7 = logical chunks 4 s + 2 hi, + 1 of its row
        int xoff[2][2];
#pragma unroll
        for (int ss = 0; ss < 2; ++ss)
#pragma unroll
            for (int e = 0; e < 2; ++e) xoff[ss][e] = ((4 * ss + 2 * hi + e) ^ fsw) * 4;
        gx3_bf16x8 pa[3][TM], pb[3][TN];
        auto frags_x3 = [&](int buf, int ss) {
            const float* a = sA + buf * A_TILE + a_row;
            const float* b = sB + buf * B_TILE + b_row;
#pragma unroll
            for (int i = 0; i < TM; ++i)
                gx3_split8(*reinterpret_cast<const f32x4*>(a + i * 32 * BK + xoff[ss][0]),
                           *reinterpret_cast<const f32x4*>(a + i * 32 * BK + xoff[ss][1]), pa[0][i], pa[1][i], pa[2][i]);
#pragma unroll
            for (int n = 0; n < TN; ++n)
                gx3_split8(*reinterpret_cast<const f32x4*>(b + n * 32 * BK + xoff[ss][0]),
                           *reinterpret_cast<const f32x4*>(b + n * 32 * BK + xoff[ss][1]), pb[0][n], pb[1][n], pb[2][n]);
        };
        auto mfma_x3 = [&]() {
            constexpr int TA[6] = {0, 1, 0, 2, 1, 0}, TB[6] = {0, 0, 1, 0, 1, 2};
#pragma unroll
            for (int term = 0; term < 6; ++term)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int n = 0; n < TN; ++n) {
                        if (term == 0) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[TA[term]][i], pb[TB[term]][n], acc[i][n], 0, 0, 0);
                        else lo[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[TA[term]][i], pb[TB[term]][n], lo[i][n], 0, 0, 0);
                    }
        };
        if (kt0 < kt1) {
            stage_slab(kt0, 0);
            lds_dma_barrier();
            int buf = 0;
            for (int kt = kt0; kt < kt1; ++kt) {
                if (kt + 1 < kt1) stage_slab(kt + 1, buf ^ 1);   // other buffer: fully read before the last barrier
                frags_x3(buf, 0);
                mfma_x3();
                frags_x3(buf, 1);
                lds_dma_barrier();                            // direct loads landed (vmcnt), this buffer fully read
                mfma_x3();
                buf ^= 1;
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += lo[i][j][r];
    } else if (kt0 < kt1) {
        stage_slab(kt0, 0);
        lds_dma_barrier();
        int buf = 0;
        load_frags(0, 0, 0);
        for (int kt = kt0; kt < kt1; ++kt) {
            const bool more = kt + 1 < kt1;
            if (more) stage_slab(kt + 1, buf ^ 1);       // other buffer: fully read before the last barrier
            load_frags(1, buf, 1);
            mfma_group(0);
            load_frags(0, buf, 2);
            mfma_group(1);
            load_frags(1, buf, 3);
            mfma_group(0);
            lds_dma_barrier();                            // direct loads landed (vmcnt), this buffer fully read
            if (more) load_frags(0, buf ^ 1, 0);
            mfma_group(1);
            buf ^= 1;
        }
    }

    float* dst = (float*)o.dst + part_offset;
    const float slope = dwc_act_slope(act);
    // bias of this lane's TN columns, loaded ONCE in one batch (r04: inside the per-element conditionals of the store loop the
    // compiler can neither hoist nor batch the load -- TM*16*TN dependent round trips per lane)
    float bcol[TN];
#pragma unroll
    for (int n = 0; n < TN; ++n) bcol[n] = 0.f;
    if (bias && !partial) {
#pragma unroll
        for (int n = 0; n < TN; ++n) bcol[n] = bias[min(n0 + (wn * TN + n) * 32 + l31, o.N - 1)];
    }
    // two loop nests (dwc_common.h, dwc_act_simple): the transcendental activations stay out of the common path's code
    auto store = [&](auto general) {
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
                const int py = oh * o.os + oph, px = ow * o.os + opw;
                float* drow = dst + (((size_t)n_img * o.OHf + py) * o.OWf + px) * o.N;
                if (o.crop && !partial) {                 // Scatter::crop: interior pixels of a padded gradient image go straight to dx
                    const int yy = py - o.crop, xx = px - o.crop;
                    if ((unsigned)yy < (unsigned)o.IH && (unsigned)xx < (unsigned)o.IW)
                        drow = (float*)o.inner + (((size_t)n_img * o.IH + yy) * o.IW + xx) * o.N;
                }
#pragma unroll
                for (int n = 0; n < TN; ++n) {
                    const int col = n0 + (wn * TN + n) * 32 + l31;
                    if (col < o.N) {
                        float v = acc[i][n][r];
                        if (!partial) {
                            v += bcol[n];
                            if constexpr (decltype(general)::value) v = dwc_act_apply(v, act, col);
                            else v = dwc_act_simple(v, slope);
                        }
                        drow[col] = v;
                    }
                }
            }
        }
    };
    if (dwc_act_is_simple(act)) store(std::false_type{});
    else store(std::true_type{});
}

template <int BM, int BN, int WM, int WN, int TM, int TN, bool X3 = false>
__global__ __launch_bounds__(256) void conv_gemm_kernel(Gather g, const float* __restrict__ wmat, size_t w_class_stride,
                                                             Scatter o, const float* __restrict__ bias, int act, int tiles_n,
                                                             int kt_per_split, size_t part_stride) {
    DWC_PROBE();
    const int cls = blockIdx.z, split = blockIdx.y;
    conv_gemm_body<BM, BN, WM, WN, TM, TN, X3>(g, wmat + (size_t)cls * w_class_stride, o, bias, act, tiles_n, split * kt_per_split,
                                           (split + 1) * kt_per_split, (size_t)split * part_stride, part_stride != 0, cls >> 1,
                                           cls & 1, blockIdx.x, gridDim.x);
}

template <int BM, int BN, int WM, int WN, int TM, int TN, bool X3 = false>
__global__ __launch_bounds__(256) void conv_gemm_strips_kernel(StripSet ss) {
    const Strip& s = ss.s[blockIdx.z];
    if ((int)blockIdx.x >= s.tiles) return;
    const int kt0 = s.kt0 + blockIdx.y * ss.kt_per_part;
    conv_gemm_body<BM, BN, WM, WN, TM, TN, X3>(s.g, (const float*)s.w, s.o, nullptr, DWC_ACT_NONE, s.tiles_n, kt0, min(s.kt1, kt0 + ss.kt_per_part),
                                           blockIdx.y * ss.part_stride, false, s.oph, s.opw, blockIdx.x, s.tiles);
}

// dst[i] = act(sum_s part[s][i] + bias[i % N]), fixed summation order
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ dst,
                                                            const float* __restrict__ bias, size_t total4, size_t stride4, int splits,
                                                            int N, int act) {
    const int nq = N >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 s = reinterpret_cast<const f32x4*>(part)[i];
        for (int z = 1; z < splits; ++z) s += reinterpret_cast<const f32x4*>(part)[(size_t)z * stride4 + i];
        const int c4 = i % nq;
        if (bias) s += reinterpret_cast<const f32x4*>(bias)[c4];
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] = dwc_act_apply(s[k], act, c4 * 4 + k);
        reinterpret_cast<f32x4*>(dst)[i] = s;
    }
}

// ------------------------------------------------------------------------------------------
// weight gradient: dW[k][n] = sum_m A[m][k] * dY[m][n], split over m ("split-K") into slabs
// ------------------------------------------------------------------------------------------
// X3: the contraction over pixels as exact bf16 split products (see conv_gemm_body): both operands are read from the unchanged
// fp32 LDS tiles -- eight pixels of one k / n column per lane and 16-deep step, the same ds_read_b32 count as the fp32 MFMA path --
// and split in registers.
template <int BN, int WM, int WN, int TM, int TN, bool X3 = false>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(Gather g, const float* __restrict__ dy, int N, float* __restrict__ slab,
                                                         int m_chunk, int splits_per_class = 0, size_t src_class_stride = 0,
                                                         size_t dy_class_stride = 0) {
    static_assert(WM * WN == 4 && WM * TM * 32 == 128 && WN * TN * 32 == BN, "tile shape");
    DWC_PROBE();
    constexpr int A_TILE = 32 * 128, B_TILE = 32 * BN;
    __shared__ __attribute__((aligned(16))) float smem[2 * (A_TILE + B_TILE)];
    float* sA = smem;               // [buf][m][k]
    float* sB = smem + 2 * A_TILE;  // [buf][m][n]
    const int t = threadIdx.x;
    const int lane = t & 63, wave = t >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    const int k0 = blockIdx.x * 128, n0 = blockIdx.y * BN;
    int split = blockIdx.z;
    if (splits_per_class > 0) {       // batched form (Winograd): blockIdx.z = class * splits_per_class + split
        const int cls = blockIdx.z / splits_per_class;
        split = blockIdx.z - cls * splits_per_class;
        g.src = (const float*)g.src + (size_t)cls * src_class_stride;
        dy += (size_t)cls * dy_class_stride;
    }
    const int m_begin = split * m_chunk;
    const int m_end = min(g.M, m_begin + m_chunk);

    // A^T tile: thread owns one 4-wide k group (fixed tap / channel) and rows (t>>5)+8i.
    // k >= K (tile tail) is clamped to a valid tap: those output rows are never stored.
    const int kg = k0 + (t & 31) * 4;
    const int tap = min(kg >> g.logSC, g.KH * g.KW - 1);
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

    const int bcol_c = min(n0 + bcol, N - 4);    // columns past N are never stored: clamp instead of masking
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // Direct global->LDS staging (global_load_lds_dwordx4): lane t deposits its 16 bytes at float
    // offset 4*t of the pass, which IS [row][k] / [row][n] order for these unpadded tiles (the MFMA
    // operands are read along the lanes, so no swizzle is needed).  Rows past the end of this
    // m-chunk must contribute nothing: their dY operand comes from the zero page, the gathered x
    // operand may then be anything finite (row index clamped).
    auto stage_slab = [&](int mb, int buf) {
        float* la = sA + buf * A_TILE + wave_u * 256;
        float* lb = sB + buf * B_TILE + wave_u * 256;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = min(mb + arow + 8 * i, g.M - 1);
            int n, oh, ow;
            if (g.logOW >= 0) {                   // power-of-two pixel grid: shifts instead of divisions
                n = m >> g.logOHW;
                const int rem = m & (ohw - 1);
                oh = rem >> g.logOW;
                ow = rem & (g.OW - 1);
            } else {
                n = m / ohw;
                const int rem = m - n * ohw;
                oh = rem / g.OW;
                ow = rem - oh * g.OW;
            }
            const int h = reflect_idx(oh * g.mul_h + dh, g.SH);
            const int w = reflect_idx(ow * g.mul_w + dw, g.SW);
            const float* s = (const float*)g.src + (((n * g.SH + h) * g.SW + w) << g.logSC) + ci;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                             (__attribute__((address_space(3))) void*)(la + i * 8 * 128), 16, 0, 0);
        }
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p) {
            const int m = mb + brow + p * B_ROWS_PER_PASS;
            const float* s = m < m_end ? dy + (size_t)m * N + bcol_c : g_zero_page;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)s,
                                             (__attribute__((address_space(3))) void*)(lb + p * B_ROWS_PER_PASS * BN), 16, 0, 0);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // Same skewed pipeline as conv_gemm_kernel: a slab is four groups of four MFMA steps; the
    // operands of group g+1 are fetched from LDS while group g is multiplied, the registers->LDS
    // store of the next slab sits between groups 1 and 2, and group 3 (fetched before the barrier)
    // is multiplied after it while the next slab's group 0 is being fetched.
    float av[2][4][TM], bv[2][4][TN];
    auto load_ops = [&](int set, int buf, int grp) {
        const float* a = sA + buf * A_TILE + (wm * TM) * 32 + l31;
        const float* b = sB + buf * B_TILE + (wn * TN) * 32 + l31;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int ml = 2 * (4 * grp + s) + hi;
#pragma unroll
            for (int i = 0; i < TM; ++i) av[set][s][i] = a[ml * 128 + i * 32];
#pragma unroll
            for (int n = 0; n < TN; ++n) bv[set][s][n] = b[ml * BN + n * 32];
        }
    };
    auto mfma_ops = [&](int set) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int n = 0; n < TN; ++n)
                    acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[set][s][i], bv[set][s][n], acc[i][n], 0, 0, 0);
    };
    if constexpr (X3) {
        f32x16 lo[TM][TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) lo[i][j][r] = 0.f;
        gx3_bf16x8 pa[3][TM], pb[3][TN];
        auto ops_x3 = [&](int buf, int ss) {              // 16-pixel step ss of the slab: lane half hi takes pixels 16 ss + 8 hi .. + 7
            const float* a = sA + buf * A_TILE + (wm * TM) * 32 + l31 + (16 * ss + 8 * hi) * 128;
            const float* b = sB + buf * B_TILE + (wn * TN) * 32 + l31 + (16 * ss + 8 * hi) * BN;
#pragma unroll
            for (int i = 0; i < TM; ++i)
                gx3_split8(f32x4{a[i * 32], a[128 + i * 32], a[2 * 128 + i * 32], a[3 * 128 + i * 32]},
                           f32x4{a[4 * 128 + i * 32], a[5 * 128 + i * 32], a[6 * 128 + i * 32], a[7 * 128 + i * 32]}, pa[0][i], pa[1][i],
                           pa[2][i]);
#pragma unroll
            for (int n = 0; n < TN; ++n)
                gx3_split8(f32x4{b[n * 32], b[BN + n * 32], b[2 * BN + n * 32], b[3 * BN + n * 32]},
                           f32x4{b[4 * BN + n * 32], b[5 * BN + n * 32], b[6 * BN + n * 32], b[7 * BN + n * 32]}, pb[0][n], pb[1][n],
                           pb[2][n]);
        };
        auto mfma_x3 = [&]() {
            constexpr int TA[6] = {0, 1, 0, 2, 1, 0}, TB[6] = {0, 0, 1, 0, 1, 2};
#pragma unroll
            for (int term = 0; term < 6; ++term)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int n = 0; n < TN; ++n) {
                        if (term == 0) acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[TA[term]][i], pb[TB[term]][n], acc[i][n], 0, 0, 0);
                        else lo[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(pa[TA[term]][i], pb[TB[term]][n], lo[i][n], 0, 0, 0);
                    }
        };
        if (m_begin < m_end) {
            stage_slab(m_begin, 0);
            lds_dma_barrier();
            int buf = 0;
            for (int mb = m_begin; mb < m_end; mb += 32) {
                if (mb + 32 < m_end) stage_slab(mb + 32, buf ^ 1);
                ops_x3(buf, 0);
                mfma_x3();
                ops_x3(buf, 1);
                lds_dma_barrier();
                mfma_x3();
                buf ^= 1;
            }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] += lo[i][j][r];
    } else if (m_begin < m_end) {
        stage_slab(m_begin, 0);
        lds_dma_barrier();
        int buf = 0;
        load_ops(0, 0, 0);
        for (int mb = m_begin; mb < m_end; mb += 32) {
            const bool more = mb + 32 < m_end;
            if (more) stage_slab(mb + 32, buf ^ 1);      // other buffer: fully read before the last barrier
            load_ops(1, buf, 1);
            mfma_ops(0);
            load_ops(0, buf, 2);
            mfma_ops(1);
            load_ops(1, buf, 3);
            mfma_ops(0);
            lds_dma_barrier();
            if (more) load_ops(0, buf ^ 1, 0);
            mfma_ops(1);
            buf ^= 1;
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
__global__ void fold_reflect_kernel(const float* __restrict__ gp, float* __restrict__ dx, int B, int H, int W, int C4, int pad,
                                    int Wp) {   // Wp: row pitch of gp in pixels (>= W + 2*pad)
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t total = (size_t)B * H * W * C4;
    if (idx >= total) return;
    const int c = idx % C4;
    size_t r = idx / C4;
    const int w = r % W;
    r /= W;
    const int h = r % H;
    const int n = r / H;
    const int Hp = H + 2 * pad;
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

// dx (holds the interior of the padded gradient image already: Scatter::crop) += the border ring of gp folded back by the reflect
// rule; only the pixels a ring pixel folds onto are visited (rows 1..pad and H-1-pad..H-2 whole, columns 1..pad and W-1-pad..W-2
// of the other rows).
__global__ __launch_bounds__(256) void fold_band_kernel(const float* __restrict__ gp, float* __restrict__ dx, int B, int H, int W, int C4,
                                                                 int pad, int Wp) {
    // one thread per (image, band pixel, channel chunk): per image the 2*pad band rows whole (W pixels each), then the 2*pad band
    // columns of the H - 2*pad other rows
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int rows_done = 2 * pad, rest = H - 2 * pad;          // band rows, other rows
    const int band = rows_done * W + rest * 2 * pad;            // band pixels per image
    const size_t total = (size_t)B * band * C4;
    if (idx >= total) return;
    const int c = idx % C4;
    size_t r = idx / C4;
    const int q = r % band;
    const size_t n = r / band;
    const int Hp = H + 2 * pad;
    int h, w;
    if (q < rows_done * W) {
        const int br = q / W;
        w = q - br * W;
        h = br < pad ? 1 + br : H - 1 - pad + (br - pad);
    } else {
        const int q2 = q - rows_done * W;
        const int hr = q2 / (2 * pad), k = q2 - hr * 2 * pad;
        // the hr-th row that is NOT a band row: rows 0, pad+1 .. H-2-pad, H-1
        h = hr == 0 ? 0 : (hr == rest - 1 ? H - 1 : pad + hr);
        w = k < pad ? 1 + k : W - 1 - pad + (k - pad);
    }
    int hs[3], ws[3], nh = 0, nw = 0;
    hs[nh++] = h + pad;
    if (h >= 1 && h <= pad) hs[nh++] = pad - h;
    if (h >= H - 1 - pad && h <= H - 2) hs[nh++] = pad + 2 * (H - 1) - h;
    ws[nw++] = w + pad;
    if (w >= 1 && w <= pad) ws[nw++] = pad - w;
    if (w >= W - 1 - pad && w <= W - 2) ws[nw++] = pad + 2 * (W - 1) - w;
    if (nh * nw == 1) return;
    const size_t o = ((n * H + h) * (size_t)W + w) * C4 + c;
    f32x4 s = reinterpret_cast<const f32x4*>(dx)[o];
    const f32x4* g4 = reinterpret_cast<const f32x4*>(gp);
    for (int a = 0; a < nh; ++a)
        for (int b = 0; b < nw; ++b) {
            if (a == 0 && b == 0) continue;               // the pixel's own (interior) value is in dx already
            s += g4[((n * Hp + hs[a]) * (size_t)Wp + ws[b]) * C4 + c];
        }
    reinterpret_cast<f32x4*>(dx)[o] = s;
}

// Weight layouts streamed by conv_gemm_kernel: one row per GEMM column n, K contiguous and
// zero-padded to a multiple of 32 ("[N][Kp]"), so the B tile is staged exactly like the A tile.
//   forward : row = co, k = (kh*KW + kw)*cin_pad + ci                      value W[co][ci][kh][kw]
//   dgrad s1: row = ci, k = (kh'*KW + kw')*cout_pad + co                   value W[co][ci][KH-1-kh'][KW-1-kw']
//   dgrad s2: [class ph*2+pw] row = ci, k = (th*2 + tw)*cout_pad + co      value W[co][ci][ph+2th][pw+2tw]  (4x4 kernel)
__global__ void weight_prepare_fwd_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int KHW,
                                          int cout_pad, int cin_pad, int Kp) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)cout_pad * Kp) return;
    const int k = idx % Kp, co = idx / Kp;
    const int ci = k % cin_pad, tap = k / cin_pad;
    float v = 0.f;
    if (co < Cout && ci < Cin && tap < KHW) v = w[((size_t)co * Cin + ci) * KHW + tap];
    out[idx] = v;
}

__global__ void weight_prepare_dgrad_kernel(const float* __restrict__ w, float* __restrict__ out, int Cout, int Cin, int KH, int KW,
                                            int stride, int cout_pad, int cin_pad, int Kp) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t per_class = (size_t)cin_pad * Kp;
    const int classes = stride == 1 ? 1 : 4;
    if (idx >= per_class * classes) return;
    const int cls = idx / per_class;
    const size_t r = idx % per_class;
    const int k = r % Kp, ci = r / Kp;
    const int co = k % cout_pad, tapo = k / cout_pad;
    int kh, kw;
    bool ok = co < Cout && ci < Cin;
    if (stride == 1) {
        ok = ok && tapo < KH * KW;
        kh = KH - 1 - tapo / KW;
        kw = KW - 1 - tapo % KW;
    } else {
        ok = ok && tapo < 4;
        kh = (cls >> 1) + 2 * (tapo >> 1);
        kw = (cls & 1) + 2 * (tapo & 1);
    }
    out[idx] = ok ? w[((size_t)co * Cin + ci) * KH * KW + kh * KW + kw] : 0.f;
}



template <int BM, int BN, int WM, int WN, int TM, int TN, bool X3 = false>
void launch_variant(const Gather& g, const float* w, size_t wcs, int classes, const Scatter& o, const float* bias, int act,
                    const Plan& p, size_t part_stride, hipStream_t st) {
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (o.N + BN - 1) / BN;
    hipLaunchKernelGGL((conv_gemm_kernel<BM, BN, WM, WN, TM, TN, X3>), dim3(tiles_m * tiles_n, p.splits, classes), dim3(256), 0, st,
                       g, w, wcs, o, bias, act, tiles_n, p.kt_per_split, part_stride);
}

// 1: the im2col GEMM takes its inner product as split products on the bf16 matrix cores (conv_gemm_body<..., X3>): default for every
// product with K >= 128 (below that the launch is latency, not arithmetic).  DWC_X3_GEMM=0: the native fp32 MFMA everywhere.
// Process-wide switch of the split-product inner products: bit 0 the forward / data-gradient GEMM, bit 1 the weight-gradient
// kernel.  Initialised from DWC_X3_GEMM / DWC_X3_WGRAD (default on); dwc_x3_gemm_mode() reads / sets it at run time (tests use it
// to obtain the native fp32 MFMA result as a yardstick).
static std::atomic<int> g_x3_mode{-1};
static int x3_mode() {
    int m = g_x3_mode.load(std::memory_order_relaxed);
    if (m < 0) {
        const int a = getenv("DWC_X3_GEMM") ? atoi(getenv("DWC_X3_GEMM")) : 1;
        const int b = getenv("DWC_X3_WGRAD") ? atoi(getenv("DWC_X3_WGRAD")) : 1;
        m = (a ? 1 : 0) | (b ? 2 : 0);
        g_x3_mode.store(m, std::memory_order_relaxed);
    }
    return m;
}
static bool wgrad_x3_on() { return (x3_mode() & 2) != 0; }
static bool gemm_x3_on(int K) { return (x3_mode() & 1) != 0 && K >= 128; }

int launch_gemm(const Gather& g, const float* w, size_t w_class_stride, int classes, Scatter o, const float* bias, int act,
                size_t dst_elems, void* ws, size_t ws_bytes, hipStream_t st) {
    Plan p = plan_gemm(g.M, o.N, g.K, classes);
    float* final_dst = (float*)o.dst;
    size_t part_stride = 0;
    if (p.splits > 1) {
        if (!ws || ws_bytes < (size_t)p.splits * dst_elems * sizeof(float)) {
            p.splits = 1;   // not enough scratch: run un-split (slower, same result up to summation order)
            p.kt_per_split = (g.K + BK - 1) / BK;
        } else {
            o.dst = (float*)ws;
            part_stride = dst_elems;
        }
    }
    if (gemm_x3_on(g.K)) {
        // (split accumulators need the registers of two tiles: the 128x128 plan runs as 128x64 with twice the column tiles)
        if (p.bm == 128 && p.bn >= 64) launch_variant<128, 64, 2, 2, 2, 1, true>(g, w, w_class_stride, classes, o, bias, act, p, part_stride, st);
        else if (p.bm == 64 && p.bn == 64) launch_variant<64, 64, 2, 2, 1, 1, true>(g, w, w_class_stride, classes, o, bias, act, p, part_stride, st);
        else launch_variant<128, 32, 4, 1, 1, 1, true>(g, w, w_class_stride, classes, o, bias, act, p, part_stride, st);
    } else if (p.bm == 128 && p.bn == 128) launch_variant<128, 128, 2, 2, 2, 2>(g, w, w_class_stride, classes, o, bias, act, p, part_stride, st);
    else if (p.bm == 128 && p.bn == 64) launch_variant<128, 64, 2, 2, 2, 1>(g, w, w_class_stride, classes, o, bias, act, p, part_stride, st);
    else if (p.bm == 64 && p.bn == 64) launch_variant<64, 64, 2, 2, 1, 1>(g, w, w_class_stride, classes, o, bias, act, p, part_stride, st);
    else launch_variant<128, 32, 4, 1, 1, 1>(g, w, w_class_stride, classes, o, bias, act, p, part_stride, st);
    DWC_LAUNCH_CHECK();
    if (p.splits > 1) {
        const size_t total4 = dst_elems / 4;
        size_t blocks = (total4 + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)ws, final_dst, bias, total4,
                           total4, p.splits, o.N, act);
        DWC_LAUNCH_CHECK();
    }
    return DWC_OK;
}


}  // namespace

extern "C" {

int dwc_version(void) { return DWC_ABI_VERSION; }

/* mode >= 0: set the split-product switch of the im2col kernels (bit 0: forward / data-gradient GEMM, bit 1: weight gradient;
 * 0 = native fp32 MFMA everywhere); returns the previous value.  mode < 0: query only. */
int dwc_x3_gemm_mode(int mode) {
    const int prev = x3_mode();
    if (mode >= 0) g_x3_mode.store(mode & 3, std::memory_order_relaxed);
    return prev;
}

size_t dwc_weight_prepared_elems(int Cout, int Cin, int KH, int KW, int stride, int cout_pad, int cin_pad, int for_dgrad) {
    if (!for_dgrad) return (size_t)cout_pad * ((KH * KW * cin_pad + BK - 1) / BK * BK);
    if (stride == 1) return (size_t)cin_pad * ((KH * KW * cout_pad + BK - 1) / BK * BK);
    return (size_t)4 * cin_pad * ((4 * cout_pad + BK - 1) / BK * BK);
}

int dwc_weight_prepare_fwd(const float* w, float* out, int Cout, int Cin, int KH, int KW, int cout_pad, int cin_pad,
                           void* stream) {
    if (cout_pad < Cout || cin_pad < Cin) return DWC_EINVAL;
    const int Kp = (KH * KW * cin_pad + BK - 1) / BK * BK;
    const size_t total = (size_t)cout_pad * Kp;
    hipLaunchKernelGGL(weight_prepare_fwd_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, out, Cout, Cin,
                       KH * KW, cout_pad, cin_pad, Kp);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

int dwc_weight_prepare_dgrad(const float* w, float* out, int Cout, int Cin, int KH, int KW, int stride, int cout_pad,
                             int cin_pad, void* stream) {
    if (cout_pad < Cout || cin_pad < Cin) return DWC_EINVAL;
    if (stride == 2 && !(KH == 4 && KW == 4)) return DWC_EINVAL;
    if (stride != 1 && stride != 2) return DWC_EINVAL;
    const int Kp = ((stride == 1 ? KH * KW : 4) * cout_pad + BK - 1) / BK * BK;
    const size_t total = (size_t)(stride == 1 ? 1 : 4) * cin_pad * Kp;
    hipLaunchKernelGGL(weight_prepare_dgrad_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, out, Cout,
                       Cin, KH, KW, stride, cout_pad, cin_pad, Kp);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

size_t dwc_conv2d_fwd_ws_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    FwdGeom f;
    if (!fwd_geom(nullptr, nullptr, B, H, W, Cin, Cout, KH, KW, stride, pad, &f)) return 0;
    return gemm_ws_bytes(f.g.M, Cout, f.g.K, 1, f.dst_elems);
}

int dwc_conv2d_fwd(const float* x, const float* w_hwio, const float* bias, float* y, int B, int H, int W, int Cin, int Cout,
                   int KH, int KW, int stride, int pad, int act, void* ws, size_t ws_bytes, void* stream) {
    FwdGeom f;
    if (!fwd_geom(x, y, B, H, W, Cin, Cout, KH, KW, stride, pad, &f)) return DWC_EINVAL;
    return launch_gemm(f.g, w_hwio, 0, 1, f.o, bias, act, f.dst_elems, ws, ws_bytes, (hipStream_t)stream);
}

// Zero-padded convolutions (the frozen VGG16 of the perceptual loss, reference networks.py:639-688: nn.Conv2d(padding=1)):
// the forward kernel with the zero rule instead of the reflect rule, and as data gradient the zero-padded correlation
// with the flipped filter on the H x W grid (the adjoint of zero padding is a crop: nothing to fold).
int dwc_conv2d_fwd_zeropad(const float* x, const float* w_prepared, const float* bias, float* y, int B, int H, int W, int Cin,
                           int Cout, int KH, int KW, int stride, int pad, int act, void* ws, size_t ws_bytes, void* stream) {
    FwdGeom f;
    if (!fwd_geom(x, y, B, H, W, Cin, Cout, KH, KW, stride, pad, &f)) return DWC_EINVAL;
    f.g.reflect = 0;
    return launch_gemm(f.g, w_prepared, 0, 1, f.o, bias, act, f.dst_elems, ws, ws_bytes, (hipStream_t)stream);
}


size_t dwc_conv2d_bwd_data_zeropad_ws_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW, int pad) {
    FwdGeom f;
    if (!zeropad_dgrad_geom(nullptr, nullptr, B, H, W, Cin, Cout, KH, KW, pad, &f)) return 0;
    return gemm_ws_bytes(f.g.M, Cin, f.g.K, 1, f.dst_elems);
}

int dwc_conv2d_bwd_data_zeropad(const float* dy, const float* w_dgrad, float* dx, int B, int H, int W, int Cin, int Cout, int KH,
                                int KW, int pad, void* ws, size_t ws_bytes, void* stream) {
    FwdGeom f;
    if (!zeropad_dgrad_geom(dy, dx, B, H, W, Cin, Cout, KH, KW, pad, &f)) return DWC_EINVAL;
    return launch_gemm(f.g, w_dgrad, 0, 1, f.o, nullptr, DWC_ACT_NONE, f.dst_elems, ws, ws_bytes, (hipStream_t)stream);
}

// forward / weight gradient with per-axis stride and reflect pad (no scratch on the forward: never split)
int dwc_conv2d_fwd_ex(const float* x, const float* w_prepared, const float* bias, float* y, int B, int H, int W, int Cin,
                      int Cout, int KH, int KW, int stride_h, int stride_w, int pad_h, int pad_w, int act, void* stream) {
    FwdGeom f;
    if (!fwd_geom_ex(x, y, B, H, W, Cin, Cout, KH, KW, stride_h, stride_w, pad_h, pad_w, &f)) return DWC_EINVAL;
    return launch_gemm(f.g, w_prepared, 0, 1, f.o, bias, act, f.dst_elems, nullptr, 0, (hipStream_t)stream);
}

static int wgrad_launch(const FwdGeom& f, const float* dy, float* dw_oihw, int Cin, int Cout, int KHW, int cin_real, int cout_real,
                        void* ws, size_t ws_bytes, hipStream_t st) {
    const Gather& g = f.g;
    int splits, chunk;
    wgrad_plan(g.M, g.K, Cout, &splits, &chunk);
    if (!ws || ws_bytes < (size_t)splits * g.K * Cout * sizeof(float)) return DWC_EWORKSPACE;
    float* slab = (float*)ws;
    const int tk = (g.K + 127) / 128;
    if (wgrad_x3_on()) {
        if (Cout > 64)
            hipLaunchKernelGGL((conv_wgrad_kernel<128, 2, 2, 2, 2, true>), dim3(tk, (Cout + 127) / 128, splits), dim3(256), 0, st, g, dy,
                               Cout, slab, chunk);
        else if (Cout > 32)
            hipLaunchKernelGGL((conv_wgrad_kernel<64, 2, 2, 2, 1, true>), dim3(tk, 1, splits), dim3(256), 0, st, g, dy, Cout, slab, chunk);
        else
            hipLaunchKernelGGL((conv_wgrad_kernel<32, 4, 1, 1, 1, true>), dim3(tk, 1, splits), dim3(256), 0, st, g, dy, Cout, slab, chunk);
    } else if (Cout > 64) {
        hipLaunchKernelGGL((conv_wgrad_kernel<128, 2, 2, 2, 2>), dim3(tk, (Cout + 127) / 128, splits), dim3(256), 0, st, g, dy,
                           Cout, slab, chunk);
    } else if (Cout > 32) {
        hipLaunchKernelGGL((conv_wgrad_kernel<64, 2, 2, 2, 1>), dim3(tk, 1, splits), dim3(256), 0, st, g, dy, Cout, slab, chunk);
    } else {
        hipLaunchKernelGGL((conv_wgrad_kernel<32, 4, 1, 1, 1>), dim3(tk, 1, splits), dim3(256), 0, st, g, dy, Cout, slab, chunk);
    }
    DWC_LAUNCH_CHECK();
    const size_t total = (size_t)g.K * Cout;
    if (!wgrad_reduce_wide(slab, dw_oihw, splits, g.K, Cout, Cin, KHW, cin_real, cout_real, st))
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, slab, dw_oihw, splits, g.K, Cout, Cin, KHW,
                           cin_real, cout_real);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

size_t dwc_conv2d_bwd_weight_ex_ws_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride_h, int stride_w,
                                         int pad_h, int pad_w) {
    FwdGeom f;
    if (!fwd_geom_ex(nullptr, nullptr, B, H, W, Cin, Cout, KH, KW, stride_h, stride_w, pad_h, pad_w, &f)) return 0;
    int splits, chunk;
    wgrad_plan(f.g.M, f.g.K, Cout, &splits, &chunk);
    return (size_t)splits * f.g.K * Cout * sizeof(float);
}

int dwc_conv2d_bwd_weight_ex(const float* x, const float* dy, float* dw_oihw, int B, int H, int W, int Cin, int Cout, int KH,
                             int KW, int stride_h, int stride_w, int pad_h, int pad_w, int cin_real, int cout_real, void* ws,
                             size_t ws_bytes, void* stream) {
    FwdGeom f;
    if (!fwd_geom_ex(x, nullptr, B, H, W, Cin, Cout, KH, KW, stride_h, stride_w, pad_h, pad_w, &f)) return DWC_EINVAL;
    if (cin_real > Cin || cout_real > Cout) return DWC_EINVAL;
    return wgrad_launch(f, dy, dw_oihw, Cin, Cout, KH * KW, cin_real, cout_real, ws, ws_bytes, (hipStream_t)stream);
}

size_t dwc_conv2d_bwd_data_ws_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    BwdGeom f;
    if (!bwd_geom(nullptr, nullptr, B, H, W, Cin, Cout, KH, KW, stride, pad, &f)) return 0;
    return gemm_ws_bytes(f.g.M, Cin, f.g.K, f.classes, f.dst_elems);
}

int dwc_reflect_pad_adjoint(const float* dxp, float* dx, int B, int H, int W, int C, int pad, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || pad < 0 || pad >= H || pad >= W) return DWC_EINVAL;
    const size_t total = (size_t)B * H * W * (C / 4);
    hipLaunchKernelGGL(fold_reflect_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, dxp, dx, B, H, W, C / 4,
                       pad, W + 2 * pad);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

/* dx (already holding the interior of the padded gradient image dxp [B][H+2pad][W+2pad][C]) += the border ring of dxp folded back by
 * the reflect rule; only the band of dx a ring pixel folds onto is visited (fp32 twin of dwc_bf16_reflect_pad_adjoint_band). */
int dwc_reflect_pad_adjoint_band(const float* dxp, float* dx, int B, int H, int W, int C, int pad, void* stream) {
    if (!dxp || !dx || B <= 0 || C <= 0 || (C & 3) || pad <= 0 || H < 2 * pad + 2 || W < 2 * pad + 2) return DWC_EINVAL;
    const int C4 = C / 4;
    const size_t band_items = (size_t)B * (2 * pad * W + (H - 2 * pad) * 2 * pad) * C4;
    hipLaunchKernelGGL(fold_band_kernel, dim3((unsigned)((band_items + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dxp, dx, B, H, W, C4, pad,
                       W + 2 * pad);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

/* The same for a padded gradient image whose rows are `pitch` >= W + 2*pad pixels apart (the 8-pixel-group grid of
 * dwc_x3_conv2d_narrow's image gradient). */
int dwc_reflect_pad_adjoint_pitch(const float* dxp, float* dx, int B, int H, int W, int C, int pad, int pitch, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || pad < 0 || pad >= H || pad >= W || pitch < W + 2 * pad) return DWC_EINVAL;
    const size_t total = (size_t)B * H * W * (C / 4);
    hipLaunchKernelGGL(fold_reflect_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, dxp, dx, B, H, W, C / 4,
                       pad, pitch);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

// dx += the border ring of the padded gradient image, folded back by the reflect rule.  dx already holds the interior;
// the ring lives in four strips: top/bottom [B][pad][Wp][C], left/right [B][H][pad][C], `parts` copies `part_stride`
// apart (partial sums over K, added in order).  Only the bands of dx that receive something are visited: per image
// 2*pad rows x W pixels (rows 1..pad, H-1-pad..H-2) then 2*pad columns x H pixels (skipping the rows already done).
__global__ void fold_ring_kernel(float* __restrict__ dx, const float* __restrict__ ring, size_t off_bottom, size_t off_left,
                                 size_t off_right, int parts, size_t part_stride, int B, int H, int W, int C4, int pad) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int band = 2 * pad * (W + H);
    const size_t total = (size_t)B * band * C4;
    if (idx >= total) return;
    const int c = idx % C4;
    size_t r = idx / C4;
    const int q = r % band;
    const int n = r / band;
    int h, w;
    if (q < 2 * pad * W) {
        const int br = q / W;
        w = q - br * W;
        h = br < pad ? 1 + br : H - 1 - pad + (br - pad);
    } else {
        const int q2 = q - 2 * pad * W, bc = q2 / H;
        h = q2 - bc * H;
        w = bc < pad ? 1 + bc : W - 1 - pad + (bc - pad);
        if ((h >= 1 && h <= pad) || (h >= H - 1 - pad && h <= H - 2)) return;   // covered by the row bands
    }
    if (h < 0 || h >= H || w < 0 || w >= W) return;
    const int Wp = W + 2 * pad;
    int hs[3], ws[3], nh = 0, nw = 0;
    hs[nh++] = h + pad;
    if (h >= 1 && h <= pad) hs[nh++] = pad - h;
    if (h >= H - 1 - pad && h <= H - 2) hs[nh++] = pad + 2 * (H - 1) - h;
    ws[nw++] = w + pad;
    if (w >= 1 && w <= pad) ws[nw++] = pad - w;
    if (w >= W - 1 - pad && w <= W - 2) ws[nw++] = pad + 2 * (W - 1) - w;
    f32x4* out = reinterpret_cast<f32x4*>(dx) + ((size_t)(n * H + h) * W + w) * C4 + c;
    f32x4 s = *out;
    for (int a = 0; a < nh; ++a)
        for (int b = 0; b < nw; ++b) {
            if (a == 0 && b == 0) continue;
            const int rh = hs[a], rw = ws[b];
            size_t e;   // element offset inside one part's ring
            if (rh < pad) e = ((size_t)(n * pad + rh) * Wp + rw) * C4;
            else if (rh >= pad + H) e = off_bottom / 4 + ((size_t)(n * pad + rh - pad - H) * Wp + rw) * C4;
            else if (rw < pad) e = off_left / 4 + ((size_t)(n * H + rh - pad) * pad + rw) * C4;
            else e = off_right / 4 + ((size_t)(n * H + rh - pad) * pad + rw - pad - W) * C4;
            for (int p = 0; p < parts; ++p) s += reinterpret_cast<const f32x4*>(ring + p * part_stride)[e + c];
        }
    *out = s;
}


size_t dwc_conv2d_bwd_data_same_ws_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW, int pad) {
    SameDgrad f;
    const int bm = strip_bm((long)B * pad * max(W + 2 * pad, H), (Cin + 63) / 64, 4, 1, false);
    if (!same_dgrad_geom(nullptr, nullptr, nullptr, nullptr, nullptr, B, H, W, Cin, Cout, KH, KW, pad, &f, 32, bm)) return 0;
    const size_t ring = (f.ring_total * f.parts * sizeof(float) + 255) / 256 * 256;
    return ring + gemm_ws_bytes(f.g.M, Cin, f.g.K, 1, f.dst_elems);
}

static int same_dgrad_run(const float* dy, const float* w_dgrad, const float* w_dgrad_t, float* dx, int B, int H, int W, int Cin,
                          int Cout, int KH, int KW, int pad, void* ws, size_t ws_bytes, void* stream, bool ring_only);

int dwc_conv2d_bwd_data_same(const float* dy, const float* w_dgrad, const float* w_dgrad_t, float* dx, int B, int H, int W,
                             int Cin, int Cout, int KH, int KW, int pad, void* ws, size_t ws_bytes, void* stream) {
    return same_dgrad_run(dy, w_dgrad, w_dgrad_t, dx, B, H, W, Cin, Cout, KH, KW, pad, ws, ws_bytes, stream, false);
}

// Only the border ring: dx must already hold the interior (e.g. from dwc_conv2d_wino with the zero rule).
int dwc_conv2d_bwd_data_ring(const float* dy, const float* w_dgrad, const float* w_dgrad_t, float* dx, int B, int H, int W,
                             int Cin, int Cout, int KH, int KW, int pad, void* ws, size_t ws_bytes, void* stream) {
    return same_dgrad_run(dy, w_dgrad, w_dgrad_t, dx, B, H, W, Cin, Cout, KH, KW, pad, ws, ws_bytes, stream, true);
}

static int same_dgrad_run(const float* dy, const float* w_dgrad, const float* w_dgrad_t, float* dx, int B, int H, int W, int Cin,
                          int Cout, int KH, int KW, int pad, void* ws, size_t ws_bytes, void* stream, bool ring_only) {
    SameDgrad f;
    const int bm = strip_bm((long)B * pad * max(W + 2 * pad, H), (Cin + 63) / 64, 4, 1, false);
    if (!same_dgrad_geom(dy, w_dgrad, w_dgrad_t, dx, (float*)ws, B, H, W, Cin, Cout, KH, KW, pad, &f, 32, bm)) return DWC_EINVAL;
    const size_t ring_bytes = (f.ring_total * f.parts * sizeof(float) + 255) / 256 * 256;
    if (!ws || ws_bytes < ring_bytes) return DWC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (!ring_only) {
        int rc = launch_gemm(f.g, w_dgrad, 0, 1, f.o, nullptr, DWC_ACT_NONE, f.dst_elems, (char*)ws + ring_bytes,
                             ws_bytes - ring_bytes, st);
        if (rc != DWC_OK) return rc;
    }
    const dim3 sgrid(f.max_tiles, f.parts, 4);
    if (gemm_x3_on(f.ss.s[0].g.K)) {
        if (bm == 128) hipLaunchKernelGGL((conv_gemm_strips_kernel<128, 64, 2, 2, 2, 1, true>), sgrid, dim3(256), 0, st, f.ss);
        else hipLaunchKernelGGL((conv_gemm_strips_kernel<64, 64, 2, 2, 1, 1, true>), sgrid, dim3(256), 0, st, f.ss);
    } else {
        if (bm == 128) hipLaunchKernelGGL((conv_gemm_strips_kernel<128, 64, 2, 2, 2, 1>), sgrid, dim3(256), 0, st, f.ss);
        else hipLaunchKernelGGL((conv_gemm_strips_kernel<64, 64, 2, 2, 1, 1>), sgrid, dim3(256), 0, st, f.ss);
    }
    DWC_LAUNCH_CHECK();
    const size_t total = (size_t)B * 2 * pad * (W + H) * (Cin / 4);
    hipLaunchKernelGGL(fold_ring_kernel, dim3((total + 255) / 256), dim3(256), 0, st, dx, (const float*)ws, f.ring_elems[0],
                       f.ring_elems[0] + f.ring_elems[1], f.ring_elems[0] + f.ring_elems[1] + f.ring_elems[2], f.parts,
                       f.ring_total, B, H, W, Cin / 4, pad);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}


size_t dwc_conv2d_bwd_data_image_ws_bytes(int B, int H, int W, int Cout, int KH, int KW, int pad) {
    FwdGeom f;
    if (!image_dgrad_geom(nullptr, nullptr, B, H, W, Cout, KH, KW, pad, &f)) return 0;
    return f.dst_elems * sizeof(float);
}

int dwc_conv2d_bwd_data_image(const float* dy, const float* w_wide, float* dx, int B, int H, int W, int Cout, int KH, int KW,
                              int pad, void* ws, size_t ws_bytes, void* stream) {
    FwdGeom f;
    if (!image_dgrad_geom(dy, (float*)ws, B, H, W, Cout, KH, KW, pad, &f)) return DWC_EINVAL;
    if (!ws || ws_bytes < f.dst_elems * sizeof(float)) return DWC_EWORKSPACE;
    const int rc = launch_gemm(f.g, w_wide, 0, 1, f.o, nullptr, DWC_ACT_NONE, f.dst_elems, nullptr, 0, (hipStream_t)stream);
    if (rc != DWC_OK) return rc;
    const size_t total = (size_t)B * H * W;
    hipLaunchKernelGGL(fold_reflect_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float*)ws, dx, B, H,
                       W, 1, pad, f.g.OW * 8);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

int dwc_conv2d_bwd_data(const float* dy, const float* w_dgrad, float* dxp, int B, int H, int W, int Cin, int Cout, int KH,
                        int KW, int stride, int pad, void* ws, size_t ws_bytes, void* stream) {
    BwdGeom f;
    if (!bwd_geom(dy, dxp, B, H, W, Cin, Cout, KH, KW, stride, pad, &f)) return DWC_EINVAL;
    return launch_gemm(f.g, w_dgrad, f.wcs, f.classes, f.o, nullptr, DWC_ACT_NONE, f.dst_elems, ws, ws_bytes, (hipStream_t)stream);
}

/* dwc_conv2d_bwd_data + dwc_reflect_pad_adjoint as one call (see dwc_bf16_conv2d_bwd_data_fold): where the GEMM runs unsplit the
 * interior of the padded gradient image goes straight into dx, only its border ring through the scratch image dxp, folded back by a
 * band kernel. */
int dwc_conv2d_bwd_data_fold(const float* dy, const float* w_dgrad, float* dxp, float* dx, int B, int H, int W, int Cin, int Cout, int KH,
                             int KW, int stride, int pad, void* ws, size_t ws_bytes, void* stream) {
    BwdGeom f;
    if ((Cin & 3) || pad <= 0 || !bwd_geom(dy, dxp, B, H, W, Cin, Cout, KH, KW, stride, pad, &f)) return DWC_EINVAL;
    if (H < 2 * pad + 2 || W < 2 * pad + 2 || H > 65535 - 2 * pad || B > 65535) return DWC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const Plan p = plan_gemm(f.g.M, f.o.N, f.g.K, f.classes);
    const bool direct = p.splits == 1;
    if (direct) {
        f.o.crop = pad; f.o.IH = H; f.o.IW = W; f.o.inner = dx;
    }
    int rc = launch_gemm(f.g, w_dgrad, f.wcs, f.classes, f.o, nullptr, DWC_ACT_NONE, f.dst_elems, ws, ws_bytes, st);
    if (rc != DWC_OK) return rc;
    if (!direct) return dwc_reflect_pad_adjoint(dxp, dx, B, H, W, Cin, pad, stream);
    const int C4 = Cin / 4;
    const size_t band_items = (size_t)B * (2 * pad * W + (H - 2 * pad) * 2 * pad) * C4;
    hipLaunchKernelGGL(fold_band_kernel, dim3((unsigned)((band_items + 255) / 256)), dim3(256), 0, st, (const float*)dxp, dx, B, H, W, C4, pad,
                       W + 2 * pad);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

/* Border ring + fold of the data gradient of a 4x4 stride-2 reflect-pad-1 convolution whose INTERIOR (the H x W pixels of dx)
 * has been written by a halo-tiled kernel (dwc_x3_conv2d_s2_bwd_data): the ring of the padded gradient image is computed as
 * eight thin strips into the scratch image dxp ([B][H+2][W+2][Cin], only its ring is touched) and folded onto dx by the
 * reflect rule (fold_band_kernel).  w_dgrad: the stride-2 data-gradient layout of dwc_weight_prepare_dgrad. */
int dwc_conv2d_bwd_data_s2_ring(const float* dy, const float* w_dgrad, float* dxp, float* dx, int B, int H, int W, int Cin, int Cout,
                                void* stream) {
    S2Ring f;
    const int bm = strip_bm((long)B * max(W / 2 + 1, H / 2), (Cin + 63) / 64, 8, 1, false);
    if (!dy || !w_dgrad || !dxp || !dx || H > 65535 - 2 || B > 65535 ||
        !s2_ring_geom(dy, w_dgrad, dxp, sizeof(float), B, H, W, Cin, Cout, &f, 32, 2, bm))
        return DWC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const dim3 sgrid(f.max_tiles, 1, 8);
    if (gemm_x3_on(f.ss.s[0].g.K)) {
        if (bm == 128) hipLaunchKernelGGL((conv_gemm_strips_kernel<128, 64, 2, 2, 2, 1, true>), sgrid, dim3(256), 0, st, f.ss);
        else hipLaunchKernelGGL((conv_gemm_strips_kernel<64, 64, 2, 2, 1, 1, true>), sgrid, dim3(256), 0, st, f.ss);
    } else {
        if (bm == 128) hipLaunchKernelGGL((conv_gemm_strips_kernel<128, 64, 2, 2, 2, 1>), sgrid, dim3(256), 0, st, f.ss);
        else hipLaunchKernelGGL((conv_gemm_strips_kernel<64, 64, 2, 2, 1, 1>), sgrid, dim3(256), 0, st, f.ss);
    }
    DWC_LAUNCH_CHECK();
    const int C4 = Cin / 4;
    const size_t band_items = (size_t)B * (2 * W + (H - 2) * 2) * C4;
    hipLaunchKernelGGL(fold_band_kernel, dim3((unsigned)((band_items + 255) / 256)), dim3(256), 0, st, (const float*)dxp, dx, B, H, W, C4, 1,
                       W + 2);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

size_t dwc_conv2d_bwd_weight_ws_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    int splits, chunk;
    wgrad_plan(B * Ho * Wo, KH * KW * Cin, Cout, &splits, &chunk);
    return (size_t)splits * KH * KW * Cin * Cout * sizeof(float);
}

int dwc_conv2d_bwd_weight(const float* x, const float* dy, float* dw_oihw, int B, int H, int W, int Cin, int Cout, int KH,
                          int KW, int stride, int pad, int cin_real, int cout_real, void* ws, size_t ws_bytes, void* stream) {
    FwdGeom f;
    if (!fwd_geom(x, nullptr, B, H, W, Cin, Cout, KH, KW, stride, pad, &f)) return DWC_EINVAL;
    if (cin_real > Cin || cout_real > Cout) return DWC_EINVAL;
    if (!ws || ws_bytes < dwc_conv2d_bwd_weight_ws_bytes(B, H, W, Cin, Cout, KH, KW, stride, pad)) return DWC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    const Gather& g = f.g;
    int splits, chunk;
    wgrad_plan(g.M, g.K, Cout, &splits, &chunk);
    float* slab = (float*)ws;
    const int tk = (g.K + 127) / 128;
    if (wgrad_x3_on()) {
        if (Cout > 64)
            hipLaunchKernelGGL((conv_wgrad_kernel<128, 2, 2, 2, 2, true>), dim3(tk, (Cout + 127) / 128, splits), dim3(256), 0, st, g, dy,
                               Cout, slab, chunk);
        else if (Cout > 32)
            hipLaunchKernelGGL((conv_wgrad_kernel<64, 2, 2, 2, 1, true>), dim3(tk, 1, splits), dim3(256), 0, st, g, dy, Cout, slab, chunk);
        else
            hipLaunchKernelGGL((conv_wgrad_kernel<32, 4, 1, 1, 1, true>), dim3(tk, 1, splits), dim3(256), 0, st, g, dy, Cout, slab, chunk);
    } else if (Cout > 64) {
        hipLaunchKernelGGL((conv_wgrad_kernel<128, 2, 2, 2, 2>), dim3(tk, (Cout + 127) / 128, splits), dim3(256), 0, st, g, dy,
                           Cout, slab, chunk);
    } else if (Cout > 32) {
        hipLaunchKernelGGL((conv_wgrad_kernel<64, 2, 2, 2, 1>), dim3(tk, 1, splits), dim3(256), 0, st, g, dy, Cout, slab, chunk);
    } else {
        hipLaunchKernelGGL((conv_wgrad_kernel<32, 4, 1, 1, 1>), dim3(tk, 1, splits), dim3(256), 0, st, g, dy, Cout, slab, chunk);
    }
    DWC_LAUNCH_CHECK();
    const size_t total = (size_t)g.K * Cout;
    if (!wgrad_reduce_wide(slab, dw_oihw, splits, g.K, Cout, Cin, KH * KW, cin_real, cout_real, st))
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, slab, dw_oihw, splits, g.K, Cout, Cin,
                           KH * KW, cin_real, cout_real);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

}  // extern "C"

#ifdef DWC_CLOCK_PROBE
extern "C" int dwc_debug_clock_probe(unsigned long long* out, int reset) {
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_clock_probe), 16) != hipSuccess) return DWC_ELAUNCH;
    if (reset) {
        const unsigned long long z[2] = {0, 0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_clock_probe), z, 16) != hipSuccess) return DWC_ELAUNCH;
    }
    return 0;
}
#endif
