// bf16-activation implicit-GEMM convolution on the bf16 matrix cores of gfx950 (v_mfma_f32_32x32x16_bf16):
// BASELINE configs[2] ("128x128 batch=128, bf16 activations + MFMA im2col conv path").
//
// Same formulation as conv_igemm.hip (reference networks.py:579-585: reflect pad + conv + bias + activation, and its
// autograd; geometry shared through conv_geom.h), with
//   * activations, prepared weights and gradients of activations in bf16 (NHWC, 16-byte chunks = 8 channels),
//   * fp32 accumulation in the MFMA, fp32 bias / activation epilogue, ONE rounding to bf16 at the store,
//   * fp32 master weights (re-laid-out and rounded once per optimiser step by dwc_bf16_weight_prepare_*), fp32 weight
//     gradients (split partial sums in fp32, reduced in a fixed order), fp32 split-K / ring partials.
// A K-slab is 64 elements = 128 bytes per tile row, i.e. the LDS image, the direct global->LDS staging and the XOR
// swizzle are byte-for-byte those of the fp32 kernel; one ds_read_b128 is one 32x32x16 operand fragment.
// The product is taken transposed (D = W_tile . X_tile^T): a lane then owns one output PIXEL and its 16 accumulator
// registers run over channels in groups of 4, so partials leave as 16-byte fp32 stores and final tiles are packed
// to bf16, staged through LDS and written as whole 16-byte chunks of a pixel's channel row.
// The weight gradient contracts over pixels, which is the slow axis of both operands: its LDS tiles stay row-major
// ([pixel][channel], coalesced from HBM) and the MFMA operands are fetched with the transposing LDS read
// ds_read_b64_tr_b16 (a 4-pixel x 16-channel block per 16 lanes, delivered channel-major).
#include <type_traits>

#include "conv_geom.h"

namespace {

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 64;          // elements per K-slab (128 bytes per tile row)
constexpr int MIN_LOG_C = 3;    // a 16-byte staging chunk is 8 channels

__device__ __forceinline__ void lds_dma_barrier() {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

__device__ __forceinline__ bf16x4 pack4(float a, float b, float c, float d) {
    bf16x4 r;
    r[0] = (bf16)a; r[1] = (bf16)b; r[2] = (bf16)c; r[3] = (bf16)d;
    return r;
}

// ------------------------------------------------------------------------------------------
// forward / data-gradient GEMM body.  F32OUT: the destination is fp32 (split-K partials, ring strips).
// ------------------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int TM, int TN, bool F32OUT, int STAGES = 2>
__device__ __forceinline__ void gemm_body_h(const Gather& g, const bf16* __restrict__ wmat, const Scatter& o,
                                            const float* __restrict__ bias, int act, int tiles_n, int kt0, int kt1_in,
                                            size_t part_offset, bool partial, int oph, int opw, int bid, int nb) {
#if defined(__HIP_DEVICE_COMPILE__)      // the buffer-resource builtins exist for the device pass only
    constexpr int NW = WM * WN;                       // waves per workgroup: 4 (256 threads) or 8 (512 threads, 256-row tiles)
    constexpr int RP = 8 * NW;                        // tile rows staged per pass (one wave instruction = 8 rows x 128 bytes)
    static_assert((NW == 4 || NW == 8) && WM * TM * 32 == BM && WN * TN * 32 == BN && BM % RP == 0 && BN % RP == 0, "tile shape");
    constexpr int A_PASSES = BM / RP, B_PASSES = BN / RP;
    constexpr int A_TILE = BM * BK, B_TILE = BN * BK;
    constexpr int LDC = BN + 8;                       // epilogue staging pitch (elements)
    constexpr int SMEM = STAGES * (A_TILE + B_TILE) > BM * LDC ? STAGES * (A_TILE + B_TILE) : BM * LDC;     // the epilogue staging re-uses it
    __shared__ __attribute__((aligned(16))) bf16 smem[SMEM];
    bf16* sA = smem;
    bf16* sB = smem + STAGES * A_TILE;

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

    const int arow = t >> 3;                               // tile row this lane stages (+RP per pass)
    const int acol = (((t & 7) ^ ((arow >> 1) & 7))) * 8;  // LOGICAL k offset fetched into physical chunk t&7
    int a_bh[A_PASSES], a_bw[A_PASSES], a_img[A_PASSES];
    const int ohw = g.OH * g.OW;
#pragma unroll
    for (int i = 0; i < A_PASSES; ++i) {
        const int mm = min(m0 + arow + RP * i, g.M - 1);
        const int n = mm / ohw;
        const int rem = mm - n * ohw;
        const int oh = rem / g.OW, ow = rem - oh * g.OW;
        a_bh[i] = oh * g.mul_h + g.off_h;
        a_bw[i] = ow * g.mul_w + g.off_w;
        a_img[i] = n * g.SH * g.SW;
    }
    const int Kp = (g.K + BK - 1) / BK * BK;
    // Staging through BUFFER loads into LDS (buffer_load_dwordx4 ... lds): a wave-uniform resource descriptor per operand,
    // ONE 32-bit byte offset per lane and row that changes only when the slab crosses a filter tap, and the wave-uniform
    // part of the address (channel offset inside the tap / K offset of the weight slab) in the scalar offset operand.  In
    // the steady state a staging instruction therefore costs no vector ALU work at all, and a tap that falls outside the
    // zero-padded image needs no zero page: its offset is pushed past the descriptor's size and the hardware returns zeros.
    // (At bf16 MFMA rates the 64-bit pointer arithmetic, pointer compares and the zero-page literal load of the
    // global_load_lds form cost more issue slots than the matrix instructions they feed: r02 PMC, 6.2 VALU per MFMA.)
    const int n_img_total = g.M / ohw;
    const unsigned src_bytes = (unsigned)n_img_total * g.SH * g.SW * g.SC * 2u;
    const unsigned w_bytes = (unsigned)o.N * Kp * 2u;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.src), 0, src_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(wmat), 0, w_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;                   // >= any descriptor size accepted by the entry points
    unsigned b_off[B_PASSES];
#pragma unroll
    for (int p = 0; p < B_PASSES; ++p) b_off[p] = ((unsigned)min(n0 + arow + RP * p, o.N - 1) * Kp + acol) * 2u;

    const int nk_all = Kp / BK;
    const int kt1 = min(nk_all, kt1_in);
    const int n_taps = g.KH * g.KW;

    int cur_tap = -1;
    unsigned a_off[A_PASSES];        // per-row byte offset of the current tap (OOB when the tap is outside a zero-padded image)
    const bool tap_uniform = g.SC >= BK;
    const bool refl = g.reflect != 0;
    auto row_offsets = [&](int tap, int ci) {
        int kh = (tap * g.kw_magic) >> 16;
        int kw = tap - kh * g.KW;
        if (g.tap_t) {
            const int x = kh;
            kh = kw;
            kw = x;
        }
#pragma unroll
        for (int i = 0; i < A_PASSES; ++i) {
            const int h = a_bh[i] + kh * g.kstep;
            const int w = a_bw[i] + kw * g.kstep;
            // the reflected index is in range for both rules (|offset| < size); the zero rule additionally masks
            const bool inb = refl || ((unsigned)h < (unsigned)g.SH && (unsigned)w < (unsigned)g.SW);
            const int hr = min(reflect_idx(h, g.SH), g.SH - 1), wr = min(reflect_idx(w, g.SW), g.SW - 1);
            const unsigned off = ((unsigned)((a_img[i] + hr * g.SW + wr) << g.logSC) + (unsigned)ci) * 2u;
            a_off[i] = inb ? off : OOB;
        }
    };
    // staging of one slab = address preparation (vector ALU, only when the slab crosses a tap) + NDMA buffer-load
    // instructions that can be issued in parts (interleaved with the MFMA groups of the slab being multiplied)
    constexpr int NDMA = A_PASSES + B_PASSES;
    int soff_a = 0, soff_b = 0;
    bf16 *st_la = sA, *st_lb = sB;
    auto prep_slab = [&](int kt, int buf) {
        if (tap_uniform) {
            const int kg0 = kt * BK;
            const int tap = min(kg0 >> g.logSC, n_taps - 1);
            if (tap != cur_tap) {
                row_offsets(tap, acol);
                cur_tap = tap;
            }
            soff_a = __builtin_amdgcn_readfirstlane((kg0 & (g.SC - 1)) * 2);      // provably wave-uniform: a scalar offset operand
        } else {
            const int kg = kt * BK + acol;
            row_offsets(min(kg >> g.logSC, n_taps - 1), kg & (g.SC - 1));
            soff_a = 0;
        }
        soff_b = kt * (BK * 2);
        st_la = sA + buf * A_TILE + wave * (8 * BK);
        st_lb = sB + buf * B_TILE + wave * (8 * BK);
    };
    auto issue_dma = [&](int first, int last) {          // instructions [first, last) of the prepared slab; compile-time bounds
#pragma unroll
        for (int d = 0; d < NDMA; ++d) {
            if (d < first || d >= last) continue;
            if (d < A_PASSES)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (__attribute__((address_space(3))) void*)(st_la + d * RP * BK), 16,
                                                         a_off[d < A_PASSES ? d : 0], soff_a, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b,
                                                         (__attribute__((address_space(3))) void*)(st_lb + (d - A_PASSES) * RP * BK), 16,
                                                         b_off[d >= A_PASSES ? d - A_PASSES : 0], soff_b, 0, 0);
        }
    };
    auto stage_slab = [&](int kt, int buf) {
        prep_slab(kt, buf);
        issue_dma(0, NDMA);
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
    for (int q = 0; q < 4; ++q) frag_off[q] = ((2 * q + hi) ^ fsw) * 8;
    const int a_row = (wm * TM * 32 + l31) * BK;
    const int b_row = (wn * TN * 32 + l31) * BK;
    bf16x8 fa[2][TM], fb[2][TN];
    auto load_frags = [&](int set, int buf, int q) {
        const bf16* a = sA + buf * A_TILE + a_row + frag_off[q];
        const bf16* b = sB + buf * B_TILE + b_row + frag_off[q];
#pragma unroll
        for (int i = 0; i < TM; ++i) fa[set][i] = *reinterpret_cast<const bf16x8*>(a + i * 32 * BK);
#pragma unroll
        for (int n = 0; n < TN; ++n) fb[set][n] = *reinterpret_cast<const bf16x8*>(b + n * 32 * BK);
    };
    // D[channel][pixel] = W_tile . X_tile^T: the weight fragment is the A operand, the activation fragment the B operand
    auto mfma_group = [&](int set) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int n = 0; n < TN; ++n)
                acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[set][n], fa[set][i], acc[i][n], 0, 0, 0);
    };

    if constexpr (STAGES == 2) {
        if (kt0 < kt1) {
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
    } else {
        // Three-slab LDS ring: the direct loads of slab kt+2 are issued at the top of slab kt and stay in flight ACROSS the
        // barrier that ends it -- each wave waits only for its loads of slab kt+1 (counted vmcnt: everything but the NDMA
        // youngest), then a raw s_barrier publishes them.  A slab's loads thus have two compute phases to land in.
        // WAR: slab kt+2 overwrites the buffer read during slab kt-1, whose reads every wave finished before the barrier
        // that ended kt-1.
        constexpr int D1 = (NDMA + 3) / 4, D2 = (2 * NDMA + 3) / 4, D3 = (3 * NDMA + 3) / 4;   // issue in four parts
        if (kt0 < kt1) {
            stage_slab(kt0, 0);
            if (kt0 + 1 < kt1) {
                stage_slab(kt0 + 1, 1);
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            int buf = 0;
            load_frags(0, 0, 0);
            for (int kt = kt0; kt < kt1; ++kt) {
                const bool more = kt + 1 < kt1, more2 = kt + 2 < kt1;
                const int nxt = buf == 2 ? 0 : buf + 1;
                // the staging instructions of slab kt+2 are spread over this slab's MFMA groups: while the matrix pipe works
                // through a group the wave issues the next two loads, instead of stalling the pipe behind a burst of them
                if (more2) prep_slab(kt + 2, nxt == 2 ? 0 : nxt + 1);
                load_frags(1, buf, 1);
                if (more2) issue_dma(0, D1);
                mfma_group(0);
                load_frags(0, buf, 2);
                if (more2) issue_dma(D1, D2);
                mfma_group(1);
                load_frags(1, buf, 3);
                if (more2) issue_dma(D2, D3);
                mfma_group(0);
                if (more2) issue_dma(D3, NDMA);
                if (more2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (more) load_frags(0, nxt, 0);
                mfma_group(1);
                buf = nxt;
            }
        }
    }

    // accumulator layout: lane l31 = pixel row of tile i, register r = channel (r&3) + 8*(r>>2) + 4*hi of tile n
    const float slope = dwc_act_slope(act);
    // bias vectors of this lane's columns, loaded ONCE in one batch (r04: inside the per-chunk conditionals of the store loops the
    // compiler can neither hoist nor batch a load: TM*TN*4 dependent round trips per lane)
    f32x4 bvec[TN][4];
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) bvec[n][q4] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (bias && !partial) {
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
                bvec[n][q4] = *reinterpret_cast<const f32x4*>(bias + min(n0 + (wn * TN + n) * 32 + 8 * q4 + 4 * hi, o.N - 4));
    }
    if constexpr (F32OUT) {
        float* dst = (float*)o.dst + part_offset;
        auto store = [&](auto general) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int m = m0 + (wm * TM + i) * 32 + l31;
                if (m >= g.M) continue;
                const int n_img = m / ohw;
                const int rem = m - n_img * ohw;
                const int oh = rem / g.OW, ow = rem - oh * g.OW;
                const size_t prow = ((size_t)n_img * o.OHf + (oh * o.os + oph)) * o.OWf + (ow * o.os + opw);
#pragma unroll
                for (int n = 0; n < TN; ++n)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const int col = n0 + (wn * TN + n) * 32 + 8 * q4 + 4 * hi;
                        if (col >= o.N) continue;
                        f32x4 v = {acc[i][n][4 * q4], acc[i][n][4 * q4 + 1], acc[i][n][4 * q4 + 2], acc[i][n][4 * q4 + 3]};
                        if (!partial) {
                            v += bvec[n][q4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                if constexpr (decltype(general)::value) v[k] = dwc_act_apply(v[k], act, col + k);
                                else v[k] = dwc_act_simple(v[k], slope);
                            }
                        }
                        *reinterpret_cast<f32x4*>(dst + prow * o.N + col) = v;
                    }
            }
        };
        if (dwc_act_is_simple(act)) store(std::false_type{});
        else store(std::true_type{});
    } else {
        __syncthreads();                                  // every wave is done with the operand tiles
        bf16* sC = smem;                                  // [BM][LDC]
        auto to_lds = [&](auto general) {
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int row = (wm * TM + i) * 32 + l31;
#pragma unroll
                for (int n = 0; n < TN; ++n)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) {
                        const int cl = (wn * TN + n) * 32 + 8 * q4 + 4 * hi;
                        const int col = n0 + cl;
                        f32x4 v = {acc[i][n][4 * q4], acc[i][n][4 * q4 + 1], acc[i][n][4 * q4 + 2], acc[i][n][4 * q4 + 3]};
                        if (col < o.N) {
                            v += bvec[n][q4];
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                if constexpr (decltype(general)::value) v[k] = dwc_act_apply(v[k], act, col + k);
                                else v[k] = dwc_act_simple(v[k], slope);
                            }
                        }
                        *reinterpret_cast<bf16x4*>(sC + row * LDC + cl) = pack4(v[0], v[1], v[2], v[3]);
                    }
            }
        };
        if (dwc_act_is_simple(act)) to_lds(std::false_type{});
        else to_lds(std::true_type{});
        __syncthreads();
        bf16* dst = (bf16*)o.dst;
        constexpr int CPR = BN / 8;                       // 16-byte chunks per tile row
        const bool wide = (o.N & 7) == 0;                 // rows are 16-byte aligned
        for (int idx = t; idx < BM * CPR; idx += 64 * NW) {
            const int row = idx / CPR, ch = idx - row * CPR;
            const int m = m0 + row, col = n0 + ch * 8;
            if (m >= g.M || col >= o.N) continue;
            const int n_img = m / ohw;
            const int rem = m - n_img * ohw;
            const int oh = rem / g.OW, ow = rem - oh * g.OW;
            const int py = oh * o.os + oph, px = ow * o.os + opw;
            const size_t prow = ((size_t)n_img * o.OHf + py) * o.OWf + px;
            bf16* d = dst + prow * o.N + col;
            if (o.crop) {                                 // interior pixels of a padded gradient image go straight to the unpadded tensor
                const int yy = py - o.crop, xx = px - o.crop;
                if ((unsigned)yy < (unsigned)o.IH && (unsigned)xx < (unsigned)o.IW)
                    d = (bf16*)o.inner + (((size_t)n_img * o.IH + yy) * o.IW + xx) * o.N + col;
            }
            const bf16* s = sC + row * LDC + ch * 8;
            if (wide && col + 8 <= o.N) {
                *reinterpret_cast<bf16x8*>(d) = *reinterpret_cast<const bf16x8*>(s);
            } else {
                *reinterpret_cast<bf16x4*>(d) = *reinterpret_cast<const bf16x4*>(s);
                if (col + 8 <= o.N) *reinterpret_cast<bf16x4*>(d + 4) = *reinterpret_cast<const bf16x4*>(s + 4);
            }
        }
    }
#endif
}

template <int BM, int BN, int WM, int WN, int TM, int TN, bool F32OUT, int STAGES = 2>
__global__ __launch_bounds__(64 * WM * WN) void gemm_kernel_h(Gather g, const bf16* __restrict__ wmat, size_t w_class_stride, Scatter o,
                                                     const float* __restrict__ bias, int act, int tiles_n, int kt_per_split,
                                                     size_t part_stride) {
    const int cls = blockIdx.z, split = blockIdx.y;
    gemm_body_h<BM, BN, WM, WN, TM, TN, F32OUT, STAGES>(g, wmat + (size_t)cls * w_class_stride, o, bias, act, tiles_n, split * kt_per_split,
                                                (split + 1) * kt_per_split, (size_t)split * part_stride, part_stride != 0,
                                                cls >> 1, cls & 1, blockIdx.x, gridDim.x);
}

// the border ring of a stride-1 data gradient: up to four small products in one launch, fp32 strips (see conv_igemm.hip)
// (F32OUT = false: whole-K strips rounded to bf16 at their place in a padded bf16 gradient image -- the stride-2 ring)
template <int BM, int BN, int WM, int WN, int TM, int TN, bool F32OUT = true>
__global__ __launch_bounds__(256) void gemm_strips_kernel_h(StripSet ss) {
    const Strip& s = ss.s[blockIdx.z];
    if ((int)blockIdx.x >= s.tiles) return;
    const int kt0 = s.kt0 + blockIdx.y * ss.kt_per_part;
    gemm_body_h<BM, BN, WM, WN, TM, TN, F32OUT>(s.g, (const bf16*)s.w, s.o, nullptr, DWC_ACT_NONE, s.tiles_n, kt0,
                                                min(s.kt1, kt0 + ss.kt_per_part), blockIdx.y * ss.part_stride, F32OUT, s.oph, s.opw,
                                                blockIdx.x, s.tiles);
}

// dst[i] = act(sum_s part[s][i] + bias[i % N]) rounded to bf16, fixed summation order
__global__ __launch_bounds__(256) void splitk_reduce_kernel_h(const float* __restrict__ part, bf16* __restrict__ dst,
                                                              const float* __restrict__ bias, size_t total4, size_t stride4,
                                                              int splits, int N, int act) {
    const int nq = N >> 2;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
        f32x4 s = reinterpret_cast<const f32x4*>(part)[i];
        for (int z = 1; z < splits; ++z) s += reinterpret_cast<const f32x4*>(part)[(size_t)z * stride4 + i];
        const int c4 = i % nq;
        if (bias) s += reinterpret_cast<const f32x4*>(bias)[c4];
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] = dwc_act_apply(s[k], act, c4 * 4 + k);
        reinterpret_cast<bf16x4*>(dst)[i] = pack4(s[0], s[1], s[2], s[3]);
    }
}

// ------------------------------------------------------------------------------------------
// weight gradient: dW[k][n] = sum_m A[m][k] * dY[m][n], tile 128(k) x BN(n), pixels in slabs of 64, split over pixel
// ranges into fp32 slabs.  LDS tiles are row-major [pixel][channel]; 16-byte chunk c of pixel row m is stored at chunk
// slot c ^ sw(m) so that the four pixel rows of a transposing read fall on different banks:
//   256-byte rows (128 channels): sw = 4*(m&3);  128-byte rows (64): sw = 4*((m>>1)&1);  64-byte rows (32): none.
// ------------------------------------------------------------------------------------------
template <int CPRW>
__device__ __forceinline__ int tr_swizzle(int m) {
    return CPRW == 16 ? 4 * (m & 3) : (CPRW == 8 ? 4 * ((m >> 1) & 1) : 0);
}

template <int BN, int WM, int WN, int TM, int TN, int DBG = 0>     // DBG: timing ablations (1 no MFMA, 2 no operand reads, 4 no staging)
__global__ __launch_bounds__(256) void wgrad_kernel_h(Gather g, const bf16* __restrict__ dy, int N, float* __restrict__ slab,
                                                      int m_chunk) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert(WM * WN == 4 && WM * TM * 32 == 128 && WN * TN * 32 == BN, "tile shape");
    constexpr int MS = 64;                               // pixels per slab
    constexpr int A_TILE = MS * 128, B_TILE = MS * BN;   // elements
    __shared__ __attribute__((aligned(16))) bf16 smem[2 * (A_TILE + B_TILE)];
    bf16* sA = smem;               // [buf][m][128 k]
    bf16* sB = smem + 2 * A_TILE;  // [buf][m][BN]
    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int k0 = blockIdx.x * 128, n0 = blockIdx.y * BN;
    const int split = blockIdx.z;
    const int m_begin = split * m_chunk;
    const int m_end = min(g.M, m_begin + m_chunk);

    // A staging: thread = (row ar [+16 per pass], chunk slot t&15); it fetches LOGICAL chunk slot ^ sw(row), whose tap and
    // channel are fixed for the thread (16*pass keeps row & 3).  k >= K (tile tail) is clamped to a valid tap: never stored.
    const int ar = t >> 4;
    const int a_lc = (t & 15) ^ tr_swizzle<16>(ar);
    const int kg = k0 + a_lc * 8;
    const int tap = min(kg >> g.logSC, g.KH * g.KW - 1);
    const int ci = kg & (g.SC - 1);
    const int kh = (tap * g.kw_magic) >> 16;
    const int kw = tap - kh * g.KW;
    const int dh = kh * g.kstep + g.off_h, dw = kw * g.kstep + g.off_w;
    constexpr int CB = BN / 8;                           // chunks per dY tile row
    constexpr int B_RPP = 256 / CB;                      // rows per pass
    constexpr int B_PASSES = MS / B_RPP;
    const int br = t / CB;
    const int b_lc = (t % CB) ^ tr_swizzle<CB>(br);
    const int bcol_c = min(n0 + b_lc * 8, N - 8 >= 0 ? N - 8 : 0);      // columns past N are never stored: clamp instead of masking
    const int ohw = g.OH * g.OW;

    // buffer loads into LDS (see gemm_body_h): the dY descriptor ends at this workgroup's last pixel row, so rows past the
    // end of the pixel range read as zeros without a mask, and its slab offset travels in the scalar operand; the gathered
    // x rows carry one 32-bit offset each.
    const unsigned x_bytes = (unsigned)(g.M / ohw) * g.SH * g.SW * g.SC * 2u;
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(g.src), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_dy =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(dy), 0, (unsigned)m_end * (unsigned)N * 2u, 0x00020000);
    unsigned b_voff[B_PASSES];
#pragma unroll
    for (int p = 0; p < B_PASSES; ++p) b_voff[p] = ((unsigned)(br + p * B_RPP) * N + bcol_c) * 2u;
    auto stage_slab = [&](int mb, int buf) {
        bf16* la = sA + buf * A_TILE + wave * 512;
        bf16* lb = sB + buf * B_TILE + wave * 512;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = min(mb + ar + 16 * i, g.M - 1);
            int n, oh, ow;
            if (g.logOW >= 0) {
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
            const unsigned off = ((unsigned)(((n * g.SH + h) * g.SW + w) << g.logSC) + (unsigned)ci) * 2u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(la + i * 16 * 128), 16, off, 0,
                                                     0, 0);
        }
        const int soff = __builtin_amdgcn_readfirstlane(mb * N * 2);
#pragma unroll
        for (int p = 0; p < B_PASSES; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_dy, (__attribute__((address_space(3))) void*)(lb + p * B_RPP * BN), 16,
                                                     b_voff[p], soff, 0, 0);
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // transposing reads: 16-lane group gq = lane>>4 covers columns 16*(gq&1).. of a 32-wide tile and pixels 8*(gq>>1)..;
    // lane 4q+p of the group addresses pixel row q, columns 4p..4p+3; it receives column (lane&15), pixels 0..3 of the block.
    const int li = lane & 15, gam = (lane >> 4) & 1, hi = lane >> 5;
    const int tq = li >> 2, tp = li & 3;
    int a_off[TM], b_off[TN];          // element offsets inside one slab tile for m-step 0, first half (pixel row 8*hi + tq)
    {
        const int m_loc = 8 * hi + tq;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int kl = (wm * TM + i) * 32 + 16 * gam + 4 * tp;
            a_off[i] = m_loc * 128 + (((kl >> 3) ^ tr_swizzle<16>(m_loc)) << 3) + (kl & 7);
        }
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            const int nl = (wn * TN + n) * 32 + 16 * gam + 4 * tp;
            b_off[n] = m_loc * BN + (((nl >> 3) ^ tr_swizzle<CB>(m_loc)) << 3) + (nl & 7);
        }
    }
    // rows advance by 4 (second half) and 16 (next m-step): both keep (m & 3); for 128-byte rows the swizzle uses bit 1 of
    // m, unchanged by +4 and +16 as well
    // Fragment reads are asm statements with hand-counted waits (tied to the fragment registers, so the MFMAs stay behind
    // them): as builtins hipcc cannot tell them from the LDS-DMA's target and drains vmcnt(0) -- the slab staged a moment
    // ago -- in front of the next read (see wgrad_halo_kernel).
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
    bf16x4 fa[2][TM][2], fb[2][TN][2];
    constexpr int NREAD = 2 * (TM + TN);                 // ds_read instructions of one load_ops
    auto tr_read = [](bf16x4& dst, unsigned addr, auto offc) {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(decltype(offc)::value));
    };
    auto load_ops = [&](auto setc, int buf, int ms) {
        constexpr int set = decltype(setc)::value;
        if ((DBG & 2) && ms != 0) return;
        const unsigned a = lds0 + (unsigned)(buf * A_TILE + ms * 16 * 128) * 2u;
        const unsigned b = lds0 + (unsigned)(2 * A_TILE + buf * B_TILE + ms * 16 * BN) * 2u;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            tr_read(fa[set][i][0], a + (unsigned)a_off[i] * 2u, std::integral_constant<int, 0>{});
            tr_read(fa[set][i][1], a + (unsigned)a_off[i] * 2u, std::integral_constant<int, 4 * 128 * 2>{});
        }
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            tr_read(fb[set][n][0], b + (unsigned)b_off[n] * 2u, std::integral_constant<int, 0>{});
            tr_read(fb[set][n][1], b + (unsigned)b_off[n] * 2u, std::integral_constant<int, 4 * BN * 2>{});
        }
    };
    // fragments of `set` have landed when at most `younger` later reads are in flight
    auto wait_ops = [&](auto setc, auto youngerc) {
        constexpr int set = decltype(setc)::value;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            bf16x4 &r0 = fa[set][i][0], &r1 = fa[set][i][1];    // (asm operands inside a generic lambda: bind first)
            asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(r0), "+v"(r1) : "n"(decltype(youngerc)::value));
        }
#pragma unroll
        for (int n = 0; n < TN; ++n) {
            bf16x4 &r0 = fb[set][n][0], &r1 = fb[set][n][1];
            asm volatile("" : "+v"(r0), "+v"(r1));
        }
    };
    auto mfma_ops = [&](auto setc) {
        constexpr int set = decltype(setc)::value;
        if (DBG & 1) {
#pragma unroll
            for (int i = 0; i < TM; ++i) { const bf16x4 r0 = fa[set][i][0], r1 = fa[set][i][1]; asm volatile("" ::"v"(r0), "v"(r1)); }
#pragma unroll
            for (int n = 0; n < TN; ++n) { const bf16x4 r0 = fb[set][n][0], r1 = fb[set][n][1]; asm volatile("" ::"v"(r0), "v"(r1)); }
            return;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int n = 0; n < TN; ++n)
                acc[i][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    __builtin_shufflevector(fa[set][i][0], fa[set][i][1], 0, 1, 2, 3, 4, 5, 6, 7),
                    __builtin_shufflevector(fb[set][n][0], fb[set][n][1], 0, 1, 2, 3, 4, 5, 6, 7), acc[i][n], 0, 0, 0);
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    typedef std::integral_constant<int, NREAD> YR;
    if (m_begin < m_end) {
        stage_slab(m_begin, 0);
        lds_dma_barrier();
        int buf = 0;
        load_ops(S0{}, 0, 0);
        for (int mb = m_begin; mb < m_end; mb += MS) {
            const bool more = mb + MS < m_end;
            if (more && !(DBG & 4)) stage_slab(mb + MS, buf ^ 1);
            load_ops(S1{}, buf, 1);
            wait_ops(S0{}, YR{});
            mfma_ops(S0{});
            load_ops(S0{}, buf, 2);
            wait_ops(S1{}, YR{});
            mfma_ops(S1{});
            load_ops(S1{}, buf, 3);
            wait_ops(S0{}, YR{});
            mfma_ops(S0{});
            wait_ops(S1{}, S0{});                        // every read of this buffer is in registers before the barrier frees it
            lds_dma_barrier();
            if (more) load_ops(S0{}, buf ^ 1, 0);
            mfma_ops(S1{});
            buf ^= 1;
        }
    }
    float* out = slab + (size_t)blockIdx.z * g.K * N;
    const int l31 = lane & 31;
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
#endif
}

// slab[s][(kh,kw,ci)][co] summed over s -> dw[co][ci][kh][kw] (state_dict layout, fp32), real channels only
__global__ void wgrad_reduce_kernel_h(const float* __restrict__ slab, float* __restrict__ dw, int splits, int K, int N, int Cin,
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

// reflect-pad adjoint on bf16 images (fp32 accumulation)
__global__ void fold_reflect_kernel_h(const bf16* __restrict__ gp, bf16* __restrict__ dx, int B, int H, int W, int C4, int pad,
                                      int Wp) {
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
    const bf16x4* g4 = reinterpret_cast<const bf16x4*>(gp);
    for (int a = 0; a < nh; ++a)
        for (int b = 0; b < nw; ++b) {
            const bf16x4 v = g4[((size_t)(n * Hp + hs[a]) * Wp + ws[b]) * C4 + c];
            s += f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
        }
    reinterpret_cast<bf16x4*>(dx)[idx] = pack4(s[0], s[1], s[2], s[3]);
}

// the same with 16-byte accesses (8 channels per lane) and a (W*C8 / 256, H, B) grid: no per-element 64-bit divisions.  r02: the
// 8-byte form ran at 3.5 TB/s on the stride-2 data gradients of configs[2] (3.4 ms per step).
__global__ __launch_bounds__(256) void fold_reflect_kernel_h8(const bf16* __restrict__ gp, bf16* __restrict__ dx, int H, int W, int C8,
                                                              int pad, int Wp) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= W * C8) return;
    const int w = idx / C8, c = idx - w * C8;
    const int h = blockIdx.y;
    const size_t n = blockIdx.z;
    const int Hp = H + 2 * pad;
    int hs[3], ws[3], nh = 0, nw = 0;
    hs[nh++] = h + pad;
    if (h >= 1 && h <= pad) hs[nh++] = pad - h;
    if (h >= H - 1 - pad && h <= H - 2) hs[nh++] = pad + 2 * (H - 1) - h;
    ws[nw++] = w + pad;
    if (w >= 1 && w <= pad) ws[nw++] = pad - w;
    if (w >= W - 1 - pad && w <= W - 2) ws[nw++] = pad + 2 * (W - 1) - w;
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = 0.f;
    const bf16x8* g8 = reinterpret_cast<const bf16x8*>(gp);
    for (int a = 0; a < nh; ++a)
        for (int b = 0; b < nw; ++b) {
            const bf16x8 v = g8[((n * Hp + hs[a]) * (size_t)Wp + ws[b]) * C8 + c];
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] += (float)v[k];
        }
    bf16x8 o;
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (bf16)s[k];
    reinterpret_cast<bf16x8*>(dx)[((n * H + h) * (size_t)W + w) * C8 + c] = o;
}

// dx (holds the interior of the padded gradient image already: Scatter::crop) += the border ring of gp folded back by the reflect
// rule.  Only the pixels a ring pixel folds onto are touched: rows 1..pad and H-1-pad..H-2 (whole rows), and columns 1..pad and
// W-1-pad..W-2 of the other rows.
__global__ __launch_bounds__(256) void fold_band_kernel_h8(const bf16* __restrict__ gp, bf16* __restrict__ dx, int B, int H, int W, int C8,
                                                                    int pad, int Wp) {
    // one thread per (image, band pixel, channel chunk): per image the 2*pad band rows whole (W pixels each), then the 2*pad band
    // columns of the H - 2*pad other rows
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    const int rows_done = 2 * pad, rest = H - 2 * pad;          // band rows, other rows
    const int band = rows_done * W + rest * 2 * pad;            // band pixels per image
    const size_t total = (size_t)B * band * C8;
    if (idx >= total) return;
    const int c = idx % C8;
    size_t r = idx / C8;
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
    const size_t o = ((n * H + h) * (size_t)W + w) * C8 + c;
    const bf16x8 cur = reinterpret_cast<const bf16x8*>(dx)[o];
    float s[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) s[k] = (float)cur[k];
    const bf16x8* g8 = reinterpret_cast<const bf16x8*>(gp);
    for (int a = 0; a < nh; ++a)
        for (int b = 0; b < nw; ++b) {
            if (a == 0 && b == 0) continue;               // the pixel's own (interior) value is in dx already
            const bf16x8 v = g8[((n * Hp + hs[a]) * (size_t)Wp + ws[b]) * C8 + c];
#pragma unroll
            for (int k = 0; k < 8; ++k) s[k] += (float)v[k];
        }
    bf16x8 res;
#pragma unroll
    for (int k = 0; k < 8; ++k) res[k] = (bf16)s[k];
    reinterpret_cast<bf16x8*>(dx)[o] = res;
}

// dx (bf16, holds the interior) += the fp32 border ring folded back by the reflect rule (see fold_ring_kernel, conv_igemm.hip)
__global__ void fold_ring_kernel_h(bf16* __restrict__ dx, const float* __restrict__ ring, size_t off_bottom, size_t off_left,
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
        if ((h >= 1 && h <= pad) || (h >= H - 1 - pad && h <= H - 2)) return;
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
    bf16x4* out = reinterpret_cast<bf16x4*>(dx) + ((size_t)(n * H + h) * W + w) * C4 + c;
    const bf16x4 cur = *out;
    f32x4 s = {(float)cur[0], (float)cur[1], (float)cur[2], (float)cur[3]};
    for (int a = 0; a < nh; ++a)
        for (int b = 0; b < nw; ++b) {
            if (a == 0 && b == 0) continue;
            const int rh = hs[a], rw = ws[b];
            size_t e;
            if (rh < pad) e = ((size_t)(n * pad + rh) * Wp + rw) * C4;
            else if (rh >= pad + H) e = off_bottom / 4 + ((size_t)(n * pad + rh - pad - H) * Wp + rw) * C4;
            else if (rw < pad) e = off_left / 4 + ((size_t)(n * H + rh - pad) * pad + rw) * C4;
            else e = off_right / 4 + ((size_t)(n * H + rh - pad) * pad + rw - pad - W) * C4;
            for (int p = 0; p < parts; ++p) s += reinterpret_cast<const f32x4*>(ring + p * part_stride)[e + c];
        }
    *out = pack4(s[0], s[1], s[2], s[3]);
}

// fp32 OIHW master weights -> bf16 [N][Kp] streaming layouts (see conv_igemm.hip: same orderings, Kp a multiple of 64)
__global__ void weight_prepare_fwd_kernel_h(const float* __restrict__ w, bf16* __restrict__ out, int Cout, int Cin, int KHW,
                                            int cout_pad, int cin_pad, int Kp) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)cout_pad * Kp) return;
    const int k = idx % Kp, co = idx / Kp;
    const int ci = k % cin_pad, tap = k / cin_pad;
    float v = 0.f;
    if (co < Cout && ci < Cin && tap < KHW) v = w[((size_t)co * Cin + ci) * KHW + tap];
    out[idx] = (bf16)v;
}

__global__ void weight_prepare_dgrad_kernel_h(const float* __restrict__ w, bf16* __restrict__ out, int Cout, int Cin, int KH, int KW,
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
    out[idx] = (bf16)(ok ? w[((size_t)co * Cin + ci) * KH * KW + kh * KW + kw] : 0.f);
}

// ---- launchers -------------------------------------------------------------------------------
template <int BM, int BN, int WM, int WN, int TM, int TN, int STAGES = 2>
void launch_variant_h(const Gather& g, const bf16* w, size_t wcs, int classes, const Scatter& o, const float* bias, int act,
                      const Plan& p, size_t part_stride, bool f32out, hipStream_t st) {
    const int tiles_m = (g.M + BM - 1) / BM, tiles_n = (o.N + BN - 1) / BN;
    if (f32out)
        hipLaunchKernelGGL((gemm_kernel_h<BM, BN, WM, WN, TM, TN, true, STAGES>), dim3(tiles_m * tiles_n, p.splits, classes),
                           dim3(64 * WM * WN), 0, st, g, w, wcs, o, bias, act, tiles_n, p.kt_per_split, part_stride);
    else
        hipLaunchKernelGGL((gemm_kernel_h<BM, BN, WM, WN, TM, TN, false, STAGES>), dim3(tiles_m * tiles_n, p.splits, classes),
                           dim3(64 * WM * WN), 0, st, g, w, wcs, o, bias, act, tiles_n, p.kt_per_split, part_stride);
}

// Tile choice for the bf16 kernels: the fp32 planner's candidates plus the 8-wave 256-row tiles (one workgroup per CU,
// 128 KiB / 96 KiB of LDS).  At bf16 MFMA rates the 128x128 tile is bound by what a CU can pull out of L2 into LDS
// (32 KB per 64-deep slab) and by LDS read bandwidth (one ds_read_b128 per MFMA with 64x64 wave tiles); a 256x256 tile
// halves both per flop (128x64 wave tiles: 0.75 reads per MFMA), a 256x128 tile saves a quarter.  `f` = measured relative
// rate at full residency (kernel_bench_bf16, r02), the cost model is plan_gemm's (rounds of resident workgroups).
Plan plan_gemm_h(int M, int N, int K, int classes) {
    struct Cand { int bm, bn, resident; float f; };
    static const Cand all[] = {{256, 256, 1, 1.45f}, {256, 128, 1, 1.25f}, {128, 128, 2, 1.0f}, {128, 64, 3, 0.8f},
                               {64, 64, 5, 0.55f}, {128, 32, 4, 0.35f}};
    const int nk = (K + BK - 1) / BK;
    Plan best = {128, 32, 1, nk};
    float best_cost = 3.0e38f;
    for (const Cand& c : all) {
        const bool ok = N <= 32 ? c.bn == 32 : (N <= 64 ? c.bn == 64 : (c.bn != 32 && (c.bn <= 128 || N > 128)));
        if (!ok) continue;
        const long blocks = (long)((M + c.bm - 1) / c.bm) * ((N + c.bn - 1) / c.bn) * classes;
        const long n = (blocks + NUM_CU - 1) / NUM_CU;
        const long full = n / c.resident, rem = n % c.resident;
        const float tile = (float)c.bm * c.bn / c.f;
        float cost = (float)full * c.resident * tile;
        if (rem) cost += (float)rem * tile / (rem == 1 && c.resident > 1 ? 0.62f : 0.9f);
        if (cost < best_cost) {
            best_cost = cost;
            best = {c.bm, c.bn, 1, nk};
        }
    }
    const long blocks = (long)((M + best.bm - 1) / best.bm) * ((N + best.bn - 1) / best.bn) * classes;
    if (blocks < NUM_CU / 2 && nk >= 8) {
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

size_t gemm_ws_bytes_h(int M, int N, int K, int classes, size_t dst_elems) {
    const Plan p = plan_gemm_h(M, N, K, classes);
    return p.splits > 1 ? (size_t)p.splits * dst_elems * sizeof(float) : 0;
}

int launch_gemm_h(const Gather& g, const bf16* w, size_t w_class_stride, int classes, Scatter o, const float* bias, int act,
                  size_t dst_elems, void* ws, size_t ws_bytes, hipStream_t st) {
    Plan p = plan_gemm_h(g.M, o.N, g.K, classes);
    bf16* final_dst = (bf16*)o.dst;
    size_t part_stride = 0;
    if (p.splits > 1) {
        if (!ws || ws_bytes < (size_t)p.splits * dst_elems * sizeof(float)) {
            p.splits = 1;
            p.kt_per_split = (g.K + BK - 1) / BK;
        } else {
            o.dst = ws;
            part_stride = dst_elems;
        }
    }
    const bool f32out = p.splits > 1;
    if (p.bm == 256 && p.bn == 256) launch_variant_h<256, 256, 2, 4, 4, 2>(g, w, w_class_stride, classes, o, bias, act, p, part_stride, f32out, st);
    else if (p.bm == 256 && p.bn == 128) launch_variant_h<256, 128, 4, 2, 2, 2, 3>(g, w, w_class_stride, classes, o, bias, act, p, part_stride, f32out, st);
    else if (p.bm == 128 && p.bn == 128) launch_variant_h<128, 128, 2, 2, 2, 2>(g, w, w_class_stride, classes, o, bias, act, p, part_stride, f32out, st);
    else if (p.bm == 128 && p.bn == 64) launch_variant_h<128, 64, 2, 2, 2, 1>(g, w, w_class_stride, classes, o, bias, act, p, part_stride, f32out, st);
    else if (p.bm == 64 && p.bn == 64) launch_variant_h<64, 64, 2, 2, 1, 1>(g, w, w_class_stride, classes, o, bias, act, p, part_stride, f32out, st);
    else launch_variant_h<128, 32, 4, 1, 1, 1>(g, w, w_class_stride, classes, o, bias, act, p, part_stride, f32out, st);
    DWC_LAUNCH_CHECK();
    if (p.splits > 1) {
        const size_t total4 = dst_elems / 4;
        size_t blocks = (total4 + 255) / 256;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(splitk_reduce_kernel_h, dim3((unsigned)blocks), dim3(256), 0, st, (const float*)ws, final_dst, bias, total4,
                           total4, p.splits, o.N, act);
        DWC_LAUNCH_CHECK();
    }
    return DWC_OK;
}

int wgrad_launch_h(const FwdGeom& f, const bf16* dy, float* dw_oihw, int Cin, int Cout, int KHW, int cin_real, int cout_real,
                   void* ws, size_t ws_bytes, hipStream_t st) {
    const Gather& g = f.g;
    if (Cout < 8) return DWC_EINVAL;
    int splits, chunk;
    wgrad_plan(g.M, g.K, Cout, &splits, &chunk, 1, 64);
    if (!ws || ws_bytes < (size_t)splits * g.K * Cout * sizeof(float)) return DWC_EWORKSPACE;
    float* slab = (float*)ws;
    const int tk = (g.K + 127) / 128;
#ifdef DWC_DEV_ABLATIONS      // timing-only ablations (WRONG results): compiled only by `make ABLATIONS=1`, never in the shipped .so
    static const int dbg = getenv("DWC_WGRAD_DBG") ? atoi(getenv("DWC_WGRAD_DBG")) : 0;
    if (Cout > 64 && dbg) {
#define WG_DBG(D) case D: hipLaunchKernelGGL((wgrad_kernel_h<128, 2, 2, 2, 2, D>), dim3(tk, (Cout + 127) / 128, splits), dim3(256), 0, st, g, dy, Cout, slab, chunk); break;
        switch (dbg) { WG_DBG(1) WG_DBG(2) WG_DBG(3) WG_DBG(4) WG_DBG(7) default: return DWC_EINVAL; }
#undef WG_DBG
    } else
#endif
    if (Cout > 64) {
        hipLaunchKernelGGL((wgrad_kernel_h<128, 2, 2, 2, 2>), dim3(tk, (Cout + 127) / 128, splits), dim3(256), 0, st, g, dy, Cout, slab,
                           chunk);
    } else if (Cout > 32) {
        hipLaunchKernelGGL((wgrad_kernel_h<64, 2, 2, 2, 1>), dim3(tk, 1, splits), dim3(256), 0, st, g, dy, Cout, slab, chunk);
    } else {
        hipLaunchKernelGGL((wgrad_kernel_h<32, 4, 1, 1, 1>), dim3(tk, 1, splits), dim3(256), 0, st, g, dy, Cout, slab, chunk);
    }
    DWC_LAUNCH_CHECK();
    const size_t total = (size_t)g.K * Cout;
    if (!wgrad_reduce_wide(slab, dw_oihw, splits, g.K, Cout, Cin, KHW, cin_real, cout_real, st))
        hipLaunchKernelGGL(wgrad_reduce_kernel_h, dim3((total + 255) / 256), dim3(256), 0, st, slab, dw_oihw, splits, g.K, Cout, Cin, KHW,
                           cin_real, cout_real);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

}  // namespace

extern "C" {

size_t dwc_bf16_weight_prepared_elems(int Cout, int Cin, int KH, int KW, int stride, int cout_pad, int cin_pad, int for_dgrad) {
    if (!for_dgrad) return (size_t)cout_pad * ((KH * KW * cin_pad + BK - 1) / BK * BK);
    if (stride == 1) return (size_t)cin_pad * ((KH * KW * cout_pad + BK - 1) / BK * BK);
    return (size_t)4 * cin_pad * ((4 * cout_pad + BK - 1) / BK * BK);
}

int dwc_bf16_weight_prepare_fwd(const float* w, void* out, int Cout, int Cin, int KH, int KW, int cout_pad, int cin_pad,
                                void* stream) {
    if (cout_pad < Cout || cin_pad < Cin) return DWC_EINVAL;
    const int Kp = (KH * KW * cin_pad + BK - 1) / BK * BK;
    const size_t total = (size_t)cout_pad * Kp;
    hipLaunchKernelGGL(weight_prepare_fwd_kernel_h, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, (bf16*)out, Cout,
                       Cin, KH * KW, cout_pad, cin_pad, Kp);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

int dwc_bf16_weight_prepare_dgrad(const float* w, void* out, int Cout, int Cin, int KH, int KW, int stride, int cout_pad,
                                  int cin_pad, void* stream) {
    if (cout_pad < Cout || cin_pad < Cin) return DWC_EINVAL;
    if (stride == 2 && !(KH == 4 && KW == 4)) return DWC_EINVAL;
    if (stride != 1 && stride != 2) return DWC_EINVAL;
    const int Kp = ((stride == 1 ? KH * KW : 4) * cout_pad + BK - 1) / BK * BK;
    const size_t total = (size_t)(stride == 1 ? 1 : 4) * cin_pad * Kp;
    hipLaunchKernelGGL(weight_prepare_dgrad_kernel_h, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w, (bf16*)out,
                       Cout, Cin, KH, KW, stride, cout_pad, cin_pad, Kp);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

size_t dwc_bf16_conv2d_fwd_ws_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    FwdGeom f;
    if (!fwd_geom(nullptr, nullptr, B, H, W, Cin, Cout, KH, KW, stride, pad, &f, MIN_LOG_C)) return 0;
    return gemm_ws_bytes_h(f.g.M, Cout, f.g.K, 1, f.dst_elems);
}

int dwc_bf16_conv2d_fwd(const void* x, const void* w_prepared, const float* bias, void* y, int B, int H, int W, int Cin, int Cout,
                        int KH, int KW, int stride, int pad, int act, void* ws, size_t ws_bytes, void* stream) {
    FwdGeom f;
    if ((Cout & 7) || !fwd_geom(x, y, B, H, W, Cin, Cout, KH, KW, stride, pad, &f, MIN_LOG_C)) return DWC_EINVAL;
    return launch_gemm_h(f.g, (const bf16*)w_prepared, 0, 1, f.o, bias, act, f.dst_elems, ws, ws_bytes, (hipStream_t)stream);
}

/* The frozen VGG16 trunk of the perceptual loss under bf16 (reference networks.py:639-688): zero-padded stride-1 convolution, forward
 * (the forward kernel with the zero rule: an out-of-image tap pushes its offset past the buffer descriptor and reads zeros) and
 * data gradient (the zero-padded correlation with the flipped filter; the adjoint of zero padding is a crop, no ring). */
int dwc_bf16_conv2d_fwd_zeropad(const void* x, const void* w_prepared, const float* bias, void* y, int B, int H, int W, int Cin,
                                int Cout, int KH, int KW, int stride, int pad, int act, void* ws, size_t ws_bytes, void* stream) {
    FwdGeom f;
    if ((Cout & 7) || !fwd_geom(x, y, B, H, W, Cin, Cout, KH, KW, stride, pad, &f, MIN_LOG_C)) return DWC_EINVAL;
    f.g.reflect = 0;
    return launch_gemm_h(f.g, (const bf16*)w_prepared, 0, 1, f.o, bias, act, f.dst_elems, ws, ws_bytes, (hipStream_t)stream);
}

size_t dwc_bf16_conv2d_bwd_data_zeropad_ws_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW, int pad) {
    FwdGeom f;
    if ((Cin & 7) || (Cout & 7) || !zeropad_dgrad_geom(nullptr, nullptr, B, H, W, Cin, Cout, KH, KW, pad, &f)) return 0;
    return gemm_ws_bytes_h(f.g.M, Cin, f.g.K, 1, f.dst_elems);
}

int dwc_bf16_conv2d_bwd_data_zeropad(const void* dy, const void* w_dgrad, void* dx, int B, int H, int W, int Cin, int Cout, int KH,
                                     int KW, int pad, void* ws, size_t ws_bytes, void* stream) {
    FwdGeom f;
    if ((Cin & 7) || (Cout & 7) || !zeropad_dgrad_geom(dy, dx, B, H, W, Cin, Cout, KH, KW, pad, &f)) return DWC_EINVAL;
    return launch_gemm_h(f.g, (const bf16*)w_dgrad, 0, 1, f.o, nullptr, DWC_ACT_NONE, f.dst_elems, ws, ws_bytes, (hipStream_t)stream);
}

int dwc_bf16_conv2d_fwd_ex(const void* x, const void* w_prepared, const float* bias, void* y, int B, int H, int W, int Cin,
                           int Cout, int KH, int KW, int stride_h, int stride_w, int pad_h, int pad_w, int act, void* stream) {
    FwdGeom f;
    if ((Cout & 7) || !fwd_geom_ex(x, y, B, H, W, Cin, Cout, KH, KW, stride_h, stride_w, pad_h, pad_w, &f, MIN_LOG_C)) return DWC_EINVAL;
    return launch_gemm_h(f.g, (const bf16*)w_prepared, 0, 1, f.o, bias, act, f.dst_elems, nullptr, 0, (hipStream_t)stream);
}

size_t dwc_bf16_conv2d_bwd_data_ws_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    BwdGeom f;
    if (!bwd_geom(nullptr, nullptr, B, H, W, Cin, Cout, KH, KW, stride, pad, &f, BK, MIN_LOG_C)) return 0;
    return gemm_ws_bytes_h(f.g.M, Cin, f.g.K, f.classes, f.dst_elems);
}

int dwc_bf16_conv2d_bwd_data(const void* dy, const void* w_dgrad, void* dxp, int B, int H, int W, int Cin, int Cout, int KH,
                             int KW, int stride, int pad, void* ws, size_t ws_bytes, void* stream) {
    BwdGeom f;
    if ((Cin & 7) || !bwd_geom(dy, dxp, B, H, W, Cin, Cout, KH, KW, stride, pad, &f, BK, MIN_LOG_C)) return DWC_EINVAL;
    return launch_gemm_h(f.g, (const bf16*)w_dgrad, f.wcs, f.classes, f.o, nullptr, DWC_ACT_NONE, f.dst_elems, ws, ws_bytes,
                         (hipStream_t)stream);
}

int dwc_bf16_reflect_pad_adjoint(const void* dxp, void* dx, int B, int H, int W, int C, int pad, void* stream);
/* dwc_bf16_conv2d_bwd_data + dwc_bf16_reflect_pad_adjoint in one call: dx ([B,H,W,Cin]) = reflect-pad adjoint of the gradient of
 * the padded image.  Where the GEMM runs unsplit the interior of that image is written straight into dx and only its border ring
 * into dxp (scratch for [B,H+2pad,W+2pad,Cin] bf16), a band kernel then folds the ring onto dx: one pass over the tensor instead of
 * three.  Otherwise (split-K) the two-step form runs. */
int dwc_bf16_conv2d_bwd_data_fold(const void* dy, const void* w_dgrad, void* dxp, void* dx, int B, int H, int W, int Cin, int Cout,
                                  int KH, int KW, int stride, int pad, void* ws, size_t ws_bytes, void* stream) {
    BwdGeom f;
    if ((Cin & 7) || pad <= 0 || !bwd_geom(dy, dxp, B, H, W, Cin, Cout, KH, KW, stride, pad, &f, BK, MIN_LOG_C)) return DWC_EINVAL;
    if (H < 2 * pad + 2 || W < 2 * pad + 2 || H > 65535 - 2 * pad || B > 65535) return DWC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    const Plan p = plan_gemm_h(f.g.M, f.o.N, f.g.K, f.classes);
    const bool direct = p.splits == 1;
    if (direct) {
        f.o.crop = pad; f.o.IH = H; f.o.IW = W; f.o.inner = dx;
    }
    int rc = launch_gemm_h(f.g, (const bf16*)w_dgrad, f.wcs, f.classes, f.o, nullptr, DWC_ACT_NONE, f.dst_elems, ws, ws_bytes, st);
    if (rc != DWC_OK) return rc;
    if (!direct) return dwc_bf16_reflect_pad_adjoint(dxp, dx, B, H, W, Cin, pad, stream);
    const int C8 = Cin / 8;
    const size_t band_items = (size_t)B * (2 * pad * W + (H - 2 * pad) * 2 * pad) * C8;
    hipLaunchKernelGGL(fold_band_kernel_h8, dim3((unsigned)((band_items + 255) / 256)), dim3(256), 0, st, (const bf16*)dxp, (bf16*)dx, B, H, W,
                       C8, pad, W + 2 * pad);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

/* bf16 twin of dwc_conv2d_bwd_data_s2_ring: ring of the padded gradient image of a 4x4 stride-2 reflect-pad-1 convolution as eight
 * strips into the bf16 scratch image dxp + band fold onto dx (interior by dwc_bf16_conv2d_s2_halo_bwd_data). */
int dwc_bf16_conv2d_bwd_data_s2_ring(const void* dy, const void* w_dgrad, void* dxp, void* dx, int B, int H, int W, int Cin, int Cout,
                                     void* stream) {
    S2Ring f;
    const int bm = strip_bm((long)B * max(W / 2 + 1, H / 2), (Cin + 63) / 64, 8, 1, true);
    if (!dy || !w_dgrad || !dxp || !dx || (Cin & 7) || H > 65535 - 2 || B > 65535 ||
        !s2_ring_geom(dy, w_dgrad, dxp, 2, B, H, W, Cin, Cout, &f, BK, MIN_LOG_C, bm))
        return DWC_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (bm == 128) hipLaunchKernelGGL((gemm_strips_kernel_h<128, 64, 2, 2, 2, 1, false>), dim3(f.max_tiles, 1, 8), dim3(256), 0, st, f.ss);
    else hipLaunchKernelGGL((gemm_strips_kernel_h<64, 64, 2, 2, 1, 1, false>), dim3(f.max_tiles, 1, 8), dim3(256), 0, st, f.ss);
    DWC_LAUNCH_CHECK();
    const int C8 = Cin / 8;
    const size_t band_items = (size_t)B * (2 * W + (H - 2) * 2) * C8;
    hipLaunchKernelGGL(fold_band_kernel_h8, dim3((unsigned)((band_items + 255) / 256)), dim3(256), 0, st, (const bf16*)dxp, (bf16*)dx, B, H, W,
                       C8, 1, W + 2);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

/* dx (already holding the interior of the padded gradient image dxp) += the border ring of dxp folded back by the reflect rule:
 * the second half of dwc_bf16_conv2d_bwd_data_fold for producers with their own epilogue (dwc_bf16_conv2d_stem_crop). */
int dwc_bf16_reflect_pad_adjoint_band(const void* dxp, void* dx, int B, int H, int W, int C, int pad, void* stream) {
    if (B <= 0 || (C & 7) || pad <= 0 || H < 2 * pad + 2 || W < 2 * pad + 2 || H > 65535 || B > 65535) return DWC_EINVAL;
    const int C8 = C / 8;
    hipStream_t st = (hipStream_t)stream;
    const size_t band_items = (size_t)B * (2 * pad * W + (H - 2 * pad) * 2 * pad) * C8;
    hipLaunchKernelGGL(fold_band_kernel_h8, dim3((unsigned)((band_items + 255) / 256)), dim3(256), 0, st, (const bf16*)dxp, (bf16*)dx, B, H, W,
                       C8, pad, W + 2 * pad);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

int dwc_bf16_reflect_pad_adjoint(const void* dxp, void* dx, int B, int H, int W, int C, int pad, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3) || pad < 0 || pad >= H || pad >= W) return DWC_EINVAL;
    if (!(C & 7) && H <= 65535 && B <= 65535) {
        hipLaunchKernelGGL(fold_reflect_kernel_h8, dim3((W * (C / 8) + 255) / 256, H, B), dim3(256), 0, (hipStream_t)stream,
                           (const bf16*)dxp, (bf16*)dx, H, W, C / 8, pad, W + 2 * pad);
        DWC_LAUNCH_CHECK();
        return DWC_OK;
    }
    const size_t total = (size_t)B * H * W * (C / 4);
    hipLaunchKernelGGL(fold_reflect_kernel_h, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const bf16*)dxp,
                       (bf16*)dx, B, H, W, C / 4, pad, W + 2 * pad);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

size_t dwc_bf16_conv2d_bwd_data_same_ws_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW, int pad) {
    SameDgrad f;
    const int bm = strip_bm((long)B * pad * max(W + 2 * pad, H), (Cin + 63) / 64, 4, 1, true);
    if (!same_dgrad_geom(nullptr, nullptr, nullptr, nullptr, nullptr, B, H, W, Cin, Cout, KH, KW, pad, &f, BK, bm)) return 0;
    const size_t ring = (f.ring_total * f.parts * sizeof(float) + 255) / 256 * 256;
    return ring + gemm_ws_bytes_h(f.g.M, Cin, f.g.K, 1, f.dst_elems);
}

static int same_dgrad_run_h(const void* dy, const void* w_dgrad, const void* w_dgrad_t, void* dx, int B, int H, int W, int Cin,
                            int Cout, int KH, int KW, int pad, void* ws, size_t ws_bytes, void* stream, bool ring_only) {
    SameDgrad f;
    const int bm = strip_bm((long)B * pad * max(W + 2 * pad, H), (Cin + 63) / 64, 4, 1, true);
    if ((Cin & 7) || !same_dgrad_geom(dy, w_dgrad, w_dgrad_t, dx, (float*)ws, B, H, W, Cin, Cout, KH, KW, pad, &f, BK, bm)) return DWC_EINVAL;
    const size_t ring_bytes = (f.ring_total * f.parts * sizeof(float) + 255) / 256 * 256;
    if (!ws || ws_bytes < ring_bytes) return DWC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (!ring_only) {
        int rc = launch_gemm_h(f.g, (const bf16*)w_dgrad, 0, 1, f.o, nullptr, DWC_ACT_NONE, f.dst_elems, (char*)ws + ring_bytes,
                               ws_bytes - ring_bytes, st);
        if (rc != DWC_OK) return rc;
    }
    if (bm == 128) hipLaunchKernelGGL((gemm_strips_kernel_h<128, 64, 2, 2, 2, 1>), dim3(f.max_tiles, f.parts, 4), dim3(256), 0, st, f.ss);
    else hipLaunchKernelGGL((gemm_strips_kernel_h<64, 64, 2, 2, 1, 1>), dim3(f.max_tiles, f.parts, 4), dim3(256), 0, st, f.ss);
    DWC_LAUNCH_CHECK();
    const size_t total = (size_t)B * 2 * pad * (W + H) * (Cin / 4);
    hipLaunchKernelGGL(fold_ring_kernel_h, dim3((total + 255) / 256), dim3(256), 0, st, (bf16*)dx, (const float*)ws, f.ring_elems[0],
                       f.ring_elems[0] + f.ring_elems[1], f.ring_elems[0] + f.ring_elems[1] + f.ring_elems[2], f.parts,
                       f.ring_total, B, H, W, Cin / 4, pad);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

int dwc_bf16_conv2d_bwd_data_same(const void* dy, const void* w_dgrad, const void* w_dgrad_t, void* dx, int B, int H, int W,
                                  int Cin, int Cout, int KH, int KW, int pad, void* ws, size_t ws_bytes, void* stream) {
    return same_dgrad_run_h(dy, w_dgrad, w_dgrad_t, dx, B, H, W, Cin, Cout, KH, KW, pad, ws, ws_bytes, stream, false);
}

/* only the border ring: dx must already hold the interior (dwc_bf16_conv2d_same_halo with the zero rule and the dgrad weights) */
int dwc_bf16_conv2d_bwd_data_ring(const void* dy, const void* w_dgrad, const void* w_dgrad_t, void* dx, int B, int H, int W,
                                  int Cin, int Cout, int KH, int KW, int pad, void* ws, size_t ws_bytes, void* stream) {
    return same_dgrad_run_h(dy, w_dgrad, w_dgrad_t, dx, B, H, W, Cin, Cout, KH, KW, pad, ws, ws_bytes, stream, true);
}

// gradient w.r.t. an NHWC8 image through a stem convolution: 4 pixels x 8 planes per GEMM row (see dwc_conv2d_bwd_data_image)
size_t dwc_bf16_conv2d_bwd_data_image_ws_bytes(int B, int H, int W, int Cout, int KH, int KW, int pad) {
    FwdGeom f;
    if (!image_dgrad_geom(nullptr, nullptr, B, H, W, Cout, KH, KW, pad, &f, 4)) return 0;
    return f.dst_elems * sizeof(bf16);
}

int dwc_bf16_conv2d_bwd_data_image(const void* dy, const void* w_wide, void* dx, int B, int H, int W, int Cout, int KH, int KW,
                                   int pad, void* ws, size_t ws_bytes, void* stream) {
    FwdGeom f;
    if (!image_dgrad_geom(dy, ws, B, H, W, Cout, KH, KW, pad, &f, 4)) return DWC_EINVAL;
    if (!ws || ws_bytes < f.dst_elems * sizeof(bf16)) return DWC_EWORKSPACE;
    const int rc = launch_gemm_h(f.g, (const bf16*)w_wide, 0, 1, f.o, nullptr, DWC_ACT_NONE, f.dst_elems, nullptr, 0,
                                 (hipStream_t)stream);
    if (rc != DWC_OK) return rc;
    const size_t total = (size_t)B * H * W * 2;       // 8 planes = 2 groups of 4 per pixel
    hipLaunchKernelGGL(fold_reflect_kernel_h, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const bf16*)ws,
                       (bf16*)dx, B, H, W, 2, pad, f.g.OW * 4);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

/* the same gradient on conv_narrow_bf16.hip (patch staged once per block, taps dealt to the waves): w_frag = the wide bank in
 * fragment order (see dwc_bf16_conv2d_narrow); scratch as dwc_bf16_conv2d_bwd_data_image_ws_bytes */
int dwc_bf16_conv2d_bwd_data_image_narrow(const void* dy, const void* w_frag, void* dx, int B, int H, int W, int Cout, int KH, int KW,
                                          int pad, void* ws, size_t ws_bytes, void* stream) {
    FwdGeom f;
    if (!image_dgrad_geom(dy, ws, B, H, W, Cout, KH, KW, pad, &f, 4)) return DWC_EINVAL;
    if (!dwc_bf16_conv2d_narrow_ok(B, H, W, Cout, f.g.OH, f.g.OW, KH, KW + 3)) return DWC_EINVAL;
    if (!ws || ws_bytes < f.dst_elems * sizeof(bf16)) return DWC_EWORKSPACE;
    const int rc = dwc_bf16_conv2d_narrow(dy, w_frag, nullptr, ws, B, H, W, Cout, f.g.OH, f.g.OW, KH, KW + 3, f.g.off_h, f.g.off_w,
                                          DWC_ACT_NONE, 0, stream);
    if (rc != DWC_OK) return rc;
    const size_t total = (size_t)B * H * W * 2;
    hipLaunchKernelGGL(fold_reflect_kernel_h, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const bf16*)ws,
                       (bf16*)dx, B, H, W, 2, pad, f.g.OW * 4);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

size_t dwc_bf16_conv2d_bwd_weight_ws_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad) {
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    int splits, chunk;
    wgrad_plan(B * Ho * Wo, KH * KW * Cin, Cout, &splits, &chunk, 1, 64);
    return (size_t)splits * KH * KW * Cin * Cout * sizeof(float);
}

int dwc_bf16_conv2d_bwd_weight(const void* x, const void* dy, float* dw_oihw, int B, int H, int W, int Cin, int Cout, int KH,
                               int KW, int stride, int pad, int cin_real, int cout_real, void* ws, size_t ws_bytes, void* stream) {
    FwdGeom f;
    if (!fwd_geom(x, nullptr, B, H, W, Cin, Cout, KH, KW, stride, pad, &f, MIN_LOG_C)) return DWC_EINVAL;
    if (cin_real > Cin || cout_real > Cout || (Cout & 7)) return DWC_EINVAL;
    return wgrad_launch_h(f, (const bf16*)dy, dw_oihw, Cin, Cout, KH * KW, cin_real, cout_real, ws, ws_bytes, (hipStream_t)stream);
}

size_t dwc_bf16_conv2d_bwd_weight_ex_ws_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride_h, int stride_w,
                                              int pad_h, int pad_w) {
    FwdGeom f;
    if (!fwd_geom_ex(nullptr, nullptr, B, H, W, Cin, Cout, KH, KW, stride_h, stride_w, pad_h, pad_w, &f, MIN_LOG_C)) return 0;
    int splits, chunk;
    wgrad_plan(f.g.M, f.g.K, Cout, &splits, &chunk, 1, 64);
    return (size_t)splits * f.g.K * Cout * sizeof(float);
}

int dwc_bf16_conv2d_bwd_weight_ex(const void* x, const void* dy, float* dw_oihw, int B, int H, int W, int Cin, int Cout, int KH,
                                  int KW, int stride_h, int stride_w, int pad_h, int pad_w, int cin_real, int cout_real, void* ws,
                                  size_t ws_bytes, void* stream) {
    FwdGeom f;
    if (!fwd_geom_ex(x, nullptr, B, H, W, Cin, Cout, KH, KW, stride_h, stride_w, pad_h, pad_w, &f, MIN_LOG_C)) return DWC_EINVAL;
    if (cin_real > Cin || cout_real > Cout || (Cout & 7)) return DWC_EINVAL;
    return wgrad_launch_h(f, (const bf16*)dy, dw_oihw, Cin, Cout, KH * KW, cin_real, cout_real, ws, ws_bytes, (hipStream_t)stream);
}

}  // extern "C"
