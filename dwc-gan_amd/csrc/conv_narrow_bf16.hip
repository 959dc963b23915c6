// bf16 7x7 convolutions from 64 channels to the 8 planes of an NHWC8 image: the fused image heads of the decoder
// (reference networks.py:218-246: tanh x3 + sigmoid, forward) and the data gradient of the 7x7 stems w.r.t. their input image
// (reference networks.py:579-585 backward).  The im2col GEMM ran these at 3-10 % of the MFMA peak: N = 8 fills a quarter of a
// 32-wide tile even in the "wide" form (4 horizontally adjacent pixels x 8 planes = 32 columns, filter bank
// [32][KH][KW+3][64], see ops._prepped 'heads_wide' / 'dgrad_image'), and every input pixel was re-staged once per tap
// (70 times).
//
// Here a workgroup owns a block of 16 rows x 8 pixel groups (= 32 pixels) of one image:
//  * the (16+KH-1) x (32+KW+2) input patch (64 channels, 128-byte rows) is staged ONCE by LDS-DMA (reflect or zero rule
//    applied while staging, chunk swizzle keyed on (patch column >> 2) so that the 8 groups of a row -- 512 bytes apart --
//    fall on different banks);
//  * the KH*(KW+3) taps are dealt round-robin to the 8 waves; a wave multiplies its taps against all 4 group tiles
//    (D[32 columns][32 groups] += W_tap[32][64] . X_tap[64][32 groups]) with its weight fragments read straight from
//    global memory / L2 one tap ahead -- no wave shares a tap, so weights need no LDS and the tap loop has no barrier;
//  * the 8 partial sums meet in LDS (fixed order), bias + activation, 8-byte stores of 4 planes.
// Output [B][OH][OWg][32] bf16 (= the NHWC8 image when OWg*4 is its width); the input window of output (oy, group gx) starts at
// (oy + off_h, 4*gx + off_w): forward off = -pad with the reflect rule; image gradient off = -(K-1) with the zero rule on
// the padded grid (OH = H + 2 pad), folded back by dwc_bf16_reflect_pad_adjoint's kernel as before.
#include <type_traits>

#include "conv_geom.h"

namespace {

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int NB_GROUPS = 8, NCH = 64;

struct NarrowArgs {
    const bf16* x;       // [B][IH][IW][64]
    const bf16* w;       // [tap, padded to a multiple of 8 with zeros][q][lane][8]: MFMA-fragment order of the wide bank (one contiguous
                         // 1 KB read per wave and k-step)
    const float* bias;   // [32] or null
    bf16* y;             // [B][OH][OWg][32]
    int B, IH, IW, OH, OWg, off_h, off_w, act, reflect;
    int blocks_x, blocks_y;
    // (timing-only ablations: -DDWC_DEV_ABLATIONS -DDWC_NARROW_DBG=<bits>: 1 no MFMA, 2 no fragment reads, 4 no patch staging, 8 no reduction)
};

// tanh on planes 0..2, sigmoid on plane 3, planes 4..7 zero (DWC_ACT_HEADS8) for a bf16 RESULT (relative 4e-3): tanh x = 2 sigmoid(2x) - 1 on
// the hardware exp2 / rcp (absolute 2e-7), the odd cubic below 1/16 where that form cancels (relative 2e-6).  The library tanhf / expf /
// IEEE division of dwc_act_apply were 1.0 of the 10.3 thousand cycles a 256-pixel block took (r06, shader-clock probes).  v: the four
// values of planes 4 hi .. 4 hi + 3 of one pixel, bias added.
__device__ __forceinline__ void narrow_heads8_bf16(float (&v)[4], int hi) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float x = v[k];
        const float sg = __builtin_amdgcn_rcpf(1.f + __expf(k < 3 ? -2.f * x : -x));
        const float th = fabsf(x) < 0.0625f ? x * (1.f - x * x * 0.33333334f) : 2.f * sg - 1.f;
        v[k] = hi ? 0.f : (k < 3 ? th : sg);
    }
}

// NB_ROWS = 8: 73.5 KB of LDS and <= 128 registers, so TWO workgroups share a CU and one's patch staging / reduction runs beside
// the other's tap loop (16 rows, one workgroup per CU: 14.5 us per 512-pixel block, most of it exposed staging latency).
template <int KH, int KWW, int NB_ROWS>
__global__ __launch_bounds__(512, NB_ROWS == 8 ? 2 : 1) void conv_narrow_kernel(NarrowArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int MT = NB_ROWS / 4;                     // 32-group tiles per block (8 groups per row)
    constexpr int PR = NB_ROWS + KH - 1, PC = 4 * NB_GROUPS + KWW - 1, PPIX = PR * PC;
    constexpr int PPASS = (PPIX + 63) / 64;
    constexpr int NTAP = KH * KWW;
    constexpr int RED = 8 * 2 * 16 * 64;                // floats of one reduction round (two tiles): [wave][tile][reg][lane]
    constexpr int SMEM_B = PPASS * 64 * NCH * 2 > RED * 4 ? PPASS * 64 * NCH * 2 : RED * 4;
    __shared__ __attribute__((aligned(16))) unsigned char smem_raw[SMEM_B];
    bf16* sP = reinterpret_cast<bf16*>(smem_raw);
    float* sR = reinterpret_cast<float*>(smem_raw);

    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    int bid = blockIdx.x;
    const int bx = bid % a.blocks_x;
    bid /= a.blocks_x;
    const int by = bid % a.blocks_y, n = bid / a.blocks_y;
    const int oy0 = by * NB_ROWS, gx0 = bx * NB_GROUPS;
#if defined(DWC_DEV_ABLATIONS) && defined(DWC_NARROW_DBG)       // compile-time: a run-time switch in the tap loop disturbs its schedule
    constexpr int DBG = DWC_NARROW_DBG;
#else
    constexpr int DBG = 0;
#endif

    // ---- patch: one LDS-DMA pass per 64 pixels; chunk swizzle by (patch column >> 2) ----------------------------------------
    const unsigned x_bytes = (unsigned)a.B * a.IH * a.IW * NCH * 2u;
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.x), 0, x_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    {
        bf16* lp = sP + wave * (8 * NCH);
#pragma unroll
        for (int i = 0; i < PPASS; ++i) {
            const int pp = (t >> 3) + 64 * i;
            const int pr = pp / PC, pc = pp - pr * PC;
            int h = oy0 + pr + a.off_h, w = 4 * gx0 + pc + a.off_w;
            bool ok = pp < PPIX;
            if (a.reflect) {
                h = reflect_idx(h, a.IH);
                w = reflect_idx(w, a.IW);
                // (blocks that hang over the image edge ask for pixels more than one reflection away: the clamp below picks an
                // in-range pixel, their outputs are masked)
            } else {
                ok = ok && (unsigned)h < (unsigned)a.IH && (unsigned)w < (unsigned)a.IW;
            }
            h = min(max(h, 0), a.IH - 1);
            w = min(max(w, 0), a.IW - 1);
            const int lc = (t & 7) ^ ((pc >> 2) & 7);
            const unsigned off = ((unsigned)((n * a.IH + h) * a.IW + w) * NCH + (unsigned)(lc * 8)) * 2u;
            if (!(DBG & 4))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(lp + i * 64 * NCH), 16,
                                                         ok ? off : OOB, 0, 0, 0);
        }
    }

    // ---- this wave's taps ---------------------------------------------------------------------------------------------------
    f32x16 acc[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
    // group of tile m and lane: g = m*32 + l31 -> (row g >> 3, group-in-row g & 7); patch pixel of tap (kh, u): row + kh, 4*gl + u
    int g_base[MT], g_col[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int g = m * 32 + l31;
        g_base[m] = (g >> 3) * PC + 4 * (g & 7);
        g_col[m] = 4 * (g & 7);
    }
    // (read in the prepared [32][Kp] layout a load instruction touched 64 different cache lines -- 17 us per block, all of it
    // address traffic; fragment order makes it one contiguous kilobyte)
    const bf16* wlane = a.w + lane * 8;
    // Weight fragments one tap ahead, two fragment sets rotating by NAME: the tap list is padded to NT_W taps per wave (taps >= NTAP
    // carry zero weights) and the NT_W steps are unrolled, so the set of a step is a compile-time index; the loads are asm statements
    // with hand-counted waits tied to the fragment registers.  (r04.  The first form copied the prefetched set into the working one at
    // the end of every tap -- which drains vmcnt to 0 in front of the copy -- and as plain loads hipcc sinks them to their first use
    // anyway: every tap waited out an L2 round trip, see conv_narrow_x3.hip.)
    constexpr int NT_W = (NTAP + 7) / 8;
    bf16x8 wset[2][4];
    auto load_item = [&](int j, bf16x8 (&dst)[4]) {
        const bf16* src = wlane + (size_t)(wave + 8 * j) * 4 * 512;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[0]) : "v"(src));
        asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(dst[1]) : "v"(src));
        asm volatile("global_load_dwordx4 %0, %1, off offset:2048" : "=v"(dst[2]) : "v"(src));
        asm volatile("global_load_dwordx4 %0, %1, off offset:3072" : "=v"(dst[3]) : "v"(src));
    };
    load_item(0, wset[0]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                    // patch landed (every wave's share)
#pragma unroll
    for (int j = 0; j < NT_W; ++j) {
        if (j + 1 < NT_W) load_item(j + 1, wset[(j + 1) & 1]);
        bf16x8 (&fb)[4] = wset[j & 1];
        if (j + 1 < NT_W) asm volatile("s_waitcnt vmcnt(4)" : "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]));
        else asm volatile("s_waitcnt vmcnt(0)" : "+v"(fb[0]), "+v"(fb[1]), "+v"(fb[2]), "+v"(fb[3]));
        const int tp = wave + 8 * j;
        int kh = tp / KWW, u = tp - kh * KWW;
        if (tp >= NTAP) kh = 0, u = 0;                  // padding tap: zero weights, any patch pixel
        const int d = kh * PC + u;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            const int pp = g_base[m] + d;
            const int sw = ((g_col[m] + u) >> 2) & 7;
            const bf16* p = sP + pp * NCH;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                bf16x8 fa;
                if (DBG & 2) fa = fb[(q + 1) & 3];
                else fa = *reinterpret_cast<const bf16x8*>(p + (((2 * q + hi) ^ sw) << 3));
                if (DBG & 1) asm volatile("" ::"v"(fa));
                else acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[q], fa, acc[m], 0, 0, 0);      // D[column][group]
            }
        }
    }

    // ---- the 8 partial sums meet in LDS, two group tiles per round -----------------------------------------------------------
    const int tile2 = t >> 8, rb = (t >> 6) & 3;        // epilogue thread: tile (of the round), register block, lane
    float bq[4] = {0.f, 0.f, 0.f, 0.f};                 // this thread's four bias values, loaded once (not per round behind a lane condition)
    if (a.bias) {
#pragma unroll
        for (int k = 0; k < 4; ++k) bq[k] = a.bias[8 * rb + 4 * hi + k];
    }
#pragma unroll
    for (int round = 0; round < MT / 2; ++round) {
        if (DBG & 8) {                                  // (ablation: no exchange; wave 0's own sums are stored)
            if (wave == 0 && oy0 < a.OH) a.y[(size_t)blockIdx.x * 64 + lane] = (bf16)(acc[0][lane & 15] + acc[MT - 1][lane & 15]);
            break;
        }
        __syncthreads();                                // patch (round 0) / previous round's sums fully read
#pragma unroll
        for (int mm = 0; mm < 2; ++mm)
#pragma unroll
            for (int r = 0; r < 16; ++r) sR[((wave * 2 + mm) * 16 + r) * 64 + lane] = acc[2 * round + mm][r];
        __syncthreads();
        // thread (tile2, rb, lane): registers 4*rb .. 4*rb+3 of lane -> columns 8*rb + 4*hi + k = pixel rb of the group, planes 4*hi+k
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float s = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) s += sR[((w8 * 2 + tile2) * 16 + 4 * rb + k) * 64 + lane];      // fixed order
            v[k] = s;
        }
        const int g = (2 * round + tile2) * 32 + l31;
        const int oy = oy0 + (g >> 3), gx = gx0 + (g & 7);
        if (oy < a.OH && gx < a.OWg) {
            const int c0 = 8 * rb + 4 * hi;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += bq[k];
            if (a.act == DWC_ACT_HEADS8) {
                narrow_heads8_bf16(v, hi);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = dwc_act_apply(v[k], a.act, c0 + k);
            }
            bf16x4 o;
            o[0] = (bf16)v[0]; o[1] = (bf16)v[1]; o[2] = (bf16)v[2]; o[3] = (bf16)v[3];
            *reinterpret_cast<bf16x4*>(a.y + ((size_t)(n * a.OH + oy) * a.OWg + gx) * 32 + c0) = o;
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------
// The same layer as a PERSISTENT workgroup (r06).  Ablations of the kernel above at B = 384 (profiles/r06_narrow_ablation.txt, 647 us on
// that box): no patch staging -231, no MFMAs and no fragment reads -204, no partial-sum exchange -54, and the remaining skeleton alone
// 231 us -- the parts ADD: a workgroup stages, waits, multiplies, exchanges, and two workgroups per CU do not interleave enough to hide any
// of it; and the skeleton is mostly filter traffic: every wave re-reads its 9 taps x 4 KB from L2 for every 256-pixel block (288 KB per
// block against 74 KB of patch).  Here ONE workgroup per CU walks over the blocks:
//  * a wave's filter taps (9 x 4 fragments = 144 registers) are loaded ONCE and stay in registers for every block;
//  * two patch buffers: the next block's patch is requested (LDS-DMA) before the tap loop of the current one and lands behind it;
//  * the fragment reads are asm ds_read_b128 with counted lgkmcnt, one item (tap, group tile) ahead of the MFMAs -- as compiler-visible LDS
//    reads every one of them would be made to wait for the DMA in flight (vmcnt(0): the compiler cannot tell the buffers apart);
//  * the partial-sum exchange aliases the patch buffer just consumed, as above.
// ------------------------------------------------------------------------------------------
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define NRW_DSR(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr))

template <int N, class F>
__device__ __forceinline__ void nrw_for(F&& f) {
    if constexpr (N > 0) {
        nrw_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

template <int KH, int KWW>
__global__ __launch_bounds__(512, 1) void conv_narrow_persist_kernel(NarrowArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NB_ROWS = 8, MT = NB_ROWS / 4;
    constexpr int PR = NB_ROWS + KH - 1, PC = 4 * NB_GROUPS + KWW - 1, PPIX = PR * PC;
    constexpr int PPASS = (PPIX + 63) / 64;
    constexpr int NTAP = KH * KWW, NT_W = (NTAP + 7) / 8, NITEM = NT_W * MT;
    constexpr int PBUF = PPASS * 64 * NCH;              // elements of one patch buffer (73 728 bytes)
    constexpr int RED = 8 * 2 * 16 * 64;                // floats of the exchange: [wave][tile][reg][lane]
    static_assert(RED * 4 <= PBUF * 2 && MT == 2, "the exchange aliases one patch buffer; one round of two tiles");
#ifdef DWC_NARROW_ALL_STAGE                             // (A/B build: every wave stages its own rows)
    constexpr bool HALF_STAGE = false;
#else
    constexpr bool HALF_STAGE = true;
#endif
    __shared__ __attribute__((aligned(128))) unsigned char smem_raw[2 * PBUF * 2];
    bf16* sP = reinterpret_cast<bf16*>(smem_raw);
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) unsigned char*)smem_raw;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
#if defined(DWC_DEV_ABLATIONS) && defined(DWC_NARROW_DBG)       // timing-only ablations (WRONG results), as in conv_narrow_kernel
    constexpr int DBG = DWC_NARROW_DBG;
#else
    constexpr int DBG = 0;
#endif
    long long probe[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#define NRW_PROBE(i)                                        \
    do {                                                    \
        if constexpr ((DBG & 32) != 0) {                    \
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
            probe[i] = (long long)clock64();                \
        }                                                   \
    } while (0)

    // ---- this wave's taps, once -----------------------------------------------------------------------------------------------
    bf16x8 wreg[NT_W][4];
    {
        const bf16* wlane = a.w + lane * 8;
        nrw_for<NT_W * 4>([&](auto K) {
            constexpr int j = decltype(K)::value / 4, q = decltype(K)::value % 4;
            wreg[j][q] = *reinterpret_cast<const bf16x8*>(wlane + ((size_t)(wave + 8 * j) * 4 + q) * 512);
        });
    }
    const unsigned x_bytes = (unsigned)a.B * a.IH * a.IW * NCH * 2u;
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.x), 0, x_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    // Patch staging, one patch ROW per wave and turn (rows wave and wave + 8), 8 pixels (1 KB) per LDS-DMA instruction.  The address work is
    // what the first persistent version lost its time on: one instruction per 64 consecutive patch pixels cost a division by the patch
    // width, two reflections, clamps and three multiplies per lane and instruction -- 339 vector instructions per wave and block against 72
    // MFMAs, and with all waves of the workgroup in the same phase nothing overlapped them.  Row-wise, the row's base is scalar arithmetic
    // and the column term (reflected column x 128 B + swizzled 16-byte chunk) is a per-lane constant of the block COLUMN, kept in six
    // registers while the workgroup walks down that column: one add and one select per instruction.
    constexpr int NJ = (PC + 7) / 8;                    // instructions per row; the last one holds PC - 8 (NJ - 1) pixels
    constexpr int TAIL_PIX = PC - 8 * (NJ - 1);
    int wq[NJ];                                         // byte offset of this lane's pixel / chunk inside an image row, or -1: no data
    auto column_terms = [&](int bx) {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int pc = 8 * j + (lane >> 3);
            int w = 4 * bx * NB_GROUPS + pc + a.off_w;
            bool ok = pc < PC;
            if (a.reflect) w = reflect_idx(w, a.IW);
            else ok = ok && (unsigned)w < (unsigned)a.IW;
            w = min(max(w, 0), a.IW - 1);
            const int lc = (lane & 7) ^ ((pc >> 2) & 7);
            wq[j] = ok ? w * (NCH * 2) + lc * 16 : -1;
        }
    };
    // A row's scalar part: source byte offset of the row start, or OOB (zero rule, row outside the image).
    auto row_base = [&](int n, int by, int pr) -> unsigned {
        int h = by * NB_ROWS + pr + a.off_h;
        bool okh = true;
        if (a.reflect) h = reflect_idx(h, a.IH);
        else okh = (unsigned)h < (unsigned)a.IH;
        h = min(max(h, 0), a.IH - 1);
        return okh ? (unsigned)((n * a.IH + h) * a.IW) * (NCH * 2u) : OOB;
    };
    // DMA instruction j of patch row pr: 8 pixels.  (The vector-memory path takes ~150 cycles per such instruction when all eight waves
    // issue their 12 back to back: 1.8 of the 10.3 thousand cycles of a block by the shader-clock probes (-DDWC_NARROW_DBG=32).  Issued one
    // at a time between the MFMAs of the tap loop they cost the same -- the issuing wave is held, not the queue -- so they stay in front.)
    auto stage_one = [&](unsigned rowbase, int pr, int j, int buf) {
        const unsigned off = (rowbase != OOB && wq[j] >= 0) ? rowbase + (unsigned)wq[j] : OOB;
        bf16* lp = sP + buf * PBUF + (pr * PC + j * 8) * NCH;
        if (DBG & 4) return;
        if (j + 1 < NJ || TAIL_PIX == 8) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)lp, 16, off, 0, 0, 0);
        } else if (lane < 8 * TAIL_PIX) {               // (the lanes past the row's end must write NOTHING: the next row starts there)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)lp, 16, off, 0, 0, 0);
        }
    };
    auto stage = [&](int n, int by, int buf, int wv) {  // the rows wave wv stages: wv and wv + 8 (scalar)
        const bool two_rows = wv + 8 < PR;
        const unsigned rb0 = row_base(n, by, wv), rb1 = two_rows ? row_base(n, by, wv + 8) : OOB;
#pragma unroll
        for (int j = 0; j < NJ; ++j) stage_one(rb0, wv, j, buf);
        if (two_rows) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) stage_one(rb1, wv + 8, j, buf);
        }
    };
    // fragment addresses (bytes inside a patch buffer) of item (tap slot j, tile m), 16-channel step 0: the other steps XOR (q << 5)
    int g_base[MT], g_col[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int g = m * 32 + l31;
        g_base[m] = (g >> 3) * PC + 4 * (g & 7);
        g_col[m] = 4 * (g & 7);
    }
    auto frag_addr = [&](int j, int m, unsigned base) -> unsigned {
        const int tp = wave + 8 * j;
        int kh = tp / KWW, u = tp - kh * KWW;
        if (tp >= NTAP) kh = 0, u = 0;                  // padding tap: zero weights, any patch pixel
        const int pp = g_base[m] + kh * PC + u;
        const int sw = ((g_col[m] + u) >> 2) & 7;
        return base + (unsigned)(pp * NCH + ((hi ^ sw) << 3)) * 2u;
    };
    const int tile2 = t >> 8, rb = (t >> 6) & 3;        // epilogue thread: tile, register block
    float bq[4] = {0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
        for (int k = 0; k < 4; ++k) bq[k] = a.bias[8 * rb + 4 * hi + k];
    }

    // XCD-aware walk: workgroup w runs on XCD w % 8 (round-robin dispatch).  Each XCD owns a contiguous eighth of the IMAGES; each of its
    // workgroups walks block COLUMNS of them top to bottom, one after the other.  The 6 halo rows a block shares with the block below are
    // then re-read a few microseconds later by the same CU, the halo columns by a neighbour CU of the same XCD: both from that XCD's L2.
    // (With blk = w + k * grid the vertical neighbour ran on another XCD and every halo row came from HBM / MALL again: 2.2x the tensor.)
    const int xcd = blockIdx.x & 7, slot = (int)blockIdx.x >> 3, per_x = (int)gridDim.x >> 3;
    const int imgs = (a.B + 7) >> 3, n_first = xcd * imgs, n_imgs = min(imgs, a.B - n_first);
    const int ncols = n_imgs > 0 ? n_imgs * a.blocks_x : 0;      // block columns of this XCD
    // step k of this workgroup: column slot + per_x * (k / blocks_y), block row k % blocks_y
    auto locate = [&](int k, int& n, int& by, int& bx) -> bool {
        const int col = slot + per_x * (k / a.blocks_y);
        by = k % a.blocks_y;
        n = n_first + col / a.blocks_x;
        bx = col % a.blocks_x;
        return col < ncols;
    };
    int n, by, bx;
    if (!locate(0, n, by, bx)) return;
    int bx_terms = bx;
    column_terms(bx);
    stage(n, by, 0, wave);
    int cur = 0;
    for (int k = 0;; ++k, cur ^= 1) {
        if (!locate(k, n, by, bx)) break;
        int n2, by2, bx2;
        const bool more = locate(k + 1, n2, by2, bx2);
        const int oy0 = by * NB_ROWS, gx0 = bx * NB_GROUPS;
        NRW_PROBE(0);
        if (!(DBG & 16) || k == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this block's patch (own share); the first time, the filter taps
        __builtin_amdgcn_s_barrier();                                 // ... everyone's share; the other buffer's exchange is fully read
        NRW_PROBE(1);
        if (more) {
            if (bx2 != bx_terms) {                                    // (the walk entered another block column: once per blocks_y blocks)
                bx_terms = bx2;
                column_terms(bx2);
            }
            // ONE half of the waves requests the next patch -- its own rows and its SIMD partner's (wave ^ 4), the halves taking turns --
            // while the other half is already in the tap loop: a wave issuing LDS-DMA instructions is held ~140 cycles per instruction
            // when all eight issue at once (the "stage" probe: 1.2-1.8 of a block's 8.6 thousand cycles with no MFMA running).
            if (HALF_STAGE) {
                if ((wave >> 2) == (k & 1)) {
                    stage(n2, by2, cur ^ 1, wave);
                    stage(n2, by2, cur ^ 1, wave ^ 4);
                }
            } else {
                stage(n2, by2, cur ^ 1, wave);
            }
        }
        NRW_PROBE(2);
        const unsigned pbase = lds0 + (unsigned)cur * (PBUF * 2u);

        f32x16 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
        // 72 fragment reads per block, a ring of four registers, three reads ahead of the MFMA that consumes them
        constexpr int NRD = NITEM * 4, AHEAD = 3;
        bf16x8 fr[AHEAD + 1];
        unsigned ad_item = frag_addr(0, 0, pbase);
        nrw_for<AHEAD>([&](auto K) {
            constexpr int k = decltype(K)::value;
            bf16x8& dst = fr[k];                          // (asm operands inside a generic lambda: through local names)
            const unsigned adr = ad_item ^ (unsigned)(k << 5);
            NRW_DSR(dst, adr);
        });
        unsigned ad_next = ad_item;
        nrw_for<NRD>([&](auto K) {                       // (compile-time indices: the filter taps and the ring must stay registers)
            constexpr int k = decltype(K)::value;
            constexpr int it = k / 4, q = k % 4, j = it / MT, m = it % MT, kn = k + AHEAD;
            if constexpr (kn < NRD) {
                if constexpr (kn % 4 == 0) ad_next = frag_addr((kn / 4) / MT, (kn / 4) % MT, pbase);
                bf16x8& dst = fr[kn % (AHEAD + 1)];
                const unsigned adr = ad_next ^ (unsigned)((kn % 4) << 5);
                NRW_DSR(dst, adr);
            }
            bf16x8& fa = fr[k % (AHEAD + 1)];
            if constexpr (kn < NRD) asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(fa));
            else if constexpr (NRD - 1 - k == 2) asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(fa));
            else if constexpr (NRD - 1 - k == 1) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(fa));
            else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa));
            if constexpr (DBG & 1) asm volatile("" ::"v"(fa));
            else acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wreg[j][q], fa, acc[m], 0, 0, 0);      // D[column][group]
        });
        if constexpr (DBG & 8) {                        // (ablation: no exchange)
            if (wave == 0 && k == 0) a.y[(size_t)blockIdx.x * 64 + lane] = (bf16)(acc[0][lane & 15] + acc[1][lane & 15]);
            continue;
        }

        // ---- the 8 partial sums meet in the patch buffer just consumed ---------------------------------------------------------
        // (asm LDS stores and loads, like the fragment reads: as compiler-visible LDS accesses they would be made to wait for the next block's
        // patch -- vmcnt(0) -- which then could no longer land behind the exchange and the epilogue, only behind the tap loop)
        // (the MFMA results feed asm statements: the wait states between a matrix instruction and a reader of its result, which the
        // compiler inserts for instructions it knows, are spelled out -- 8-pass MFMA -> LDS data operand: 18)
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]));
        NRW_PROBE(3);
        __builtin_amdgcn_s_barrier();
        NRW_PROBE(4);                   // every wave is past its fragment reads of this buffer (lgkmcnt(0) above).  A bare
                                                        // barrier: __syncthreads() carries a fence that waits for the DMA in flight
        {
            const unsigned wad = pbase + (unsigned)(wave * 2 * 16 * 64 + lane) * 4u;      // [wave][tile][reg][lane] floats
            nrw_for<16>([&](auto I) {                   // two registers per store, 256 bytes (one register row) apart
                constexpr int i = 2 * decltype(I)::value;
                const float v0 = acc[i / 16][i % 16], v1 = acc[i / 16][i % 16 + 1];
                const unsigned adr = wad;
                asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(adr), "v"(v0), "v"(v1), "i"(i), "i"(i + 1));
            });
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        NRW_PROBE(5);
        __builtin_amdgcn_s_barrier();
        NRW_PROBE(6);
        float v[4];
        {
            // thread (tile2, rb, lane): registers 4 rb + k of the eight waves' tile `tile2`; wave w8's copy is 32 register rows further on
            const unsigned rad = pbase + (unsigned)((tile2 * 16 + 4 * rb) * 64 + lane) * 4u;
            f32x2 pv[4][4];                             // [k][pair of waves]
            nrw_for<16>([&](auto I) {
                constexpr int k = decltype(I)::value / 4, pr2 = decltype(I)::value % 4;
                f32x2& dst = pv[k][pr2];
                const unsigned adr = rad;
                asm volatile("ds_read2st64_b32 %0, %1 offset0:%2 offset1:%3" : "=v"(dst) : "v"(adr), "i"(2 * pr2 * 32 + k), "i"((2 * pr2 + 1) * 32 + k));
            });
            nrw_for<4>([&](auto K) {
                constexpr int k = decltype(K)::value;
                f32x2 (&p4)[4] = pv[k];
                if constexpr (k == 0) asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(p4[0]), "+v"(p4[1]), "+v"(p4[2]), "+v"(p4[3]));
                else if constexpr (k == 1) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(p4[0]), "+v"(p4[1]), "+v"(p4[2]), "+v"(p4[3]));
                else if constexpr (k == 2) asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(p4[0]), "+v"(p4[1]), "+v"(p4[2]), "+v"(p4[3]));
                else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(p4[0]), "+v"(p4[1]), "+v"(p4[2]), "+v"(p4[3]));
                float sum = 0.f;
#pragma unroll
                for (int w2 = 0; w2 < 4; ++w2) sum = (sum + p4[w2][0]) + p4[w2][1];             // waves 0..7 in order
                v[k] = sum;
            });
        }
        NRW_PROBE(7);
        const int g = tile2 * 32 + l31;
        const int oy = oy0 + (g >> 3), gx = gx0 + (g & 7);
        if (oy < a.OH && gx < a.OWg) {
            const int c0 = 8 * rb + 4 * hi;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += bq[k];
            if (a.act == DWC_ACT_HEADS8) {
                narrow_heads8_bf16(v, hi);
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k) v[k] = dwc_act_apply(v[k], a.act, c0 + k);
            }
            bf16x4 o;
            o[0] = (bf16)v[0]; o[1] = (bf16)v[1]; o[2] = (bf16)v[2]; o[3] = (bf16)v[3];
            *reinterpret_cast<bf16x4*>(a.y + ((size_t)(n * a.OH + oy) * a.OWg + gx) * 32 + c0) = o;
        }
        NRW_PROBE(8);
#if defined(DWC_DEV_ABLATIONS) && defined(DWC_NARROW_DBG)
        if constexpr ((DBG & 32) != 0) {
            if (blockIdx.x == 8 && t == 0 && (k == 6 || k == 7 || k == 20))
                printf("k %d: wait %lld barrier %lld stage %lld taps %lld bar %lld xwrite %lld bar %lld xread %lld epilogue %lld | whole %lld\n", k,
                       probe[1] - probe[0], 0ll, probe[2] - probe[1], probe[3] - probe[2], probe[4] - probe[3], probe[5] - probe[4],
                       probe[6] - probe[5], probe[7] - probe[6], probe[8] - probe[7], probe[8] - probe[0]);
        }
#endif
    }
#endif
}

// ------------------------------------------------------------------------------------------
// The opposite shape: 7x7 convolutions from the 8 planes of an NHWC8 image to 64 channels -- the stems (forward, reference
// networks.py:163-166 / :60-66) and the data gradient of the image heads.  K = 49 taps x 8 planes: one MFMA k-step is TWO taps
// (lane half hi takes tap 2j+hi: 8 planes = one 16-byte pixel of the patch), 25 steps.  The whole filter (25 x 64 x 32 bytes
// = 50 KB, halves of a row swapped by (row>>3)&1 as in conv_halo_x3.hip) stays in LDS while the workgroup walks over 16x16-pixel
// blocks (persistent: gridDim.x workgroups stride over the blocks); the patch is 22x22 pixels x 16 bytes.  4 waves, 2x2 tiles
// of 32 pixels x 32 channels each; output through LDS as 16-byte channel chunks.  The layer is bound by its 64-channel
// output (128 bytes per pixel), not by the matrix pipe.
// ------------------------------------------------------------------------------------------
struct StemArgs {
    const bf16* x;       // [B][IH][IW][8]
    const bf16* w;       // [25][64][16]: k-step j, channel co, (tap 2j + h, plane p) at ((h ^ ((co>>3)&1))*8 + p)
    const float* bias;   // [64] or null
    bf16* y;             // [B][OH][OW][64]
    int B, IH, IW, OH, OW, off, act, reflect;
    int blocks_x, blocks_y, nblocks;
    // crop > 0 (data gradient of the image heads: y is the gradient of the PADDED tensor): an output pixel whose cropped coordinate
    // lies inside CH x CW goes straight to `inner` ([B][CH][CW][64]), only the border ring to y; a band fold follows
    int crop, CH, CW;
    bf16* inner;
};

__global__ __launch_bounds__(256, 2) void conv_stem_kernel(StemArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int KS = 7, PW = 16 + KS - 1, PPIX = PW * PW, NKS = 25;
    constexpr int W_EL = NKS * 64 * 16;                 // 25600 elements = 51200 bytes
    constexpr int P_EL = 512 * 8;                       // one patch buffer: 484 pixels x 8 planes (whole DMA instructions)
    constexpr int LDC = 64 + 8;                         // epilogue staging pitch
    constexpr int OPER = W_EL + P_EL;
    constexpr int SMEM = OPER + 128 * LDC;             // 77.6 KB: two workgroups per CU; the epilogue stages 128 pixels at a time
    // (a second patch buffer + 64-pixel epilogue phases, to prefetch the next block's patch, measured 25 % slower: the four
    // extra barriers per block cost more than the exposed 8 KB patch load)
    __shared__ __attribute__((aligned(16))) bf16 smem[SMEM];
    bf16* sW = smem;
    bf16* sP = smem + W_EL;
    bf16* sC = smem + OPER;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave;                                // wave w: pixel tiles 2w, 2w+1 (64 pixels); both channel tiles
    // ---- the filter, once -------------------------------------------------------------------------------------------------
    {
        const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.w), 0, W_EL * 2u, 0x00020000);
#pragma unroll
        for (int i = 0; i < (W_EL * 2 + 4095) / 4096; ++i) {          // 256 lanes x 16 bytes per instruction
            const unsigned off = (unsigned)(i * 4096 + t * 16);
            if (off < W_EL * 2u)                        // (masked lanes write nothing: the tail must not spill zeros over the patch)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(sW + i * 2048 + wave * 512),
                                                         16, off, 0, 0, 0);
        }
    }
    const unsigned x_bytes = (unsigned)a.B * a.IH * a.IW * 16u;
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.x), 0, x_bytes, 0x00020000);
    auto stage_patch = [&](int blk, int buf) {          // 2 DMA instructions of 256 pixels
        int bid = blk;
        const int bx = bid % a.blocks_x;
        bid /= a.blocks_x;
        const int by = bid % a.blocks_y, n = bid / a.blocks_y;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int pp = t + 256 * i;
            const int pr = pp / PW, pc = pp - pr * PW;
            int h = by * 16 + pr + a.off, w = bx * 16 + pc + a.off;
            bool ok = pp < PPIX;
            if (a.reflect) {
                h = reflect_idx(h, a.IH);
                w = reflect_idx(w, a.IW);
            } else {
                ok = ok && (unsigned)h < (unsigned)a.IH && (unsigned)w < (unsigned)a.IW;
            }
            h = min(max(h, 0), a.IH - 1);
            w = min(max(w, 0), a.IW - 1);
            const unsigned off = (unsigned)((n * a.IH + h) * a.IW + w) * 16u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(sP + buf * P_EL + i * 2048 + wave * 512),
                                                     16, ok ? off : 0x80000000u, 0, 0, 0);
        }
    };
    // fragment addressing: pixel tile i of this wave: block pixel pb = (2*wm + i)*32 + l31 -> patch pixel (pb>>4)*PW + (pb&15)
    int pp0[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int pb = (2 * wm + i) * 32 + l31;
        pp0[i] = (pb >> 4) * PW + (pb & 15);
    }
    const int b_off = l31 * 16 + ((hi ^ ((l31 >> 3) & 1)) * 8);
    const float slope = dwc_act_slope(a.act);
    f32x4 bvs[2][4];                                    // bias vectors of this lane's columns, loaded once for all blocks
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
            bvs[c][q4] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + c * 32 + 8 * q4 + 4 * hi) : f32x4{0.f, 0.f, 0.f, 0.f};

    constexpr int buf = 0;
#if defined(DWC_DEV_ABLATIONS) && defined(DWC_STEM_DBG)
    long long probe[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int kiter = 0;
#define STEM_PROBE(i) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); probe[i] = (long long)clock64(); } while (0)
#else
#define STEM_PROBE(i) do { } while (0)
#endif
    for (int blk = blockIdx.x; blk < a.nblocks; blk += gridDim.x) {
        int bid = blk;
        const int bx = bid % a.blocks_x;
        bid /= a.blocks_x;
        const int by = bid % a.blocks_y, n = bid / a.blocks_y;
        const int oy0 = by * 16, ox0 = bx * 16;
        STEM_PROBE(0);
        __syncthreads();                                              // every wave is past the previous block's reads
        STEM_PROBE(1);
        stage_patch(blk, 0);
        STEM_PROBE(2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // this block's patch (and, the first time, the filter)
        __syncthreads();
        STEM_PROBE(3);
        const bf16* p = sP + buf * P_EL;

        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll 5
        for (int j = 0; j < NKS; ++j) {
            const int tp = min(2 * j + hi, KS * KS - 1);             // (tap 49 of the last step has zero weights)
            const int kh = tp / KS, kw = tp - kh * KS;
            const int d = kh * PW + kw;
            bf16x8 fa[2], fb[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) fa[i] = *reinterpret_cast<const bf16x8*>(p + (pp0[i] + d) * 8);
#pragma unroll
            for (int c = 0; c < 2; ++c) fb[c] = *reinterpret_cast<const bf16x8*>(sW + (j * 64 + c * 32) * 16 + b_off);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int c = 0; c < 2; ++c) acc[i][c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[c], fa[i], acc[i][c], 0, 0, 0);
        }
        STEM_PROBE(4);
        // ---- epilogue: D[channel][pixel]: lane = pixel, 4 consecutive channels per register quad -> LDS -> 16-byte chunks,
        // 128 pixels (the tiles of two waves) per phase -----------------------------------------------------------------------
        // (bias, activation and rounding by ALL four waves at once, results packed in registers: the staging buffer holds 128 pixels, so the
        // waves write it in two turns -- and with the arithmetic inside the turn, 2 300 cycles of vector work per turn by the shader-clock
        // probes, the other pair of waves stood at the barrier for it: 4 600 of a block's 13 000 cycles.  r06.)
        bf16x4 packed[2][2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    f32x4 v = {acc[i][c][4 * q4], acc[i][c][4 * q4 + 1], acc[i][c][4 * q4 + 2], acc[i][c][4 * q4 + 3]};
                    v += bvs[c][q4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) packed[i][c][q4][k] = (bf16)dwc_act_simple(v[k], slope);
                }
#pragma unroll
        for (int ph = 0; ph < 2; ++ph) {
            if ((wm >> 1) == ph) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int row = (2 * (wm & 1) + i) * 32 + l31;       // row inside the phase's 128 pixels
#pragma unroll
                    for (int c = 0; c < 2; ++c)
#pragma unroll
                        for (int q4 = 0; q4 < 4; ++q4)
                            *reinterpret_cast<bf16x4*>(sC + row * LDC + c * 32 + 8 * q4 + 4 * hi) = packed[i][c][q4];
                }
            }
            if (ph == 0) STEM_PROBE(6);
            __syncthreads();
            if (ph == 0) STEM_PROBE(7);
            for (int idx = t; idx < 128 * 8; idx += 256) {
                const int row = idx >> 3, ch = idx & 7;
                const int pb = ph * 128 + row;
                const int yy = oy0 + (pb >> 4), xx = ox0 + (pb & 15);
                if (yy < a.OH && xx < a.OW) {
                    bf16* d = a.y + ((size_t)(n * a.OH + yy) * a.OW + xx) * 64 + ch * 8;
                    if (a.crop) {
                        const int cy = yy - a.crop, cx = xx - a.crop;
                        if ((unsigned)cy < (unsigned)a.CH && (unsigned)cx < (unsigned)a.CW)
                            d = a.inner + ((size_t)(n * a.CH + cy) * a.CW + cx) * 64 + ch * 8;
                    }
                    *reinterpret_cast<bf16x8*>(d) = *reinterpret_cast<const bf16x8*>(sC + row * LDC + ch * 8);
                }
            }
            if (ph == 0) __syncthreads();
        }
        STEM_PROBE(5);
#if defined(DWC_DEV_ABLATIONS) && defined(DWC_STEM_DBG)
        if (blockIdx.x == 9 && t == 0 && (kiter == 5 || kiter == 6 || kiter == 12))
            printf("stem k %d: barrier %lld stage-issue %lld patch-wait %lld taps %lld epilogue %lld (phase-0 convert+write %lld, barrier %lld, rest %lld) | whole %lld\n", kiter, probe[1] - probe[0],
                   probe[2] - probe[1], probe[3] - probe[2], probe[4] - probe[3], probe[5] - probe[4], probe[6] - probe[4], probe[7] - probe[6], probe[5] - probe[7], probe[5] - probe[0]);
        ++kiter;
#endif
    }
#endif
}

bool stem_ok(int B, int IH, int IW, int OH, int OW, int K, int act) {
    return B > 0 && K == 7 && IH >= 7 && IW >= 7 && OH > 0 && OW > 0 && act <= DWC_ACT_LRELU &&
           (size_t)B * IH * IW * 16 < 0x80000000ull;
}

// ------------------------------------------------------------------------------------------
// Weight gradients of both 7x7 shapes: a correlation between an 8-plane image A and a 64-channel tensor Bt over the pixels q of a
// grid, for the 49 offsets:   C[tap][plane][c] = sum_q A[q + tap + offA][plane] * Bt[q + offB][c].
//   stems:  A = x image (reflect rule, offA = -3), Bt = dY, grid = H x W            -> dW[c][plane][kh][kw] = C
//   heads:  A = gradient image g (zero rule, offA = -6), Bt = x (reflect, offB = -3), grid = (H+6) x (W+6) padded positions
//           -> dW[plane][c][6-kh][6-kw] = C          (x_pad[q] * g[q - tap], summed over the padded grid)
// The im2col kernel ran them at 5-6 % of the peak.  Here an MFMA row tile is 4 horizontally adjacent taps x 8 planes = the 64
// contiguous bytes of 4 patch pixels (pixel-major 16-byte rows), fetched with the transposing LDS read at a per-lane pixel
// offset; 7 filter rows x 2 groups of 4 columns (the 8th column is a dummy) = 14 row tiles, dealt to the 8 waves; contraction
// over 8x16-pixel units (A patch 14 x 24 pixels = 5 KB, Bt block 16 KB, double buffered); fp32 slabs per pixel split, summed in
// a fixed order by smallk_reduce_kernel.
// ------------------------------------------------------------------------------------------
struct SmallWgradArgs {
    const bf16* a8;      // [B][AH][AW][8]
    const bf16* b64;     // [B][BH][BW][64]
    float* slab;         // [splits][14*32][64]
    int B, AH, AW, BH, BW, GH, GW, offA, offB, reflA, reflB;
    int units_x, units_per_img, total_units, units_per_split;
};

__global__ __launch_bounds__(512, 2) void smallk_wgrad_kernel(SmallWgradArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int PCA = 24, PRA = 14;                   // A patch: (8+6) rows x (16+7 -> 24) pixels
    constexpr int A_EL = 512 * 8, B_EL = 128 * 64;      // one DMA instruction of pixels; 128 pixels x 64 channels
    __shared__ __attribute__((aligned(16))) bf16 smem[2 * (A_EL + B_EL)];
    bf16* sA = smem;
    bf16* sB = smem + 2 * A_EL;
    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int split = blockIdx.x;
    const int u0 = split * a.units_per_split, u1 = min(a.total_units, u0 + a.units_per_split);
    const __amdgpu_buffer_rsrc_t rsrc_a =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.a8), 0, (unsigned)a.B * a.AH * a.AW * 16u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.b64), 0, (unsigned)a.B * a.BH * a.BW * 128u, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;

    auto stage_unit = [&](int u, int buf) {
        const int n = u / a.units_per_img, ur = u - n * a.units_per_img;
        const int uy = ur / a.units_x, ux = ur - uy * a.units_x;
        const int gy0 = uy * 8, gx0 = ux * 16;
        {   // A patch: lane t = patch pixel t
            const int pr = t / PCA, pc = t - pr * PCA;
            int h = gy0 + pr + a.offA, w = gx0 + pc + a.offA;
            bool ok = t < PRA * PCA;
            if (a.reflA) {
                h = reflect_idx(h, a.AH);
                w = reflect_idx(w, a.AW);
            } else {
                ok = ok && (unsigned)h < (unsigned)a.AH && (unsigned)w < (unsigned)a.AW;
            }
            h = min(max(h, 0), a.AH - 1);
            w = min(max(w, 0), a.AW - 1);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a, (__attribute__((address_space(3))) void*)(sA + buf * A_EL + wave * 512), 16,
                                                     ok ? (unsigned)((n * a.AH + h) * a.AW + w) * 16u : OOB, 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {   // Bt block: chunk id -> (pixel m, slot c); pixels outside the grid contribute zeros
            const int id = t + 512 * i;
            const int m = id >> 3, c = id & 7;
            const int gy = gy0 + (m >> 4), gx = gx0 + (m & 15);
            int h = gy + a.offB, w = gx + a.offB;
            bool ok = gy < a.GH && gx < a.GW;
            if (a.reflB) {
                h = reflect_idx(h, a.BH);
                w = reflect_idx(w, a.BW);
            } else {
                ok = ok && (unsigned)h < (unsigned)a.BH && (unsigned)w < (unsigned)a.BW;
            }
            h = min(max(h, 0), a.BH - 1);
            w = min(max(w, 0), a.BW - 1);
            const int lc = c ^ (4 * ((m >> 1) & 1));
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, (__attribute__((address_space(3))) void*)(sB + buf * B_EL + i * 4096 + wave * 512),
                                                     16, ok ? ((unsigned)((n * a.BH + h) * a.BW + w) * 64u + (unsigned)(lc * 8)) * 2u : OOB,
                                                     0, 0, 0);
        }
    };

    // this wave's row tiles (tap groups): g = wave and wave + 8 (< 14): filter row g >> 1, columns 4*(g & 1) ..
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][c][r] = 0.f;
    const int ng = wave + 8 < 14 ? 2 : 1;
    const int li = lane & 15, gam = (lane >> 4) & 1, hi = lane >> 5;
    const int tq = li >> 2, tp = li & 3;
    const int pxl = 8 * hi + tq;
    const int row_el = 16 * gam + 4 * tp;               // (tap-in-group, plane) row -> element offset from the tap group's first pixel
    // Fragment reads as asm statements with hand-counted waits tied to the fragment registers: as builtins hipcc cannot tell them
    // from the LDS-DMA's target and drains vmcnt(0) -- the unit staged a moment ago -- in front of the next read (see
    // wgrad_halo_kernel in conv_halo_bf16.hip).  One k-step (a row of 16 pixels) of fragments is read ahead of the MFMAs.
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)sA;
    const unsigned ldsB = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)sB;
    auto tr_read = [](bf16x4& dst, unsigned addr) { asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst) : "v"(addr)); };
    bf16x4 fa[2][2][2], fb[2][2][2];                    // [set][tile][half]
    auto fetch = [&](auto setc, unsigned pbase, unsigned dbase, int ks) {
        constexpr int set = decltype(setc)::value;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            const int col = ct * 32 + 16 * gam + 4 * tp;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int m = ks * 16 + pxl + 4 * half;
                bf16x4& dst = fb[set][ct][half];
                tr_read(dst, dbase + (unsigned)(m * 64 + (((col >> 3) ^ (4 * ((m >> 1) & 1))) << 3) + (col & 7)) * 2u);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int g = wave + 8 * (i < ng ? i : 0);       // (ng == 1: the second tile re-reads the first, unused)
            const int kh = g >> 1, kw0 = 4 * (g & 1);
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int pp = (ks + kh) * PCA + pxl + 4 * half + kw0;
                bf16x4& dst = fa[set][i][half];
                tr_read(dst, pbase + (unsigned)(pp * 8 + row_el) * 2u);
            }
        }
    };
    auto compute = [&](auto setc, auto youngerc) {
        constexpr int set = decltype(setc)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bf16x4 &a0 = fa[set][i][0], &a1 = fa[set][i][1], &b0 = fb[set][i][0], &b1 = fb[set][i][1];
            asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a0), "+v"(a1), "+v"(b0), "+v"(b1) : "n"(decltype(youngerc)::value));
        }
        const bf16x8 fb0 = __builtin_shufflevector(fb[set][0][0], fb[set][0][1], 0, 1, 2, 3, 4, 5, 6, 7);
        const bf16x8 fb1 = __builtin_shufflevector(fb[set][1][0], fb[set][1][1], 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if (i < ng) {
                const bf16x8 fav = __builtin_shufflevector(fa[set][i][0], fa[set][i][1], 0, 1, 2, 3, 4, 5, 6, 7);
                acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fav, fb0, acc[i][0], 0, 0, 0);
                acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fav, fb1, acc[i][1], 0, 0, 0);
            }
        }
    };
    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, 1> S1;
    typedef std::integral_constant<int, 8> Y8;             // the 8 reads of the k-step fetched ahead may stay in flight

    if (u0 < u1) {
        stage_unit(u0, 0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int buf = 0;
        for (int u = u0; u < u1; ++u) {
            if (u + 1 < u1) stage_unit(u + 1, buf ^ 1);
            const unsigned pbase = lds0 + (unsigned)(buf * A_EL) * 2u, dbase = ldsB + (unsigned)(buf * B_EL) * 2u;
            fetch(S0{}, pbase, dbase, 0);
#pragma unroll
            for (int ks = 0; ks < 8; ks += 2) {
                fetch(S1{}, pbase, dbase, ks + 1);
                compute(S0{}, Y8{});
                if (ks + 2 < 8) {
                    fetch(S0{}, pbase, dbase, ks + 2);
                    compute(S1{}, Y8{});
                } else {
                    compute(S1{}, S0{});                   // last k-step: everything has landed, the barrier may free the buffer
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            buf ^= 1;
        }
    }
    // slab[split][g*32 + row][c]: D[row][c], lane = column c, registers = rows
    const int l31 = lane & 31;
    float* out = a.slab + (size_t)split * (14 * 32 * 64);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        if (i >= ng) continue;
        const int g = wave + 8 * i;
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
                out[(size_t)(g * 32 + row) * 64 + c * 32 + l31] = acc[i][c][r];
            }
    }
#endif
}

// sum the slabs in a fixed order; (g, row) -> filter row g >> 1, column 4*(g & 1) + (row >> 3) (column 7 is the dummy), plane row & 7;
// dst index = c * sc + plane * sp + kh' * 7 + kw' with (kh', kw') flipped for the heads
__global__ __launch_bounds__(256) void smallk_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int splits, int planes,
                                                            int sc, int sp, int flip) {
    // up to 512 slabs of 28 672 floats: one thread per element summed them one after the other (141 us per call, 12x what the bytes
    // cost).  Now a workgroup owns 32 elements: thread (g, e) sums the g-th eighth of the slabs for element e, eight loads in flight,
    // and the eight partial sums are added in order through LDS -- a fixed order, independent of the launch.
    __shared__ float part[8][32];
    const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + e;
    const int per = (splits + 7) / 8;
    const int z0 = g * per, z1 = min(splits, z0 + per);
    float s = 0.f;
    if (idx < 14 * 32 * 64) {
        const float* p = slab + idx;
        int z = z0;
        for (; z + 8 <= z1; z += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(z + u) * (14 * 32 * 64)];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; z < z1; ++z) s += p[(size_t)z * (14 * 32 * 64)];
    }
    part[g][e] = s;
    __syncthreads();
    if (g != 0 || idx >= 14 * 32 * 64) return;
#pragma unroll
    for (int q = 1; q < 8; ++q) s += part[q][e];
    const int c = idx & 63, k = idx >> 6;
    const int gg = k >> 5, row = k & 31;
    const int kh = gg >> 1, kw = 4 * (gg & 1) + (row >> 3), pl = row & 7;
    if (kw >= 7 || pl >= planes) return;
    const int khd = flip ? 6 - kh : kh, kwd = flip ? 6 - kw : kw;
    dw[(size_t)c * sc + (size_t)pl * sp + khd * 7 + kwd] = s;
}

void smallk_plan(int B, int GH, int GW, int* splits, int* ups, int* units_x, int* upi) {
    *units_x = (GW + 15) / 16;
    *upi = ((GH + 7) / 8) * *units_x;
    const int units = B * *upi;
    int s = 512;                                        // two workgroups per CU
    if (s > units / 4) s = units / 4 > 0 ? units / 4 : 1;
    *ups = (units + s - 1) / s;
    *splits = (units + *ups - 1) / *ups;
}

bool narrow_ok(int B, int IH, int IW, int Cin, int OH, int OWg, int KH, int KWW) {
    return B > 0 && Cin == NCH && KH == 7 && KWW == 10 && OH > 0 && OWg > 0 && IH >= KH && IW >= KWW &&
           (size_t)B * IH * IW * NCH * 2 < 0x80000000ull;
}

}  // namespace

extern "C" {

int dwc_bf16_conv2d_narrow_ok(int B, int IH, int IW, int Cin, int OH, int OWg, int KH, int KWW) {
    return narrow_ok(B, IH, IW, Cin, OH, OWg, KH, KWW) ? 1 : 0;
}

/* y[B][OH][OWg][32] (bf16; 4 pixels x 8 planes per group) = act(sum over the KH x KWW wide taps and 64 channels + bias32) with
 * the input window of (oy, gx) starting at (oy + off_h, 4*gx + off_w) of x[B][IH][IW][64].  w_frag = the [32][KH][KWW][64]
 * wide bank in MFMA-fragment order [tap][q][hi][row][8] (tap = kh*KWW + u, q = 16-channel step, hi = 8-channel half), the tap
 * count rounded up to a multiple of 8 with ZERO taps (r04): dwc_bf16_weight_prepare_fwd's [32][Kp] layout permuted + padded.  reflect != 0: reflect rule (forward heads), else zero rule (image
 * gradient on the padded grid). */
int dwc_bf16_conv2d_narrow(const void* x, const void* w_frag, const float* bias32, void* y, int B, int IH, int IW, int Cin, int OH,
                           int OWg, int KH, int KWW, int off_h, int off_w, int act, int reflect, void* stream) {
    if (!x || !w_frag || !y || !narrow_ok(B, IH, IW, Cin, OH, OWg, KH, KWW)) return DWC_EINVAL;
    NarrowArgs a;
    a.x = (const bf16*)x; a.w = (const bf16*)w_frag; a.bias = bias32; a.y = (bf16*)y;
    a.B = B; a.IH = IH; a.IW = IW; a.OH = OH; a.OWg = OWg;
    a.off_h = off_h; a.off_w = off_w; a.act = act; a.reflect = reflect;
    a.blocks_x = (OWg + NB_GROUPS - 1) / NB_GROUPS; a.blocks_y = (OH + 7) / 8;      // (8-row blocks; 16-row blocks measured equal, not instantiated)
    const int nblocks = a.blocks_x * a.blocks_y * B;
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
    }
    const char* e = getenv("DWC_NARROW_PERSIST");       // (read per call: tests / A-B runs switch it between launches)
    const int persist = e ? atoi(e) : 1;
    // (r06) one persistent workgroup per CU where the block columns of an XCD's images fill the rounds of its workgroups to >= 85 %
    // (128 x 128: B = 64, 128, 192, 256, 384 ... fill them exactly); other launches keep the block-per-workgroup form
    const int ncols = ((B + 7) / 8) * a.blocks_x, per_x = cus / 8;
    const int rounds = per_x > 0 ? (ncols + per_x - 1) / per_x : 0;
    if (persist && per_x > 0 && ncols >= per_x && 100 * ncols >= 85 * rounds * per_x)
        hipLaunchKernelGGL((conv_narrow_persist_kernel<7, 10>), dim3(cus & ~7), dim3(512), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL((conv_narrow_kernel<7, 10, 8>), dim3(nblocks), dim3(512), 0, (hipStream_t)stream, a);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

int dwc_bf16_conv2d_stem_ok(int B, int IH, int IW, int OH, int OW, int K, int act) { return stem_ok(B, IH, IW, OH, OW, K, act) ? 1 : 0; }

/* y[B][OH][OW][64] = act(conv7x7(x[B][IH][IW][8 planes]) + bias), window of output (oy, ox) starting at (oy + off, ox + off);
 * reflect != 0: reflect rule (stems forward, off = -3), else zero rule (data gradient of the image heads on the padded grid,
 * off = -6).  act: none / relu / lrelu.  w_steps: [25][64][16] bf16, element (k-step j, channel co, tap 2j+h, plane p) at
 * ((h ^ ((co>>3)&1))*8 + p), taps beyond 48 zero (built by the caller from the OIHW filter). */
int dwc_bf16_conv2d_stem_crop(const void* x, const void* w_steps, const float* bias, void* y, void* inner, int crop, int B, int IH, int IW,
                              int OH, int OW, int K, int off, int act, int reflect, void* stream);
int dwc_bf16_conv2d_stem(const void* x, const void* w_steps, const float* bias, void* y, int B, int IH, int IW, int OH, int OW, int K,
                         int off, int act, int reflect, void* stream) {
    return dwc_bf16_conv2d_stem_crop(x, w_steps, bias, y, nullptr, 0, B, IH, IW, OH, OW, K, off, act, reflect, stream);
}

/* The same with the interior of the output grid diverted: output pixel (oy, ox) with (oy - crop, ox - crop) inside
 * (OH - 2 crop) x (OW - 2 crop) is written to `inner` ([B][OH-2crop][OW-2crop][64]) instead of y -- the data gradient of the image
 * heads on the padded grid leaves only its border ring in y, for dwc_bf16_reflect_pad_adjoint_band to fold onto `inner`. */
int dwc_bf16_conv2d_stem_crop(const void* x, const void* w_steps, const float* bias, void* y, void* inner, int crop, int B, int IH, int IW,
                              int OH, int OW, int K, int off, int act, int reflect, void* stream) {
    if (!x || !w_steps || !y || !stem_ok(B, IH, IW, OH, OW, K, act) || crop < 0 || (crop > 0 && (!inner || OH <= 2 * crop || OW <= 2 * crop)))
        return DWC_EINVAL;
    StemArgs a;
    a.x = (const bf16*)x; a.w = (const bf16*)w_steps; a.bias = bias; a.y = (bf16*)y;
    a.crop = crop; a.CH = OH - 2 * crop; a.CW = OW - 2 * crop; a.inner = (bf16*)inner;
    a.B = B; a.IH = IH; a.IW = IW; a.OH = OH; a.OW = OW; a.off = off; a.act = act; a.reflect = reflect;
    a.blocks_x = (OW + 15) / 16; a.blocks_y = (OH + 15) / 16; a.nblocks = a.blocks_x * a.blocks_y * B;
    const int grid = a.nblocks < 512 ? a.nblocks : 512;              // two persistent workgroups per CU
    hipLaunchKernelGGL(conv_stem_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

size_t dwc_bf16_conv7_smallk_wgrad_ws_bytes(int B, int H, int W, int heads) {
    int splits, ups, ux, upi;
    smallk_plan(B, heads ? H + 6 : H, heads ? W + 6 : W, &splits, &ups, &ux, &upi);
    return (size_t)splits * 14 * 32 * 64 * sizeof(float);
}

/* Weight gradient of the two 7x7 layer shapes between an NHWC8 image (img8: [B][H][W][8] bf16) and a 64-channel tensor (t64:
 * [B][H][W][64] bf16), pad 3, reflect padding in the forward:
 *   heads == 0 (stems, 8 -> 64): img8 = the input image x, t64 = dY;        dw: [64][planes][7][7] fp32
 *   heads != 0 (image heads, 64 -> 8): img8 = dY (pre-activation gradient), t64 = the input x;   dw: [planes][64][7][7] fp32
 * `planes` <= 8 real planes are written.  Scratch: dwc_bf16_conv7_smallk_wgrad_ws_bytes. */
int dwc_bf16_conv7_smallk_wgrad(const void* img8, const void* t64, float* dw, int B, int H, int W, int planes, int heads, void* ws,
                                size_t ws_bytes, void* stream) {
    if (!img8 || !t64 || !dw || B <= 0 || H < 7 || W < 7 || planes < 1 || planes > 8 ||
        (size_t)B * (H + 6) * (W + 6) * 128 >= 0x80000000ull)
        return DWC_EINVAL;
    SmallWgradArgs a;
    int splits;
    a.GH = heads ? H + 6 : H; a.GW = heads ? W + 6 : W;
    smallk_plan(B, a.GH, a.GW, &splits, &a.units_per_split, &a.units_x, &a.units_per_img);
    if (!ws || ws_bytes < (size_t)splits * 14 * 32 * 64 * sizeof(float)) return DWC_EWORKSPACE;
    a.a8 = (const bf16*)img8; a.b64 = (const bf16*)t64; a.slab = (float*)ws;
    a.B = B; a.AH = H; a.AW = W; a.BH = H; a.BW = W;
    a.offA = heads ? -6 : -3; a.offB = heads ? -3 : 0; a.reflA = heads ? 0 : 1; a.reflB = heads ? 1 : 0;
    a.total_units = B * a.units_per_img;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(smallk_wgrad_kernel, dim3(splits), dim3(512), 0, st, a);
    DWC_LAUNCH_CHECK();
    hipLaunchKernelGGL(smallk_reduce_kernel, dim3((14 * 32 * 64 + 31) / 32), dim3(256), 0, st, (const float*)ws, dw, splits, planes,
                       heads ? 49 : planes * 49, heads ? 64 * 49 : 49, heads ? 1 : 0);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

}  // extern "C"
