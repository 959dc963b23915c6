// fp32 7x7 convolutions from 64 channels to the 4 planes of an NHWC4 image as exact split products on the bf16 matrix cores: the
// fused image heads of the decoder (reference networks.py:218-246, forward) and the data gradient of the 7x7 stems w.r.t. their
// input image (reference networks.py:579-585 backward).  The fp32 twin of conv_narrow_bf16.hip's conv_narrow_kernel (r04; until
// then these layers ran on the im2col GEMM at 0.10-0.15 of the bf16 peak: N = 4 planes fills an eighth of a 32-wide tile even in
// the "wide" form -- 8 horizontally adjacent pixels x 4 planes = 32 columns of a [32][KH][KW+7][64] filter bank, see
// ops._prepped 'heads_wide' / 'dgrad_image' -- and every input pixel was re-staged once per tap, 98 times).
//
// A workgroup (8 waves) owns 8 rows x 4 pixel groups (= 32 pixels) of one image = ONE 32-group MFMA tile:
//  * per 16-channel slab the (8+KH-1) x (32+KW+6) input patch is gathered ONCE from the fp32 tensor (reflect or zero rule),
//    split in registers into three bf16 planes (v = p0 + p1 + p2 exactly, conv_halo_x3.hip) and written to LDS as
//    two half planes [pixel][8 ch] each (16-byte slots; one spare slot per 8 pixels and a row pitch of 4 mod 16 slots make the 16
//    lanes a fragment read serves at a time -- 4 rows x 4 groups -- hit 16 different bank windows);
//  * the KH*(KW+7) taps are dealt round-robin to the 8 waves; a wave multiplies its taps against the group tile as the six
//    leading cross products of the planes (D[32 columns][32 groups] += W_tap[32][16] . X_tap[16][32 groups]; the leading product
//    and the five corrections in separate accumulators), its weight fragments -- pre-split planes in fragment order -- read
//    straight from global memory / L2 one tap ahead: no wave shares a tap, so weights need no LDS and the tap loop no barrier;
//  * the next slab's patch is in flight (registers) during the taps of the current one; 70 KB of LDS and <= 128 registers: two
//    workgroups per CU;
//  * the 8 partial sums meet in LDS (fixed order), bias + activation, 16-byte stores of one pixel's 4 planes.
// Output [B][OH][OWg][32] fp32 (= the NHWC4 image when OWg*8 is its width); the input window of output (oy, group gx) starts at
// (oy + off_h, 8*gx + off_w): forward off = -pad with the reflect rule; image gradient off = -(K-1) with the zero rule on the
// padded grid (OH = H + 2 pad), folded back by dwc_reflect_pad_adjoint_pitch.
#include "conv_geom.h"

namespace {

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int NX_GROUPS = 4, NX_CH = 64;
constexpr int SMALLK_X3_SLOTS = 256;      // workgroups of smallk_wgrad_x3_kernel resident at once (61 KB of LDS, see its registers)

// fp32 x4 -> three planes of 4 bf16 (packed two per dword), exact: v = p0 + p1 + p2 (see conv_halo_x3.hip)
__device__ __forceinline__ void nx_split3(f32x4 v, u32x2& p0, u32x2& p1, u32x2& p2) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const f32x2 x = {v[2 * k], v[2 * k + 1]};
        const f32x2 hb = {__uint_as_float(__float_as_uint(x[0]) & 0xffff0000u), __uint_as_float(__float_as_uint(x[1]) & 0xffff0000u)};
        const f32x2 r = x - hb;
        const f32x2 mb = {__uint_as_float(__float_as_uint(r[0]) & 0xffff0000u), __uint_as_float(__float_as_uint(r[1]) & 0xffff0000u)};
        const f32x2 l = r - mb;
        p0[k] = __builtin_amdgcn_perm(__float_as_uint(x[1]), __float_as_uint(x[0]), 0x07060302u);
        p1[k] = __builtin_amdgcn_perm(__float_as_uint(r[1]), __float_as_uint(r[0]), 0x07060302u);
        p2[k] = __builtin_amdgcn_perm(__float_as_uint(l[1]), __float_as_uint(l[0]), 0x07060302u);
    }
}

struct NarrowX3Args {
    const float* x;      // [B][IH][IW][64] fp32
    const bf16* w;       // [slab 4][tap, padded to a multiple of 8 with zeros][plane 3][lane 64][8]: MFMA-fragment order (lane =
                         // channel half * 32 + column) of the split bank
    const float* bias;   // [32] or null
    float* y;            // [B][OH][OWg][32]
    int B, IH, IW, OH, OWg, off_h, off_w, act, reflect;
    int blocks_x, blocks_y;
};

// MT: 32-group tiles per workgroup (8 rows each).  1: 70 KB of LDS, <= 128 registers (PF = 1), two workgroups per CU.  2: 16 rows,
// 110 KB, one workgroup per CU -- every weight fragment (1 KB per wave, tap, slab and plane, streamed from L2: 1.2 MB per block)
// feeds two tiles.  Measured equal within 2 % once the weight loads are really in flight ahead of their use (see load_item).
template <int KH, int KWW, int MT, int PF>      // PF: weight fragment sets in flight ahead of the one being multiplied (1 or 2)
__global__ __launch_bounds__(512, MT == 1 && PF == 1 ? 2 : 1) void conv_narrow_x3_kernel(NarrowX3Args a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NX_ROWS = 8 * MT, NX_CS = 16;
    constexpr int PR = NX_ROWS + KH - 1, PC = 8 * NX_GROUPS + KWW - 1, PPIX = PR * PC;
    constexpr int PPASS = (PPIX + 127) / 128;            // gather passes: 128 pixels x 4 channel quads per pass
    constexpr int NTAP = KH * KWW;
    constexpr int NSLAB = NX_CH / NX_CS;
    // A plane is TWO half planes (channels 0-7 / 8-15 of the slab), each [pixel slot][8 ch] = 16-byte slots: pixel (pr, pc) in slot
    // pr*PITCH_S + pc + (pc >> 3), PITCH_S = 4 mod 16.  A fragment read (ds_read_b128) is served 16 lanes at a time -- 4 rows
    // {0,3,5,6} / {1,2,4,7} x 4 groups, one channel half -- whose slots are then 4*row + 9*group mod 16: all different, no bank
    // conflict (with whole 32-byte pixel slots the 16 lanes of a pass can only reach 8 different 16-byte bank windows: 2-way).
    constexpr int ROW_S = PC + ((PC - 1) >> 3) + 1;
    constexpr int PITCH_S = (ROW_S + 11) / 16 * 16 + 4;
    static_assert(PITCH_S >= ROW_S && PITCH_S % 16 == 4, "row pitch");
    constexpr int HPLANE = PR * PITCH_S * 8;              // elements of a half plane
    constexpr int P_PLANE = 2 * HPLANE;
    constexpr int RED = 8 * MT * 16 * 64;                // floats of the reduction: [wave][tile][reg][lane]
    constexpr int SMEM_B = 3 * P_PLANE * 2 > RED * 4 ? 3 * P_PLANE * 2 : RED * 4;
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
    const int oy0 = by * NX_ROWS, gx0 = bx * NX_GROUPS;

    // ---- patch gather map: thread = (patch pixel t>>2 [+128 per pass], channel quad t&3 of the slab) -------------------------
    const unsigned x_bytes = (unsigned)((size_t)a.B * a.IH * a.IW * NX_CH * 4);
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, x_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    unsigned p_off[PPASS];                               // byte offset of the pixel's channel quad in slab 0 (OOB: zero / past the patch)
    int p_dst[PPASS];
#pragma unroll
    for (int i = 0; i < PPASS; ++i) {
        const int pp = (t >> 2) + 128 * i;
        const int pr = pp / PC, pc = pp - pr * PC;
        int h = oy0 + pr + a.off_h, w = 8 * gx0 + pc + a.off_w;
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
        p_off[i] = ok ? ((unsigned)((n * a.IH + h) * a.IW + w) * NX_CH + (unsigned)((t & 3) * 4)) * 4u : OOB;
        p_dst[i] = pp < PPIX ? (pr * PITCH_S + pc + (pc >> 3)) * 8 + ((t & 3) >> 1) * HPLANE + (t & 1) * 4 : -1;
    }
    f32x4 pv[PPASS];
    auto load_patch = [&](int q) {
#pragma unroll
        for (int i = 0; i < PPASS; ++i)
            pv[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, p_off[i], (unsigned)(q * NX_CS * 4), 0));
    };
    auto write_patch = [&]() {
#pragma unroll
        for (int i = 0; i < PPASS; ++i) {
            if (p_dst[i] < 0) continue;
            u32x2 p0, p1, p2;
            nx_split3(pv[i], p0, p1, p2);
            *reinterpret_cast<u32x2*>(sP + p_dst[i]) = p0;
            *reinterpret_cast<u32x2*>(sP + P_PLANE + p_dst[i]) = p1;
            *reinterpret_cast<u32x2*>(sP + 2 * P_PLANE + p_dst[i]) = p2;
        }
    };

    // ---- this wave's taps -------------------------------------------------------------------------------------------------------
    f32x16 acc[MT], lo[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][r] = 0.f, lo[m][r] = 0.f;
    // group of this lane in tile m: g = 32 m + l31 -> (row g >> 2, group-in-row g & 3); patch pixel of tap (kh, u): (row + kh, 8*gl + u)
    const int g_base = ((l31 >> 2) * PITCH_S + (l31 & 3) * (8 + 1)) * 8 + hi * HPLANE;
    const bf16* wlane = a.w + lane * 8;
    // Weight fragments: TWO taps ahead (a tap is 6 MFMAs = 192 cycles of this wave, ~770 with four waves per SIMD: one tap of lead
    // does not cover an L2 round trip), three fragment sets in rotation by NAME: the tap list is padded to NT_W = 13 taps per wave
    // (taps >= KH*KWW carry zero weights) and the 4 x 13 steps are unrolled, so the set of a step is a compile-time index.  (The
    // first version rotated by copying; the copy of the set loaded in the SAME iteration drained vmcnt to 0 at every tap.)
    constexpr int NT_W = (NTAP + 7) / 8, NTAP_P = 8 * NT_W, N_ITEMS = NSLAB * NT_W;
    bf16x8 wset[PF + 1][3];
    // The weight loads are asm statements with hand-counted waits tied to the fragment registers (wait_item): as plain loads hipcc
    // sinks every one of them to just in front of its first use (the kernel is at 124 registers), and a step then waits out a whole
    // L2 round trip -- 52 of them in a row per block, ~49 us per pair of resident blocks whatever the tile size or the prefetch
    // distance written in the source.  Invisible to the compiler's own vmcnt bookkeeping (the patch gather), they can only make ITS
    // waits longer, never shorter.
    auto load_item = [&](int j, bf16x8 (&dst)[3]) {     // item j = (slab j / NT_W, tap wave + 8 * (j % NT_W))
        const int qq = j / NT_W;
        const bf16* src = wlane + ((size_t)(qq * NTAP_P + wave + 8 * (j - qq * NT_W)) * 3) * 512;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst[0]) : "v"(src));
        asm volatile("global_load_dwordx4 %0, %1, off offset:1024" : "=v"(dst[1]) : "v"(src));
        asm volatile("global_load_dwordx4 %0, %1, off offset:2048" : "=v"(dst[2]) : "v"(src));
    };
    // vm operations issued AFTER the loads of the item about to be used: the two items ahead (3 loads each) and, for the first two
    // steps of a slab, the PPASS gather loads of the next slab's patch issued at the slab's top
    auto wait_item = [&](bf16x8 (&cur)[3], int younger) {
#define DWC_NX_WAIT(N) asm volatile("s_waitcnt vmcnt(" #N ")" : "+v"(cur[0]), "+v"(cur[1]), "+v"(cur[2]))
        static_assert(PPASS <= 10, "vmcnt immediates below");
        switch (younger) {
            case 0: DWC_NX_WAIT(0); break;
            case 3: DWC_NX_WAIT(3); break;
            case 6: DWC_NX_WAIT(6); break;
            case 3 + PPASS: if (PPASS == 5) DWC_NX_WAIT(8); else DWC_NX_WAIT(3); break;
            case 6 + PPASS: if (PPASS == 5) DWC_NX_WAIT(11); else DWC_NX_WAIT(6); break;
            default: DWC_NX_WAIT(0); break;
        }
#undef DWC_NX_WAIT
    };
    load_patch(0);
    load_item(0, wset[0]);
    if (PF == 2) load_item(1, wset[PF == 2 ? 1 : 0]);
    write_patch();
#pragma unroll
    for (int q = 0; q < NSLAB; ++q) {
        if (q + 1 < NSLAB) load_patch(q + 1);           // in flight during this slab's taps
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads();                                // this slab's patch is in LDS (every wave's share)
#pragma unroll
        for (int j = 0; j < NT_W; ++j) {
            constexpr int dummy = 0;
            (void)dummy;
            const int item = q * NT_W + j;
            if (item + PF < N_ITEMS) load_item(item + PF, wset[(item + PF) % (PF + 1)]);
            const int tp = wave + 8 * j;
            int kh = tp / KWW, u = tp - kh * KWW;
            if (tp >= NTAP) kh = 0, u = 0;              // padding tap: zero weights, any patch pixel
            const bf16* p = sP + g_base + (kh * PITCH_S + u + (u >> 3)) * 8;
            bf16x8 fa[MT][3];
#pragma unroll
            for (int m = 0; m < MT; ++m)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) fa[m][pl] = *reinterpret_cast<const bf16x8*>(p + m * 8 * PITCH_S * 8 + pl * P_PLANE);
            // D[column][group]: planes (weight, patch); the leading product apart from the five corrections (conv_halo_x3.hip)
            bf16x8 (&cur)[3] = wset[item % (PF + 1)];
            wait_item(cur, (item + 1 < N_ITEMS ? 3 : 0) + (PF == 2 && item + 2 < N_ITEMS ? 3 : 0) + ((j < PF && q + 1 < NSLAB) ? PPASS : 0));
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[0], fa[m][0], acc[m], 0, 0, 0);
                lo[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[0], fa[m][1], lo[m], 0, 0, 0);
                lo[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[1], fa[m][0], lo[m], 0, 0, 0);
                lo[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[0], fa[m][2], lo[m], 0, 0, 0);
                lo[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[1], fa[m][1], lo[m], 0, 0, 0);
                lo[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(cur[2], fa[m][0], lo[m], 0, 0, 0);
            }
        }
        __syncthreads();                                // every wave is past its last read of this slab's patch
        if (q + 1 < NSLAB) write_patch();
    }

    // ---- the 8 partial sums meet in LDS (the patch is dead: barrier above) ----------------------------------------------------
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int r = 0; r < 16; ++r) sR[((wave * MT + m) * 16 + r) * 64 + lane] = acc[m][r] + lo[m][r];
    __syncthreads();
    // epilogue thread = (tile, register block rb, lane): registers 4*rb + k of lane (l31, hi) = columns 8*rb + 4*hi + k
    // = pixel 2*rb + hi of the group, plane k
#pragma unroll
    for (int pass = 0; pass < (MT + 1) / 2; ++pass) {
        const int m = 2 * pass + (t >> 8), rb = (t >> 6) & 3;
        if (m >= MT) break;
        f32x4 v;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float s = 0.f;
#pragma unroll
            for (int w8 = 0; w8 < 8; ++w8) s += sR[((w8 * MT + m) * 16 + 4 * rb + k) * 64 + lane];      // fixed order
            v[k] = s;
        }
        const int oy = oy0 + 8 * m + (l31 >> 2), gx = gx0 + (l31 & 3);
        if (oy < a.OH && gx < a.OWg) {
            const int c0 = 8 * rb + 4 * hi;
            if (a.bias) v += *reinterpret_cast<const f32x4*>(a.bias + c0);
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = dwc_act_apply(v[k], a.act, c0 + k);
            *reinterpret_cast<f32x4*>(a.y + ((size_t)(n * a.OH + oy) * a.OWg + gx) * 32 + c0) = v;
        }
    }
#endif
}

// ------------------------------------------------------------------------------------------
// The opposite shape, fp32: 7x7 convolutions from the 4 planes of an NHWC4 image to 64 channels as split products -- the stems
// (forward, reference networks.py:163-166 / :60-66) and the data gradient of the image heads.  The fp32 twin of conv_stem_kernel
// (conv_narrow_bf16.hip).  K = 49 taps x 4 planes: one MFMA k-step is FOUR taps (lane half hi takes taps 4j+2hi, 4j+2hi+1: two
// 8-byte pixels of a patch plane), 13 steps.  The whole filter as three bf16 planes (3 x 13 x 64 x 32 bytes = 78 KB, halves of a
// row swapped by (row>>3)&1 as in conv_halo_x3.hip) stays in LDS while the workgroup walks over 16x16-pixel blocks (persistent);
// the 22x22 patch is gathered from the fp32 image (one pixel per thread), split in registers and stored as three planes of 8-byte
// pixels.  8 waves, one 32-pixel tile x both 32-channel tiles each; six products per tile pair, leading product and corrections
// in separate accumulators; fp32 results stored straight from registers (a lane owns a pixel and 4 consecutive channels).
// ------------------------------------------------------------------------------------------
struct StemX3Args {
    const float* x;      // [B][IH][IW][4]
    const bf16* w;       // [3 planes][13][64][16]: k-step j, channel co, (tap 4j + 2h + t, plane p) at ((h ^ ((co>>3)&1))*8 + 4t + p)
    const float* bias;   // [64] or null
    float* y;            // [B][OH][OW][64]
    int B, IH, IW, OH, OW, off, act, reflect;
    int blocks_x, blocks_y, nblocks;
    // crop > 0 (data gradient of the image heads: y is the gradient of the PADDED tensor): an output pixel whose cropped coordinate
    // lies inside CH x CW goes straight to `inner` ([B][CH][CW][64]), only the border ring to y; a band fold follows
    int crop, CH, CW;
    float* inner;
    unsigned long long* ys = nullptr;      // optional absmax slot of y (two-plane consumers, dwc_common.h), raised from the store pass
    unsigned ys_epoch = 0;
};

__global__ __launch_bounds__(512, 1) void conv_stem_x3_kernel(StemX3Args a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int KS = 7, PW = 16 + KS - 1, PPIX = PW * PW, NKS = 13;
    constexpr int W_PLANE = NKS * 64 * 16;              // elements of one weight plane
    constexpr int P_PLANE = 512 * 4;                    // one patch plane: 484 pixels x 4 planes (one pixel per thread)
    __shared__ __attribute__((aligned(16))) bf16 smem[3 * W_PLANE + 3 * P_PLANE];
    bf16* sW = smem;
    bf16* sP = smem + 3 * W_PLANE;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    unsigned y_am = 0;                                  // largest |y| this thread stores
    // ---- the filter, once -------------------------------------------------------------------------------------------------
    {
        const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.w), 0, 3 * W_PLANE * 2u, 0x00020000);
#pragma unroll
        for (int i = 0; i < (3 * W_PLANE * 2 + 8191) / 8192; ++i) {   // 512 lanes x 16 bytes per instruction
            const unsigned off = (unsigned)(i * 8192 + t * 16);
            if (off < 3 * W_PLANE * 2u)                 // (masked lanes write nothing: the tail must not spill zeros over the patch)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(sW + i * 4096 + wave * 512),
                                                         16, off, 0, 0, 0);
        }
    }
    const unsigned x_bytes = (unsigned)((size_t)a.B * a.IH * a.IW * 16u);
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, x_bytes, 0x00020000);
    auto load_patch = [&](int blk) -> f32x4 {           // this thread's patch pixel of block blk (zero past the patch / image)
        int bid = blk;
        const int bx = bid % a.blocks_x;
        bid /= a.blocks_x;
        const int by = bid % a.blocks_y, n = bid / a.blocks_y;
        const int pr = t / PW, pc = t - pr * PW;
        int h = by * 16 + pr + a.off, w = bx * 16 + pc + a.off;
        bool ok = t < PPIX;
        if (a.reflect) {
            h = reflect_idx(h, a.IH);
            w = reflect_idx(w, a.IW);
        } else {
            ok = ok && (unsigned)h < (unsigned)a.IH && (unsigned)w < (unsigned)a.IW;
        }
        h = min(max(h, 0), a.IH - 1);
        w = min(max(w, 0), a.IW - 1);
        const unsigned off = (unsigned)((n * a.IH + h) * a.IW + w) * 16u;
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, ok ? off : 0x80000000u, 0, 0));
    };
    // fragment addressing: this wave's pixel tile: block pixel pb = wave*32 + l31 -> patch pixel (pb>>4)*PW + (pb&15)
    const int pb = wave * 32 + l31;
    const int pp0 = (pb >> 4) * PW + (pb & 15);
    const int b_off = l31 * 16 + ((hi ^ ((l31 >> 3) & 1)) * 8);
    const float slope = dwc_act_slope(a.act);
    f32x4 bvs[2][4];                                    // bias vectors of this lane's columns, loaded once for all blocks
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
            bvs[c][q4] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + c * 32 + 8 * q4 + 4 * hi) : f32x4{0.f, 0.f, 0.f, 0.f};

    f32x4 pv = load_patch(blockIdx.x < (unsigned)a.nblocks ? blockIdx.x : 0);
    for (int blk = blockIdx.x; blk < a.nblocks; blk += gridDim.x) {
        int bid = blk;
        const int bx = bid % a.blocks_x;
        bid /= a.blocks_x;
        const int by = bid % a.blocks_y, n = bid / a.blocks_y;
        const int oy0 = by * 16, ox0 = bx * 16;
        __syncthreads();                                              // every wave is past the previous block's reads
        {
            u32x2 p0, p1, p2;
            nx_split3(pv, p0, p1, p2);
            *reinterpret_cast<u32x2*>(sP + t * 4) = p0;
            *reinterpret_cast<u32x2*>(sP + P_PLANE + t * 4) = p1;
            *reinterpret_cast<u32x2*>(sP + 2 * P_PLANE + t * 4) = p2;
        }
        if (blk + (int)gridDim.x < a.nblocks) pv = load_patch(blk + gridDim.x);      // in flight during this block's products
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (blk == (int)blockIdx.x) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the filter (first block only; also drains pv)
        __syncthreads();

        f32x16 acc[2], lo[2];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][r] = 0.f, lo[c][r] = 0.f;
#pragma unroll
        for (int j = 0; j < NKS; ++j) {
            const int ta = min(4 * j + 2 * hi, KS * KS - 1), tb = min(4 * j + 2 * hi + 1, KS * KS - 1);   // (taps >= 49: zero weights)
            const int da = (ta / KS) * PW + ta % KS, db = (tb / KS) * PW + tb % KS;
            bf16x8 fa[3], fb[3][2];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                const u32x2 va = *reinterpret_cast<const u32x2*>(sP + pl * P_PLANE + (pp0 + da) * 4);
                const u32x2 vb = *reinterpret_cast<const u32x2*>(sP + pl * P_PLANE + (pp0 + db) * 4);
                fa[pl] = __builtin_bit_cast(bf16x8, u32x4{va[0], va[1], vb[0], vb[1]});
#pragma unroll
                for (int c = 0; c < 2; ++c) fb[pl][c] = *reinterpret_cast<const bf16x8*>(sW + pl * W_PLANE + (j * 64 + c * 32) * 16 + b_off);
            }
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0][c], fa[0], acc[c], 0, 0, 0);
                lo[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0][c], fa[1], lo[c], 0, 0, 0);
                lo[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[1][c], fa[0], lo[c], 0, 0, 0);
                lo[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[0][c], fa[2], lo[c], 0, 0, 0);
                lo[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[1][c], fa[1], lo[c], 0, 0, 0);
                lo[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[2][c], fa[0], lo[c], 0, 0, 0);
            }
        }
        // ---- epilogue: D[channel][pixel]: lane = pixel, 4 consecutive channels per register quad: 16-byte fp32 stores ----------
        const int yy = oy0 + (pb >> 4), xx = ox0 + (pb & 15);
        if (yy < a.OH && xx < a.OW) {
            float* d = a.y + ((size_t)(n * a.OH + yy) * a.OW + xx) * 64;
            if (a.crop) {
                const int cy = yy - a.crop, cx = xx - a.crop;
                if ((unsigned)cy < (unsigned)a.CH && (unsigned)cx < (unsigned)a.CW) d = a.inner + ((size_t)(n * a.CH + cy) * a.CW + cx) * 64;
            }
#pragma unroll
            for (int c = 0; c < 2; ++c)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    f32x4 v = {acc[c][4 * q4] + lo[c][4 * q4], acc[c][4 * q4 + 1] + lo[c][4 * q4 + 1], acc[c][4 * q4 + 2] + lo[c][4 * q4 + 2],
                               acc[c][4 * q4 + 3] + lo[c][4 * q4 + 3]};
                    v += bvs[c][q4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[k] = dwc_act_simple(v[k], slope);
                    y_am = max(max(y_am, max(dwc_abs_bits(v[0]), dwc_abs_bits(v[1]))), max(dwc_abs_bits(v[2]), dwc_abs_bits(v[3])));
                    *reinterpret_cast<f32x4*>(d + c * 32 + 8 * q4 + 4 * hi) = v;
                }
        }
    }
    __shared__ unsigned s_am[8];
    dwc_amax_block_publish(a.ys, a.ys_epoch, y_am, s_am);      // (a.ys is uniform over the launch)
#endif
}

// out[p][i] = plane p of the exact three-way bf16 split of (idx[i] < 0 ? 0 : src[idx[i]]): the prepared filter banks of the two
// kernels above are a fixed permutation (+ zero padding) of the OIHW filter -- one launch per optimiser step instead of a dozen
// torch ops (hipdwc.ops._prepped builds the index table once per layout)
__global__ __launch_bounds__(256) void x3_gather_split_kernel(const float* __restrict__ src, const int* __restrict__ idx,
                                                              unsigned short* __restrict__ out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int k = idx[i];
    const float v = k < 0 ? 0.f : src[k];
    const unsigned hb = __float_as_uint(v) & 0xffff0000u;
    const float r1 = v - __uint_as_float(hb);
    const unsigned mb = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mb);
    out[i] = (unsigned short)(hb >> 16);
    out[(size_t)n + i] = (unsigned short)(mb >> 16);
    out[2 * (size_t)n + i] = (unsigned short)(__float_as_uint(r2) >> 16);
}

bool stem_x3_ok(int B, int IH, int IW, int OH, int OW, int K, int act) {
    return B > 0 && K == 7 && IH >= 7 && IW >= 7 && OH > 0 && OW > 0 && act <= DWC_ACT_LRELU &&
           (size_t)B * IH * IW * 16 < 0x80000000ull;
}

bool narrow_x3_ok(int B, int IH, int IW, int Cin, int OH, int OWg, int KH, int KWW) {
    return B > 0 && IH > 0 && IW > 0 && Cin == NX_CH && OH > 0 && OWg > 0 && KH == 7 && KWW == 14 &&
           (size_t)B * IH * IW * NX_CH * 4 < 0x7fffffffull;
}


// ------------------------------------------------------------------------------------------
// Weight gradients of both 7x7 shapes in fp32 (r05; the fp32 twin of smallk_wgrad_kernel, conv_narrow_bf16.hip -- until then these
// ran on conv_wgrad_kernel<X3> at 0.14 of the peak, 1.05 ms of a c1 step): a correlation between a 4-plane fp32 image A and a
// 64-channel fp32 tensor Bt over the pixels q of a grid, for the 49 offsets:
//     C[tap][plane][c] = sum_q A[q + tap + offA][plane] * Bt[q + offB][c]
//   stems:  A = x image (reflect rule, offA = -3), Bt = dY, grid = H x W               -> dW[c][plane][kh][kw] = C
//   heads:  A = gradient image g (zero rule, offA = -6), Bt = x (reflect, offB = -3), grid = (H+6) x (W+6) padded positions
//           -> dW[plane][c][6-kh][6-kw] = C
// as exact split products (three bf16 planes per operand, six of the nine cross products, leading product and corrections in
// separate accumulators -- conv_halo_x3.hip).  An MFMA row tile is ONE filter row: 8 horizontally adjacent taps (the 8th a dummy) x
// 4 planes = the 32 contiguous bf16 of 8 patch pixels in a pixel-major plane image, fetched with the transposing LDS read at a
// per-lane pixel offset; the 7 filter rows go to waves 0..6 (all 8 waves stage).  Contraction over 8x16-pixel units: the fp32 A
// patch (14 x 24 pixels) and Bt block (128 pixels x 64 channels) of the NEXT unit are loaded into registers during the MFMAs of the
// current one, split and written to the single LDS buffer between two barriers.  fp32 slabs per pixel split, summed in a fixed
// order by smallk_x3_reduce_kernel.
// ------------------------------------------------------------------------------------------
struct SmallWgradX3Args {
    const float* a4;     // [B][AH][AW][4]
    const float* b64;    // [B][BH][BW][64]
    float* slab;         // [splits][7*32][64]
    int B, AH, AW, BH, BW, GH, GW, offA, offB, reflA, reflB;
    int units_x, units_per_img, total_units, units_per_split;
};

__global__ __launch_bounds__(512) void smallk_wgrad_x3_kernel(SmallWgradX3Args a) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int PCA = 24, PRA = 14;                   // A patch: (8+6) rows x (16+7 -> 24) pixels
    constexpr int A_PL = 512 * 4, B_PL = 128 * 64;      // elements of one plane: A [pixel slot][4 planes], Bt [pixel][64 channels]
    __shared__ __attribute__((aligned(16))) bf16 smem[3 * (A_PL + B_PL)];
    bf16* sA = smem;                                    // [plane of the split][pixel][4]
    bf16* sB = smem + 3 * A_PL;                         // [plane of the split][pixel][64], 16-byte chunks swizzled by (pixel >> 1) & 1
    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int split = blockIdx.x;
    const int u0 = split * a.units_per_split, u1 = min(a.total_units, u0 + a.units_per_split);

    // ---- staging: fp32 global -> registers (next unit, during the MFMAs) -> three bf16 planes in LDS -------------------------
    f32x4 ra, rb[4];
    auto load_unit = [&](int u) {
        const int n = u / a.units_per_img, ur = u - n * a.units_per_img;
        const int uy = ur / a.units_x, ux = ur - uy * a.units_x;
        const int gy0 = uy * 8, gx0 = ux * 16;
        {   // A patch: thread t = patch pixel t
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
            ra = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ok) ra = *reinterpret_cast<const f32x4*>(a.a4 + ((size_t)(n * a.AH + h) * a.AW + w) * 4);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // Bt block: quad id -> (pixel m, channels 4q..4q+3); pixels outside the grid contribute zeros
            const int id = t + 512 * i;
            const int m = id >> 4, q = id & 15;
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
            rb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (ok) rb[i] = *reinterpret_cast<const f32x4*>(a.b64 + ((size_t)(n * a.BH + h) * a.BW + w) * 64 + q * 4);
        }
    };
    auto write_unit = [&]() {
        {
            u32x2 p0, p1, p2;
            nx_split3(ra, p0, p1, p2);
            *reinterpret_cast<u32x2*>(sA + 0 * A_PL + t * 4) = p0;
            *reinterpret_cast<u32x2*>(sA + 1 * A_PL + t * 4) = p1;
            *reinterpret_cast<u32x2*>(sA + 2 * A_PL + t * 4) = p2;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int id = t + 512 * i;
            const int m = id >> 4, q = id & 15;
            const int el = m * 64 + (((q >> 1) ^ (4 * ((m >> 1) & 1))) << 3) + (q & 1) * 4;
            u32x2 p0, p1, p2;
            nx_split3(rb[i], p0, p1, p2);
            *reinterpret_cast<u32x2*>(sB + 0 * B_PL + el) = p0;
            *reinterpret_cast<u32x2*>(sB + 1 * B_PL + el) = p1;
            *reinterpret_cast<u32x2*>(sB + 2 * B_PL + el) = p2;
        }
    };

    // ---- fragments: transposing reads (ds_read_b64_tr_b16), lane 4q+p of a 16-lane group addresses pixel q, elements 4p..4p+3 ------
    const int kh = wave;                                // waves 0..6: filter row kh; wave 7 only stages
    f32x16 acc[2], lo[2];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f, lo[c][r] = 0.f;
    const int li = lane & 15, gam = (lane >> 4) & 1, hi = lane >> 5;
    const int tq = li >> 2, tp = li & 3;
    const int pxl = 8 * hi + tq;
    // A: row (kw, plane) = kw*4 + plane IS the element offset from the row's first pixel (taps are consecutive pixels, planes
    // consecutive elements): this lane addresses pixel + (4*gam + tp), all four planes
    const int row_el = 16 * gam + 4 * tp;
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) bf16x4* lds4;
    auto tr = [](const bf16* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)p); };
    // fragments of k-step ks into set `st` (two sets: the reads of step ks + 1 are issued in front of the MFMAs of step ks)
    bf16x8 fa[2][3], fb[2][2][3];
    auto fetch = [&](int st, int ks) {
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
            const bf16* pa = sA + pl * A_PL + ((ks + kh) * PCA + pxl) * 4 + row_el;
            fa[st][pl] = __builtin_shufflevector(tr(pa), tr(pa + 4 * 4), 0, 1, 2, 3, 4, 5, 6, 7);
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int col = ct * 32 + 16 * gam + 4 * tp;
                const int m0 = ks * 16 + pxl, m1 = m0 + 4;
                const bf16* b0 = sB + pl * B_PL + m0 * 64 + (((col >> 3) ^ (4 * ((m0 >> 1) & 1))) << 3) + (col & 7);
                const bf16* b1 = sB + pl * B_PL + m1 * 64 + (((col >> 3) ^ (4 * ((m1 >> 1) & 1))) << 3) + (col & 7);
                fb[st][ct][pl] = __builtin_shufflevector(tr(b0), tr(b1), 0, 1, 2, 3, 4, 5, 6, 7);
            }
        }
    };
    auto compute = [&](int st) {
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            lo[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][2], fb[st][ct][0], lo[ct], 0, 0, 0);
            lo[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][1], fb[st][ct][1], lo[ct], 0, 0, 0);
            lo[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][0], fb[st][ct][2], lo[ct], 0, 0, 0);
            lo[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][1], fb[st][ct][0], lo[ct], 0, 0, 0);
            lo[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][0], fb[st][ct][1], lo[ct], 0, 0, 0);
            acc[ct] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[st][0], fb[st][ct][0], acc[ct], 0, 0, 0);
        }
    };

    if (u0 < u1) {
        load_unit(u0);
        write_unit();
        __syncthreads();
        for (int u = u0; u < u1; ++u) {
            const bool next = u + 1 < u1;
            if (next) load_unit(u + 1);                  // in flight during the MFMAs of this unit
            if (kh < 7) {
                fetch(0, 0);
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    if (ks + 1 < 8) fetch((ks + 1) & 1, ks + 1);
                    compute(ks & 1);
                }
            }
            __syncthreads();                             // every wave has read the unit
            if (next) {
                write_unit();
                __syncthreads();
            }
        }
    }
    // slab[split][kh*32 + row][c]: D[row][c], lane = column c, registers = rows
    if (kh < 7) {
        const int l31 = lane & 31;
        float* out = a.slab + (size_t)split * (7 * 32 * 64);
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (r & 3) + 8 * (r >> 2) + 4 * hi;
                out[(size_t)(kh * 32 + row) * 64 + c * 32 + l31] = acc[c][r] + lo[c][r];
            }
    }
#endif
}

// sum the slabs in a fixed order; (kh, row) -> column kw = row >> 2 (column 7 is the dummy), plane row & 3;
// dst index = c * sc + plane * sp + kh' * 7 + kw' with (kh', kw') flipped for the heads  (smallk_reduce_kernel of conv_narrow_bf16.hip)
__global__ __launch_bounds__(256) void smallk_x3_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int splits, int planes,
                                                               int sc, int sp, int flip) {
    constexpr int N = 7 * 32 * 64;
    __shared__ float part[8][32];
    const int e = threadIdx.x & 31, g = threadIdx.x >> 5;
    const int idx = blockIdx.x * 32 + e;
    const int per = (splits + 7) / 8;
    const int z0 = g * per, z1 = min(splits, z0 + per);
    float s = 0.f;
    if (idx < N) {
        const float* p = slab + idx;
        int z = z0;
        for (; z + 8 <= z1; z += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = p[(size_t)(z + u) * N];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; z < z1; ++z) s += p[(size_t)z * N];
    }
    part[g][e] = s;
    __syncthreads();
    if (g != 0 || idx >= N) return;
#pragma unroll
    for (int q = 1; q < 8; ++q) s += part[q][e];
    const int c = idx & 63, k = idx >> 6;
    const int kh = k >> 5, row = k & 31;
    const int kw = row >> 2, pl = row & 3;
    if (kw >= 7 || pl >= planes) return;
    const int khd = flip ? 6 - kh : kh, kwd = flip ? 6 - kw : kw;
    dw[(size_t)c * sc + (size_t)pl * sp + khd * 7 + kwd] = s;
}

void smallk_x3_plan(int B, int GH, int GW, int* splits, int* ups, int* units_x, int* upi) {
    *units_x = (GW + 15) / 16;
    *upi = ((GH + 7) / 8) * *units_x;
    const int units = B * *upi;
    int s = SMALLK_X3_SLOTS;
    if (s > units / 4) s = units / 4 > 0 ? units / 4 : 1;
    *ups = (units + s - 1) / s;
    *splits = (units + *ups - 1) / *ups;
}

}  // namespace

extern "C" {

int dwc_x3_conv2d_narrow_ok(int B, int IH, int IW, int Cin, int OH, int OWg, int KH, int KWW) {
    return narrow_x3_ok(B, IH, IW, Cin, OH, OWg, KH, KWW) ? 1 : 0;
}

/* bf16 elements of the split, fragment-ordered filter bank of dwc_x3_conv2d_narrow: [4 slabs][taps: KH*KWW rounded up to a multiple of
 * 8, the padding taps ZERO][3 planes][64 lanes][8] */
size_t dwc_x3_conv2d_narrow_weight_elems(int KH, int KWW) { return (size_t)4 * ((KH * KWW + 7) / 8 * 8) * 3 * 64 * 8; }

/* y[B][OH][OWg][32] (fp32; 8 pixels x 4 planes per group) = act(sum over the KH x KWW wide taps and 64 channels + bias32) with the
 * input window of (oy, gx) starting at (oy + off_h, 8*gx + off_w) of x[B][IH][IW][64] fp32, as exact three-way bf16 split
 * products.  w_frag: the [32][64][KH][KWW] wide bank (copy p of the real filter shifted right by p taps, see
 * hipdwc.ops._shifted_bank) split into three bf16 planes in MFMA-fragment order [slab q][tap kh*KWW+u, zero taps up to a multiple of 8][plane][lane][8]: lane =
 * half*32 + column holds channels 16q + 8*half .. +7 of bank row `column`.  reflect != 0: reflect rule (forward heads), else the
 * zero rule (image gradient on the padded grid).  KH = 7, KWW = 14 only. */
int dwc_x3_conv2d_narrow(const float* x, const void* w_frag, const float* bias32, float* y, int B, int IH, int IW, int Cin, int OH,
                         int OWg, int KH, int KWW, int off_h, int off_w, int act, int reflect, void* stream) {
    if (!x || !w_frag || !y || !narrow_x3_ok(B, IH, IW, Cin, OH, OWg, KH, KWW)) return DWC_EINVAL;
    NarrowX3Args a;
    a.x = x; a.w = (const bf16*)w_frag; a.bias = bias32; a.y = y;
    a.B = B; a.IH = IH; a.IW = IW; a.OH = OH; a.OWg = OWg;
    a.off_h = off_h; a.off_w = off_w; a.act = act; a.reflect = reflect;
    // 8 rows per workgroup, weight fragments one set ahead (120 registers: two workgroups per CU).  Measured per c1 step, heads forward +
    // image gradient: 1.17 ms; two sets ahead (132 registers, one workgroup per CU) 1.26; 16 rows 1.16 / 1.17 -- not instantiated.
    a.blocks_x = (OWg + NX_GROUPS - 1) / NX_GROUPS;
    a.blocks_y = (OH + 7) / 8;
    hipLaunchKernelGGL((conv_narrow_x3_kernel<7, 14, 1, 1>), dim3(a.blocks_x * a.blocks_y * B), dim3(512), 0, (hipStream_t)stream, a);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

/* out (bf16, [3][n]) = the three planes of the exact split of src[idx[i]] (idx[i] < 0: zero), i < n: filter-bank preparation of
 * dwc_x3_conv2d_narrow / dwc_x3_conv2d_stem from an index table (a fixed permutation + padding of the OIHW filter). */
int dwc_x3_gather_split(const float* src, const int* idx, void* out, int n, void* stream) {
    if (!src || !idx || !out || n <= 0) return DWC_EINVAL;
    hipLaunchKernelGGL(x3_gather_split_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, idx, (unsigned short*)out, n);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

int dwc_x3_conv2d_stem_ok(int B, int IH, int IW, int OH, int OW, int K, int act) { return stem_x3_ok(B, IH, IW, OH, OW, K, act) ? 1 : 0; }

/* bf16 elements of the split filter of dwc_x3_conv2d_stem: [3 planes][13 k-steps][64 channels][16] */
size_t dwc_x3_conv2d_stem_weight_elems(void) { return (size_t)3 * 13 * 64 * 16; }

/* y[B][OH][OW][64] (fp32) = act(conv7x7(x[B][IH][IW][4 planes] fp32) + bias) as split products, window of output (oy, ox) starting
 * at (oy + off, ox + off); reflect != 0: reflect rule (stems forward, off = -3), else zero rule (data gradient of the image heads on
 * the padded grid, off = -6).  act: none / relu / lrelu.  w_steps: the OIHW filter [64][4][7][7] as three exact bf16 planes
 * [plane][13][64][16]: element (k-step j, channel co, tap 4j + 2h + t, image plane p) at ((h ^ ((co>>3)&1))*8 + 4t + p), taps
 * beyond 48 zero (built by the caller).  crop > 0: output pixels whose cropped coordinate lies inside (OH - 2 crop) x (OW - 2 crop)
 * go to `inner` ([B][OH-2crop][OW-2crop][64]) instead of y (see dwc_bf16_conv2d_stem_crop). */
int dwc_x3_conv2d_stem_crop(const float* x, const void* w_steps, const float* bias, float* y, float* inner, int crop, int B, int IH, int IW,
                            int OH, int OW, int K, int off, int act, int reflect, void* stream) {
    if (!x || !w_steps || !y || !stem_x3_ok(B, IH, IW, OH, OW, K, act) || crop < 0 || (crop > 0 && (!inner || OH <= 2 * crop || OW <= 2 * crop)))
        return DWC_EINVAL;
    StemX3Args a;
    a.x = x; a.w = (const bf16*)w_steps; a.bias = bias; a.y = y;
    a.crop = crop; a.CH = OH - 2 * crop; a.CW = OW - 2 * crop; a.inner = inner;
    a.B = B; a.IH = IH; a.IW = IW; a.OH = OH; a.OW = OW; a.off = off; a.act = act; a.reflect = reflect;
    a.blocks_x = (OW + 15) / 16; a.blocks_y = (OH + 15) / 16; a.nblocks = a.blocks_x * a.blocks_y * B;
    const int grid = a.nblocks < 256 ? a.nblocks : 256;              // one persistent workgroup per CU (the filter alone is 78 KB)
    hipLaunchKernelGGL(conv_stem_x3_kernel, dim3(grid), dim3(512), 0, (hipStream_t)stream, a);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

int dwc_x3_conv2d_stem(const float* x, const void* w_steps, const float* bias, float* y, int B, int IH, int IW, int OH, int OW, int K,
                       int off, int act, int reflect, void* stream) {
    return dwc_x3_conv2d_stem_crop(x, w_steps, bias, y, nullptr, 0, B, IH, IW, OH, OW, K, off, act, reflect, stream);
}

/* dwc_x3_conv2d_stem raising the absmax slot of y from its store pass (see dwc_instnorm_fwd_amax; y_amax NULL: the plain call) */
int dwc_x3_conv2d_stem_amax(const float* x, const void* w_steps, const float* bias, float* y, void* y_amax, unsigned y_epoch, int B, int IH,
                            int IW, int OH, int OW, int K, int off, int act, int reflect, void* stream) {
    if (!x || !w_steps || !y || !stem_x3_ok(B, IH, IW, OH, OW, K, act)) return DWC_EINVAL;
    StemX3Args a;
    a.x = x; a.w = (const bf16*)w_steps; a.bias = bias; a.y = y;
    a.crop = 0; a.CH = OH; a.CW = OW; a.inner = nullptr;
    a.B = B; a.IH = IH; a.IW = IW; a.OH = OH; a.OW = OW; a.off = off; a.act = act; a.reflect = reflect;
    a.ys = (unsigned long long*)y_amax; a.ys_epoch = y_epoch;
    a.blocks_x = (OW + 15) / 16; a.blocks_y = (OH + 15) / 16; a.nblocks = a.blocks_x * a.blocks_y * B;
    const int grid = a.nblocks < 256 ? a.nblocks : 256;
    hipLaunchKernelGGL(conv_stem_x3_kernel, dim3(grid), dim3(512), 0, (hipStream_t)stream, a);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

size_t dwc_x3_conv7_smallk_wgrad_ws_bytes(int B, int H, int W, int heads) {
    int splits, ups, ux, upi;
    smallk_x3_plan(B, heads ? H + 6 : H, heads ? W + 6 : W, &splits, &ups, &ux, &upi);
    return (size_t)splits * 7 * 32 * 64 * sizeof(float);
}

/* Weight gradient of the two 7x7 layer shapes between an NHWC4 fp32 image (img4: [B][H][W][4]) and a 64-channel fp32 tensor
 * (t64: [B][H][W][64]), pad 3, reflect padding in the forward, as exact split products (smallk_wgrad_x3_kernel):
 *   heads == 0 (stems, 4 -> 64): img4 = the input image x, t64 = dY;                            dw: [64][planes][7][7] fp32
 *   heads != 0 (image heads, 64 -> 4): img4 = dY (pre-activation gradient), t64 = the input x;   dw: [planes][64][7][7] fp32
 * `planes` <= 4 real planes are written.  Scratch: dwc_x3_conv7_smallk_wgrad_ws_bytes.  Replaces the same reference call sites
 * (networks_v2.py:106,159-160, networks.py:432 through autograd) as dwc_bf16_conv7_smallk_wgrad. */
int dwc_x3_conv7_smallk_wgrad(const float* img4, const float* t64, float* dw, int B, int H, int W, int planes, int heads, void* ws,
                              size_t ws_bytes, void* stream) {
    if (!img4 || !t64 || !dw || B <= 0 || H < 7 || W < 7 || planes < 1 || planes > 4) return DWC_EINVAL;
    SmallWgradX3Args a;
    int splits;
    a.GH = heads ? H + 6 : H; a.GW = heads ? W + 6 : W;
    smallk_x3_plan(B, a.GH, a.GW, &splits, &a.units_per_split, &a.units_x, &a.units_per_img);
    if (!ws || ws_bytes < (size_t)splits * 7 * 32 * 64 * sizeof(float)) return DWC_EWORKSPACE;
    a.a4 = img4; a.b64 = t64; a.slab = (float*)ws;
    a.B = B; a.AH = H; a.AW = W; a.BH = H; a.BW = W;
    a.offA = heads ? -6 : -3; a.offB = heads ? -3 : 0; a.reflA = heads ? 0 : 1; a.reflB = heads ? 1 : 0;
    a.total_units = B * a.units_per_img;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(smallk_wgrad_x3_kernel, dim3(splits), dim3(512), 0, st, a);
    DWC_LAUNCH_CHECK();
    hipLaunchKernelGGL(smallk_x3_reduce_kernel, dim3((7 * 32 * 64 + 31) / 32), dim3(256), 0, st, (const float*)ws, dw, splits, planes,
                       heads ? 49 : planes * 49, heads ? 64 * 49 : 49, heads ? 1 : 0);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

}  // extern "C"
