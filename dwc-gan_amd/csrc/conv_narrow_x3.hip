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

// MT: 32-group tiles per workgroup (8 rows each).  1: 70 KB of LDS, <= 128 registers, two workgroups per CU.  2: 16 rows, 110 KB,
// one workgroup per CU -- every weight fragment (1 KB per wave, tap, slab and plane, streamed from L2: 1.2 MB per block) feeds two
// tiles: the MT = 1 form is bound by exactly that stream (2.4 MB per CU and ~20 us: ~12 TB/s over the chip).
template <int KH, int KWW, int MT>
__global__ __launch_bounds__(512, MT == 1 ? 2 : 1) void conv_narrow_x3_kernel(NarrowX3Args a) {
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
    bf16x8 wset[3][3];
    auto load_item = [&](int j, bf16x8 (&dst)[3]) {     // item j = (slab j / NT_W, tap wave + 8 * (j % NT_W))
        const int qq = j / NT_W;
        const bf16* src = wlane + ((size_t)(qq * NTAP_P + wave + 8 * (j - qq * NT_W)) * 3) * 512;
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) dst[pl] = *reinterpret_cast<const bf16x8*>(src + pl * 512);
    };
    load_patch(0);
    load_item(0, wset[0]);
    load_item(1, wset[1]);
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
            if (item + 2 < N_ITEMS) load_item(item + 2, wset[(item + 2) % 3]);
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
            bf16x8 (&cur)[3] = wset[item % 3];
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

bool narrow_x3_ok(int B, int IH, int IW, int Cin, int OH, int OWg, int KH, int KWW) {
    return B > 0 && IH > 0 && IW > 0 && Cin == NX_CH && OH > 0 && OWg > 0 && KH == 7 && KWW == 14 &&
           (size_t)B * IH * IW * NX_CH * 4 < 0x7fffffffull;
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
    // rows per workgroup: 16 (every weight fragment feeds two tiles: the kernel is bound by that stream) unless the launch would then
    // leave CUs idle; DWC_X3_NARROW_ROWS=8|16 pins it
    static const int force = getenv("DWC_X3_NARROW_ROWS") ? atoi(getenv("DWC_X3_NARROW_ROWS")) : 0;
    a.blocks_x = (OWg + NX_GROUPS - 1) / NX_GROUPS;
    const long blocks16 = (long)a.blocks_x * ((OH + 15) / 16) * B;
    // (measured, c1: 16 rows 105 / 304 us against 97 / 294 us at batch 16 / 48 -- one workgroup per CU loses what the halved weight
    // stream gains; 8 rows stay the default)
    const int rows = force == 16 ? 16 : 8;
    a.blocks_y = (OH + rows - 1) / rows;
    const dim3 grid(a.blocks_x * a.blocks_y * B);
    (void)blocks16;
    if (rows == 16) hipLaunchKernelGGL((conv_narrow_x3_kernel<7, 14, 2>), grid, dim3(512), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((conv_narrow_x3_kernel<7, 14, 1>), grid, dim3(512), 0, (hipStream_t)stream, a);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

}  // extern "C"
