// fp32 convolution of the stride-1 "same" 3x3 / 5x5 layers on the bf16 matrix cores, exact split products.
// (ResBlock and upsampling-block convolutions, reference networks.py:514-515, networks_v2.py:153-156; forward and the interior
// of the data gradient.)
//
// MI355X has 157 TFLOP/s of fp32 MFMA and 2.5 PFLOP/s of bf16 MFMA, a ratio of 16.  An fp32 number is EXACTLY the sum of three
// bf16 numbers, a = a0 + a1 + a2 (8 significand bits each: truncate, subtract, truncate, subtract -- no rounding anywhere), a
// product of two bf16 numbers is exact in fp32 (16 significand bits), and the matrix core accumulates in fp32.  So
//     a*b = a0*b0 + (a0*b1 + a1*b0) + (a0*b2 + a1*b1 + a2*b0) + [a1*b2 + a2*b1 + a2*b2]
// where the bracket is below 2^-23 |a*b| -- the size of ONE fp32 rounding of the product, which the native fp32 MFMA commits
// as well.  Six bf16 MFMAs per fp32 MFMA-equivalent cost 6/16 of the native instruction's time: fp32-accurate results at up to
// 2.6x the fp32 MFMA peak (tests/test_x3_parity.py measures both paths against a float64 product: same error).
//
// Kernel shape (the halo form of conv_halo_bf16.hip): a workgroup owns a 16x16 block of output pixels of one image and BN
// output channels.  Per 16-channel slab the (16+K-1)^2 input patch is read ONCE from the fp32 tensor (reflect or zero rule
// applied while gathering), split in registers and written to LDS as three bf16 planes [pixel][16 ch] (32-byte rows: every
// fragment read is a contiguous, conflict-free ds_read_b128); all K*K taps read that patch at a pixel offset.  The weights are
// split once per optimiser step by dwc_x3_weight_prepare into [tap][slab][plane][channel][16] and arrive by LDS-DMA through a
// three-slot ring (two taps ahead), one barrier per tap.  Per tap a wave issues 3*(TM+TN) fragment reads for 6*TM*TN MFMAs
// (0.375 per MFMA at 4x2 tiles; the plain bf16 kernels need 0.75 and are LDS-read bound).
// D = W_tile . X_tile^T, so a lane owns a pixel and 4 consecutive output channels per accumulator quad: fp32 results are stored
// straight from registers (16 bytes per lane) with bias and activation applied.
#include <type_traits>

#include "conv_geom.h"

namespace {

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

constexpr int TB = 16;                 // block edge: 16x16 output pixels
constexpr int CS = 16;                 // channels per slab = one MFMA k-step

// fp32 x4 -> three planes of 4 bf16 (packed two per dword), exact: v = p0 + p1 + p2
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3(f32x4 v, u32x2& p0, u32x2& p1, u32x2& p2) {
    // (pairs: the two subtractions compile to v_pk_add_f32; the high halves of v and of r = v - hi are the hi / mid planes as they are)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const f32x2 x = {v[2 * k], v[2 * k + 1]};
        const f32x2 hb = {__uint_as_float(__float_as_uint(x[0]) & 0xffff0000u), __uint_as_float(__float_as_uint(x[1]) & 0xffff0000u)};
        const f32x2 r = x - hb;
        const f32x2 mb = {__uint_as_float(__float_as_uint(r[0]) & 0xffff0000u), __uint_as_float(__float_as_uint(r[1]) & 0xffff0000u)};
        const f32x2 l = r - mb;
        // bytes 2,3 of the even element below bytes 2,3 of the odd one
        p0[k] = __builtin_amdgcn_perm(__float_as_uint(x[1]), __float_as_uint(x[0]), 0x07060302u);
        p1[k] = __builtin_amdgcn_perm(__float_as_uint(r[1]), __float_as_uint(r[0]), 0x07060302u);
        p2[k] = __builtin_amdgcn_perm(__float_as_uint(l[1]), __float_as_uint(l[0]), 0x07060302u);
    }
}

// ---- two-plane f16 split ("h2", r05) ------------------------------------------------------------------------------------------
// s*a = hi + lo with hi = f16(s*a) and lo = f16(s*a - hi), both ROUNDED TO NEAREST: hi holds 11 significand bits, the residual (exact
// in fp32) is at most half an ulp of hi and lo holds 11 of its bits -- |s*a - hi - lo| <= 2^-24 |s*a|, the size of an fp32 rounding.
// a*b = [hi*hi + hi*lo + lo*hi] / (sa sb) up to lo*lo <= 2^-24 |ab|: THREE f16 MFMAs per fp32 MFMA-equivalent instead of the six of
// the three-plane bf16 split (f16 products are 22 bits: exact in the fp32 accumulator).  f16 has 5 exponent bits, so every operand
// TENSOR is brought to a working range by a power of two s (exact): its largest magnitude lands in [2^13, 2^14); values down to
// 2^-16 of it keep all 22-24 bits (their residual is a normal f16), smaller ones lose bits gradually (f16 subnormals, which the
// MFMA honours: benchmarks/split2_lab.hip) with an absolute error of at most 2^-38 of the tensor's largest magnitude -- invisible at
// the output scale.  (r05 first carried lo as (s*a - hi) * 2^11: full precision down to 2^-27 of the largest magnitude, but then the
// corrections weigh 2^-11 and need an accumulator set of their own.  With lo unscaled all three products have the same weight and
// share ONE matrix-core accumulator, which is flushed into fp32 vector accumulators every 16-25 k-steps anyway; the 64 registers
// pay for a second fragment set.  Lab, K = 6400: 2.6e-7 of the output scale against 2.0e-7 -- the native fp32 MFMA: 3.1e-6.)
// The largest magnitude comes from a caller-owned 64-bit slot: (epoch << 32) | bits of max |a|, raised by atomic max from the
// kernel that produced the tensor or from dwc_absmax; a slot whose epoch is not the one the caller names poisons the result
// with NaN (a stale or never-written slot must not pass for a small tensor).  Non-finite input: the slot reads inf / NaN, s = 1,
// and the NaN of inf - inf in the lo plane reaches every output the value touches.
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
// fp32 x4 and the tensor's scale s -> two planes of 4 f16: hi = f16(s v), lo = f16(s v - hi).  Four vector instructions per pair of
// elements: one packed multiply, one packed conversion, and the lo plane straight from v_fma_mixlo/hi_f16 -- fma(hi read as f16,
// -1, s v), exact in fp32, rounded once to f16 and written to its half of the packed register (the C form -- convert hi back,
// subtract, convert -- compiles to six; the fused form was checked bit-identical on 2^20 values incl. +-0, inf, NaN).  The
// weight-gradient kernel is bound by exactly this arithmetic.
__device__ __forceinline__ void split2h(f32x4 v, float s, u32x2& p0, u32x2& p1) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const f32x2 x = {v[2 * k], v[2 * k + 1]};
        const f32x2 t = x * s;
        const unsigned h = __builtin_bit_cast(unsigned, __builtin_convertvector(t, f16x2));
        unsigned l;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "v"(t[0]));
        asm("v_fma_mixhi_f16 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "v"(t[1]));
        p0[k] = h;
        p1[k] = l;
    }
}
template <int NPL> struct Planes { u32x2 p[NPL]; };
template <int NPL> __device__ __forceinline__ Planes<NPL> split_planes(f32x4 v, float s) {
    Planes<NPL> r;
    if constexpr (NPL == 3) split3(v, r.p[0], r.p[1], r.p[2]);
    else split2h(v, s, r.p[0], r.p[1]);
    return r;
}
template <int NPL> __device__ __forceinline__ f32x16 x3_mfma(bf16x8 a, bf16x8 b, f32x16 c) {
    if constexpr (NPL == 3) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}

struct X3Args {
    const float* x;      // [B][H][W][Cin] fp32
    const bf16* w;       // [K*K][Cin/16][NPL][rows][16]   (dwc_x3_weight_prepare / dwc_h2_weight_prepare: + {s, 1/s} behind it)
    const float* bias;   // [N] or null
    const float* add;    // [B][H][W][N] or null: added behind bias + activation (a second gradient w.r.t. the same tensor)
    float* y;            // [B][H][W][N] fp32
    int B, H, W, Cin, N, rows, act, reflect;
    int blocks_x, blocks_per_img, tiles_n;
    float* part = nullptr;          // KSP == 2: [tiles][256 pixels x 64 channels] fp32, the first arriver's half sum
    unsigned* tickets = nullptr;    // KSP == 2: [tiles], zero between launches (caller-owned, self-resetting)
    unsigned* status = nullptr;     // KSP == 2: sticky word, bit 0 set when a tile's hand-off expired or crossed XCDs (the tile is NaN)
    int split_from = 0;             // KSP == 2: tiles [0, split_from) run whole (one workgroup), [split_from, tiles) as two halves
    const unsigned long long* xs = nullptr;      // NPL == 2: absmax slot of x and the epoch it must carry
    unsigned xs_epoch = 0;
    size_t w_elems = 0;             // NPL == 2: 16-bit elements of the prepared planes; {s_w, 1 / s_w} (fp32) sit behind them
    unsigned long long* ys = nullptr;            // NPL == 2, optional: absmax slot of y, raised from the store pass (epoch ys_epoch)
    unsigned ys_epoch = 0;
};

// WM x WN waves (8), each TM x TN 32x32 accumulators: block = 256 pixels x BN channels
// DBG (development, timing only -- results are wrong when set): 1 no MFMA, 2 no fragment reads, 4 no weight staging in the
// loop, 8 no barrier, 16 no patch refresh
// PB: patch buffers.  2: the next slab's patch is converted during the taps of the current one.  1: it is converted at the slab
// boundary (exposed, but the workgroup then fits TWICE on a CU: 4-wave workgroups of 256 pixels x 64 channels, <= 80 KB of
// LDS and 256 registers -- two independent workgroups per CU drift out of phase, so one's MFMAs run beside the other's
// fragment reads, staging and barriers, which the eight lock-stepped waves of one workgroup never do).
// S2 (KS == 2): the 4x4 stride-2 reflect-pad-1 layers (reference networks.py:90,94,437, networks_v2.py:107-111) as 2x2-tap
// stride-1 convolutions over the space-to-depth image, the space-to-depth done by the patch gather: a "slab" is (input-pixel
// parity (dy, dx), 16-channel slab), patch pixel (py, px) of it is input pixel (2 (y0 + py) + dy - 1, 2 (x0 + px) + dx - 1) under
// the reflect rule, and tap (th, tw) of that slab is filter tap (2 th + dy, 2 tw + dx) of the ordinary K = 4 prepared weights.
// a.H, a.W are the input dimensions, the output is H/2 x W/2.
// S2 == 2 (KS == 2): the INTERIOR of the data gradient of those layers.  Output pixel (h, w) of parity (ry, rx) = (h & 1, w & 1)
// receives dy[i' - 1 + ry + a][j' - 1 + rx + b] . W[.][.][3 - ry - 2a][3 - rx - 2b] over a, b in {0, 1}, (i', j') = (h >> 1, w >> 1):
// per output parity class a 2x2-tap zero-padded convolution over dY, patch origin (y0 - 1 + ry, x0 - 1 + rx).  The class is
// the fastest part of the block index (the four classes of a tile read the same dY pixels), a.H x a.W is the dY grid, the
// result is scattered to (2 i' + ry, 2 j' + rx) of the 2 a.H x 2 a.W tensor.  With the data-gradient weights of
// dwc_x3_weight_prepare (K = 4, filter rotated) kernel tap (a, b) is prepared tap (ry + 2a, rx + 2b).  The border ring of the
// padded image (the reflect rule's adjoint) is dwc_conv2d_bwd_data_s2_ring's.
// KSP == 2: launches of at most 256 tiles (3x3 256->256 at batch 16: one 4-wave workgroup per CU, nothing beside it to overlap
// with, 0.39 of the MFMA peak against 0.52 at two per CU) are cut along the CONTRACTION instead: two workgroups per tile, each
// walks half of the channel slabs.  Whichever of the two takes the tile's ticket first publishes its half sum in `part` and
// leaves; the other waits for that (the first is running and waits for nobody), adds it to its own -- a + b = b + a, so the
// result does not depend on the order of arrival -- and runs the epilogue.  The ticket is back at zero when the tile is done.
// The same instantiation serves launches whose LAST round of workgroups would be at most half full (768 tiles on 512 slots: the
// 256 stragglers run one per CU): tiles [0, split_from) run whole and are dispatched first, only the tail is split.
// NPL: planes per operand -- 3: exact bf16 split, six products; 2: f16 hi / lo split with per-tensor power-of-two scales, three
// products (see h2_scale above).
// RING (r06; two-plane form, stride-1 data gradients, a.reflect == 0): the border ring of the padded gradient image -- the reflect
// rule's adjoint, rounds 2-5: an im2col strip GEMM + a fold launch behind every data gradient -- comes out of THIS launch.  Ring row -m
// folds onto block row m (top tiles), ring column -x onto block column x (left tiles; bottom / right mirrored): under the taps that
// reach real dY pixels (kh >= p + m, kw >= p + x) such a pixel needs, besides its regular patch pixel (m + kh, x + kw), the pixels
// (kh - m, x + kw), (m + kh, kw - x) and -- in a corner -- (kh - m, kw - x).  The MFMA is linear in the pixel operand and a lane owns a
// pixel, so those lanes simply read a PRE-SUMMED pixel: after every patch conversion a border tile adds the n = p(p+1)/2 row / column
// pairs in fp32 (from the raw fp32 patch in LDS: exact up to one fp32 rounding of the sum), splits the sums into the two planes and
// stores them as n extra patch rows and columns; the fragment address of lane (row, column) under tap (kh, kw) gets a per-lane offset
// (two registers per pixel tile for the rows, two for the columns, selected by the tap).  No extra MFMA.  See conv_halo16_bf16.inc
// for the bf16 twin.
template <int KS, int BN, int WM, int WN, int TM, int TN, int DBG = 0, int PB = 2, int S2 = 0, int KSP = 1, int NPL = 3, int RING = 0>
__global__ __launch_bounds__(64 * WM * WN, WM * WN == 4 ? 2 : 1) void conv_halo_x3_kernel(X3Args a) {
#if defined(__HIP_DEVICE_COMPILE__)
    // 8 waves, two per SIMD, one workgroup per CU -- or 4 waves and two workgroups per CU (PB == 1).  (4 "fat" waves, one per
    // SIMD with up to 512 registers -- 4x2 tiles, a second fragment set, all latency hiding inside the wave's own instruction
    // stream -- compile from the same source and were measured 15-25 % SLOWER.  r05, two-plane form: 8 waves of 2x1 tiles, ~101 registers,
    // TWO such workgroups per CU = 4 waves per SIMD, no second fragment set: 3-7 % slower than the 4-wave 2x2 tile -- 188 / 285 / 285
    // against 182 / 267 / 268 us, profiles/r05_h2_ablation.txt -- the smaller tile reads 6 fragments per 6 MFMAs instead of 8 per 12.)
    constexpr int NW = WM * WN, THREADS = 64 * NW;
    constexpr int PPT = THREADS / 4;                   // patch pixels per gather pass (4 threads x 4 channels per pixel)
    static_assert((NW == 8 || NW == 4) && WM * TM * 32 == 256 && WN * TN * 32 == BN, "tile shape");
    static_assert(!S2 || KS == 2, "stride-2 form: 2x2 taps per parity");
    static_assert(KSP == 1 || (KSP == 2 && S2 != 2 && NW == 4 && TM * TN <= 4), "contraction split: the two-per-CU tile only");
    static_assert(!RING || (NPL == 2 && PB == 1 && ((S2 == 0 && (KS == 3 || KS == 5)) || S2 == 2)), "fused border ring: two-plane data gradients");
    constexpr int PW = TB + KS - 1;                    // patch edge
    // RING: pre-summed rows / columns behind the PW real ones (stride 1: p (p + 1) / 2; the stride-2 data gradient: one)
    constexpr int NALT = RING ? (S2 == 2 ? 1 : ((KS - 1) / 2) * ((KS - 1) / 2 + 1) / 2) : 0;
    constexpr int PWA = PW + NALT;
    constexpr int PPIX = PW * PW;                      // patch pixels
    constexpr int PPASS = (PPIX + PPT - 1) / PPT;      // gather passes
    // LDS image of a patch plane: pixel (py, px) at py*PITCH + px*16 elements, PITCH = PW*16 + 8: consecutive patch rows are
    // offset by HALF a 32-byte pixel slot.  A ds_read_b128 is served in lane groups {0-3,12-15,20-27} / {4-11,16-19,28-31}
    // (MI355X_MICROARCH.md), i.e. 8 pixels of one patch row + 8 of the next: with a plain pixel-linear image both rows hit
    // the same eight 16-byte bank slots (every fragment read 2-way conflicted); with the half-slot offset all 16 differ.
    constexpr int PITCH = PWA * CS + 8;
    constexpr int P_PLANE = PWA * PITCH;               // elements per plane of a patch buffer
    constexpr int W_CHUNKS = NPL * BN * 2;             // 16-byte chunks of one tap's weight slab (NPL planes x BN rows x 32 B)
    constexpr int W_INSTR = (W_CHUNKS + THREADS - 1) / THREADS;    // LDS-DMA instructions per thread and tap
    constexpr int W_SLOT = W_INSTR * THREADS * 8;      // elements per ring slot (whole instructions)
    constexpr int NTAP = KS * KS;
    constexpr int PAD = (KS - 1) / 2;
    // RAW (the two-plane form): the next slab's fp32 patch travels global -> LDS by DMA (16 bytes per lane, the reflect / zero rule in
    // the per-lane buffer offset) and is split from there at the slab boundary, instead of waiting in PPASS x 4 registers for a
    // whole slab.  The registers pay for `big`: every FLUSH slabs the leading-product accumulators are added to fp32 VALU
    // accumulators (round to nearest) and restart from zero.  The error of a long contraction on the matrix cores is dominated by
    // the accumulate of each MFMA into an accumulator that has grown large (benchmarks/split2_lab.hip: K = 6400, error / output
    // scale 8.9e-7 unflushed, 2.0e-7 flushed every 25 steps -- a third of the native fp32 MFMA's).
    // (Ablation of this form, profiles/r05_h2_ablation.txt, 3x3 256->256 at B=48: the conversion at the slab boundary accounts for 54
    // of 219 us.  Spreading it over the taps of the previous slab into a SECOND pair of plane buffers -- 79.5 KB of LDS, no extra
    // barrier -- was built and measured SLOWER: 3x3 170 -> 189 us, stride 2 177 -> 201 us, c1 332 -> 319 images/s: the work moves
    // into the taps, where it delays this wave's MFMAs, while at the boundary the CU's OTHER workgroup covers it.  Not kept.)
    constexpr bool RAW = NPL == 2;
    static_assert(!RAW || PB == 1, "raw patch staging: the single-buffer form");
    constexpr int FLUSH = !RAW ? 0 : (NTAP >= 25 ? 1 : (NTAP >= 9 ? 2 : 4));      // slabs between flushes: 16 - 25 k-steps
    constexpr int RAW_ELEMS = RAW ? PPASS * THREADS * 8 : 0;                      // 16 bytes per thread and pass, in 2-byte elements
    // PIPE (the two-plane form): two fragment sets, see the main loop
    constexpr bool PIPE = NPL == 2 && PB == 1 && TM * TN <= 4;
    constexpr int NRING = 3, AHEAD = NRING - 1;        // weight ring: filled two taps ahead
    __shared__ __attribute__((aligned(16))) bf16 smem[PB * NPL * P_PLANE + NRING * W_SLOT + RAW_ELEMS];
    bf16* sP = smem;                                   // [PB buffers][NPL planes][pixel][16]
    bf16* sW = smem + PB * NPL * P_PLANE;              // [NRING slots][NPL planes][BN][16]
    bf16* sR = sW + NRING * W_SLOT;                    // RAW: [pass][thread] f32x4

    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, hi = lane >> 5;
    H2Scale sx = {1.f, 1.f}, sw = {1.f, 1.f};
    if constexpr (NPL == 2) {
        sx = h2_scale(a.xs, a.xs_epoch);
        const float* wt = reinterpret_cast<const float*>(a.w + a.w_elems);
        sw.s = wt[0];
        sw.inv = wt[1];
    }

    int bid = blockIdx.x;
    int khalf = 0;                                                       // KSP == 2: which half of the channel slabs
    bool halved = false;                                                 // KSP == 2: this workgroup is one half of a split tile
    if constexpr (KSP == 2) {
        // hardware workgroups h and h + 8 run on the same XCD (h % 8): they are the two halves of a tile, so that the half sum
        // travels through THAT XCD's L2 and nothing wider (split_from and the number of split tiles are multiples of 8); XCD x
        // owns a contiguous range of the whole tiles and one of the split tiles, as in the plain remap below
        const int x = bid & 7;
        if (bid < a.split_from) {
            bid = x * (a.split_from >> 3) + (bid >> 3);
        } else {
            const int slot = (bid - a.split_from) >> 3;
            khalf = slot & 1;
            halved = true;
            bid = a.split_from + x * (int)((gridDim.x - a.split_from) >> 4) + (slot >> 1);
        }
    } else {
        const int nb = gridDim.x;
        if (nb >= 16) {     // XCD-aware remap (neighbouring blocks share halo pixels and the weight slabs in one L2)
            const int q = nb >> 3, r = nb & 7, x = bid & 7, y = bid >> 3;
            bid = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
        }
    }
    int ry = 0, rx = 0;                                                  // S2 == 2: output parity class of this workgroup
    if constexpr (S2 == 2) {
        ry = (bid >> 1) & 1;
        rx = bid & 1;
        bid >>= 2;
    }
    const int tile_id = bid;
    const int tile_n = bid % a.tiles_n, blk = bid / a.tiles_n;
    const int n_img = blk / a.blocks_per_img, bi = blk - n_img * a.blocks_per_img;
    const int by = bi / a.blocks_x, bx = bi - by * a.blocks_x;
    const int y0 = by * TB, x0 = bx * TB, n0 = tile_n * BN;
    const int OH = S2 == 1 ? a.H >> 1 : a.H, OW = S2 == 1 ? a.W >> 1 : a.W;      // grid the blocks tile (S2 == 2: the dY grid)
    const int ncsr = a.Cin / CS;                                        // 16-channel slabs of the input tensor
    const int ncs_all = S2 == 1 ? 4 * ncsr : ncsr;                      // slabs of the contraction: x 4 input-pixel parities
    const int ncs = halved ? ncs_all >> 1 : ncs_all, cs0 = khalf * ncs; // slabs this workgroup walks: cs0 .. cs0 + ncs - 1
    const int nsteps = (DBG & 512) ? 0 : ncs * NTAP;      // (DBG: development ablations, DWC_DEV_ABLATIONS builds only)

    // ---- patch gather map: thread = (patch pixel t>>2 [+128 per pass], channel quad t&3) -----------------------------------
    const float* p_src[(S2 == 1 || RAW) ? 1 : PPASS];
    unsigned p_off[RAW ? PPASS : 1];                    // RAW: byte offset of the pixel's channel quad in x, 2^31 = "reads as zero"
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(a.x), 0, (unsigned)min((size_t)0xffffffffu, (size_t)a.B * a.H * a.W * a.Cin * 4), 0x00020000);
    // S2: element offset of the patch pixel's input pixel for parity (0, 0), and what the odd row / column parity adds to it
    // (kept as base + masked delta: indexing a register array by the parity would send it to scratch)
    int p_base[S2 == 1 ? PPASS : 1], p_drow[S2 == 1 ? PPASS : 1], p_dcol[S2 == 1 ? PPASS : 1];
    int p_dst[PPASS];                                   // LDS element offset of (py, px), channels 4*(t&3)..
    unsigned p_ok = 0, p_in = 0;                        // loads are unconditional (the vmcnt arithmetic below counts them)
#pragma unroll
    for (int i = 0; i < PPASS; ++i) {
        const int pp = (t >> 2) + PPT * i;
        const int py = pp / PW, px = pp - py * PW;
        bool ok = pp < PPIX;
        if constexpr (S2 == 1) {
            const int h0 = min(reflect_idx(2 * (y0 + py) - 1, a.H), a.H - 1), h1 = min(reflect_idx(2 * (y0 + py), a.H), a.H - 1);
            const int w0 = min(reflect_idx(2 * (x0 + px) - 1, a.W), a.W - 1), w1 = min(reflect_idx(2 * (x0 + px), a.W), a.W - 1);
            p_base[i] = ((n_img * a.H + h0) * a.W + w0) * a.Cin;
            p_drow[i] = (h1 - h0) * a.W * a.Cin;
            p_dcol[i] = (w1 - w0) * a.Cin;
        } else {
            int h = y0 - PAD + py, w = x0 - PAD + px;
            if constexpr (S2 == 2) {
                h = y0 - 1 + ry + py;
                w = x0 - 1 + rx + px;
            }
            if (a.reflect) {
                h = reflect_idx(h, a.H);
                w = reflect_idx(w, a.W);
            } else {
                ok = ok && (unsigned)h < (unsigned)a.H && (unsigned)w < (unsigned)a.W;
            }
            h = min(max(h, 0), a.H - 1);
            w = min(max(w, 0), a.W - 1);
            if constexpr (RAW) p_off[i] = ok ? (unsigned)(((n_img * a.H + h) * a.W + w) * a.Cin + (t & 3) * 4) * 4u : 0x80000000u;
            else p_src[i] = a.x + ((size_t)(n_img * a.H + h) * a.W + w) * a.Cin + (t & 3) * 4;
        }
        p_dst[i] = py * PITCH + px * CS + (t & 3) * 4;
        p_ok |= ok ? 1u << i : 0u;
        p_in |= pp < PPIX ? 1u << i : 0u;
    }
    f32x4 pv[RAW ? 1 : PPASS];
    auto load_patch = [&](int cs_local) {
        const int cs = cs0 + cs_local;
        if constexpr (RAW) {
            bf16* lr = sR + wave * 512;
            if constexpr (S2 == 1) {
                const int par = cs / ncsr, csl = cs - par * ncsr;
                const int my = -(par >> 1), mx = -(par & 1);
                const int soff = __builtin_amdgcn_readfirstlane(csl * CS * 4);
#pragma unroll
                for (int i = 0; i < PPASS; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(lr + i * THREADS * 8), 16,
                                                             (unsigned)(p_base[i] + (p_drow[i] & my) + (p_dcol[i] & mx) + (t & 3) * 4) * 4u, soff, 0, 0);
            } else {
                const int soff = __builtin_amdgcn_readfirstlane(cs * CS * 4);
#pragma unroll
                for (int i = 0; i < PPASS; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(lr + i * THREADS * 8), 16, p_off[i],
                                                             soff, 0, 0);
            }
        } else if constexpr (S2 == 1) {
            const int par = cs / ncsr, csl = cs - par * ncsr;
            const float* base = a.x + csl * CS + (t & 3) * 4;
            const int my = -(par >> 1), mx = -(par & 1);
#pragma unroll
            for (int i = 0; i < PPASS; ++i) pv[i] = *reinterpret_cast<const f32x4*>(base + (p_base[i] + (p_drow[i] & my) + (p_dcol[i] & mx)));
        } else {
#pragma unroll
            for (int i = 0; i < PPASS; ++i) pv[i] = *reinterpret_cast<const f32x4*>(p_src[i] + cs * CS);
        }
    };
    auto write_patch = [&](int buf) {
        bf16* dst = sP + buf * NPL * P_PLANE;
#pragma unroll
        for (int i = 0; i < PPASS; ++i) {
            if (!((p_in >> i) & 1)) continue;                             // slot past the patch
            f32x4 v;
            if constexpr (RAW) v = *reinterpret_cast<const f32x4*>(sR + (i * THREADS + t) * 8);      // (this lane's own DMA: zeros where the rule says so)
            else v = (p_ok >> i) & 1 ? pv[i] : f32x4{0.f, 0.f, 0.f, 0.f};
            const Planes<NPL> q = split_planes<NPL>(v, sx.s);
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<u32x2*>(dst + pl * P_PLANE + p_dst[i]) = q.p[pl];
        }
    };

    // ---- weight ring: chunk g = t + 512*p of the slab [plane][row][half] ----------------------------------------------------
    const unsigned w_bytes = (unsigned)(S2 ? 16 : NTAP) * ncsr * (unsigned)NPL * a.rows * CS * 2u;
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.w), 0, w_bytes, 0x00020000);
    unsigned w_off[W_INSTR];
#pragma unroll
    for (int p = 0; p < W_INSTR; ++p) {
        const int g = t + THREADS * p;
        const int plane = g / (2 * BN), rem = g - plane * 2 * BN;
        const int row = min(n0 + (rem >> 1), a.rows - 1);
        w_off[p] = g < W_CHUNKS ? (unsigned)(((plane * a.rows + row) * CS + (rem & 1) * 8) * 2) : 0x80000000u;
    }
    const int w_step_bytes = NPL * a.rows * CS * 2;                       // one (tap, slab) block of the prepared tensor
    auto stage_w = [&](int step, int slot) {                              // step = cs * NTAP + tap -> block tap * ncs + cs
        const int csl_ = step / NTAP, tap = step - csl_ * NTAP;
        const int cs = cs0 + csl_;
        bf16* lw = sW + slot * W_SLOT + wave * 512;
        int blk;
        if constexpr (S2 == 1) {
            const int par = cs / ncsr, csl = cs - par * ncsr;
            blk = ((2 * (tap >> 1) + (par >> 1)) * 4 + 2 * (tap & 1) + (par & 1)) * ncsr + csl;
        } else if constexpr (S2 == 2) {
            blk = ((ry + 2 * (tap >> 1)) * 4 + rx + 2 * (tap & 1)) * ncsr + cs;
        } else {
            blk = tap * ncs_all + cs;
        }
        const int soff = __builtin_amdgcn_readfirstlane(blk * w_step_bytes);
#pragma unroll
        for (int p = 0; p < W_INSTR; ++p)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, (__attribute__((address_space(3))) void*)(lw + p * THREADS * 8), 16, w_off[p],
                                                     soff, 0, 0);
    };

    // ---- fragments --------------------------------------------------------------------------------------------------------
    // pixel tile i of this wave: block pixel pb = (wm*TM + i)*32 + l31 -> patch pixel (pb>>4)*PW + (pb&15) + tap offset
    int pp0[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const int pb = (wm * TM + i) * 32 + l31;
        pp0[i] = (pb >> 4) * PITCH + (pb & 15) * CS + hi * 8;
    }
    // weight rows: the two 16-byte halves of row r are stored swapped when (r>>3)&1 (dwc_x3_weight_prepare), same reason
    const int b_row = (wn * TN * 32 + l31) * CS + ((hi ^ ((l31 >> 3) & 1)) * 8);

    // ---- RING: which border this tile touches, and the per-lane offsets that send a ring pixel's lane to its pre-summed pixel -------
    // slot(m, kl): the extra patch row (column) that holds row m + kh plus row kh - m for a top (left) tile under tap row (column) kl =
    // kh; bottom / right tiles are the mirror image (m counted from the far edge, kl = K - 1 - kh).  kl takes the values 2p and, for
    // p = 2, 2p - 1: "A" and "B" below.
    int row_side = 0, col_side = 0;                    // 1: top / left border tile, 2: bottom / right (workgroup-uniform)
    int khA = -1, khB = -1, kwA = -1, kwB = -1;        // the tap rows / columns that have redirected lanes in this tile
    int dRA[TM], dRB[TM], dCA = 0, dCB = 0;            // element offsets of those lanes under tap row khA / khB, tap column kwA / kwB
    auto ring_slot = [](int m, int kl) {
        int off = 0;
        for (int q = 1; q < m; ++q) off += PAD - q + 1;
        return off + kl - (PAD + m);
    };
#pragma unroll
    for (int i = 0; i < TM; ++i) dRA[i] = dRB[i] = 0;
    if constexpr (RING && S2 == 2) {
        // Stride-2 4x4 layers (a.H x a.W = the dY grid, one output parity class (ry, rx) per workgroup): padded row 0 = dY[0] . W[kh = 0]
        // folds onto dx row 1 = class row 0 of ry = 1, whose tap a = 1 carries W[kh = 0] and regularly reads dY row 1: its lanes read
        // dY[1] + dY[0] = patch rows 1 + 0.  Padded row H + 1 = dY[last] . W[kh = 3] folds onto dx row H - 2 = the last class row of
        // ry = 0, tap a = 0 (W[kh = 3], regular source patch row 15): patch rows 15 + 16.  Columns alike with rx.
        row_side = (y0 == 0 && ry == 1) ? 1 : ((y0 + TB == a.H && ry == 0) ? 2 : 0);
        col_side = (x0 == 0 && rx == 1) ? 1 : ((x0 + TB == a.W && rx == 0) ? 2 : 0);
        if (row_side) {
            khA = row_side == 1 ? 1 : 0;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = ((wm * TM + i) * 32 + l31) >> 4;
                if (r == (row_side == 1 ? 0 : 15)) dRA[i] = (PW - (r + khA)) * PITCH;
            }
        }
        if (col_side) {
            kwA = col_side == 1 ? 1 : 0;
            const int x = l31 & 15;
            if (x == (col_side == 1 ? 0 : 15)) dCA = (PW - (x + kwA)) * CS;
        }
    } else if constexpr (RING) {
        row_side = y0 == 0 ? 1 : (y0 + TB == a.H ? 2 : 0);
        col_side = x0 == 0 ? 1 : (x0 + TB == a.W ? 2 : 0);
        if (row_side) {
            khA = row_side == 1 ? KS - 1 : 0;
            khB = PAD == 2 ? (row_side == 1 ? KS - 2 : 1) : -1;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int r = ((wm * TM + i) * 32 + l31) >> 4;
                const int m = row_side == 1 ? r : 15 - r;
                if (m >= 1 && m <= PAD) {
                    dRA[i] = (PW + ring_slot(m, KS - 1) - (r + khA)) * PITCH;
                    if (PAD == 2 && m == 1) dRB[i] = (PW + ring_slot(m, KS - 2) - (r + khB)) * PITCH;
                }
            }
        }
        if (col_side) {
            kwA = col_side == 1 ? KS - 1 : 0;
            kwB = PAD == 2 ? (col_side == 1 ? KS - 2 : 1) : -1;
            const int x = l31 & 15;
            const int m = col_side == 1 ? x : 15 - x;
            if (m >= 1 && m <= PAD) {
                dCA = (PW + ring_slot(m, KS - 1) - (x + kwA)) * CS;
                if (PAD == 2 && m == 1) dCB = (PW + ring_slot(m, KS - 2) - (x + kwB)) * CS;
            }
        }
    }
    const bool ring_tile = RING && (row_side | col_side);
    // the pre-summed rows / columns of the slab whose raw fp32 patch sits in sR (every wave's DMA has landed and passed a barrier).
    // One item = 4 channels of one extra pixel: rows {r1[, r2]} x columns {c1[, c2]} of the raw patch added in fp32, split, stored.
    auto ring_presum = [&]() {
        if constexpr (RING) {
            const int ncol = col_side ? NALT * PW : 0;                        // extra columns, real rows
            const int wrow = PW + (col_side ? NALT : 0);                      // pixels of one extra row
            const int items = (ncol + (row_side ? NALT * wrow : 0)) * 4;
            auto pair = [&](int sl, int side, int& u, int& v, int base) {   // slot -> the two source rows / columns of block row / column `base`
                (void)base;
                if constexpr (S2 == 2) {
                    u = side == 1 ? 1 : 15;
                    v = side == 1 ? 0 : 16;
                    return;
                }
                int m = 1, kl = PAD + 1 + sl;
                if (PAD == 2 && sl == 2) m = 2, kl = 4;
                const int k = side == 1 ? kl : KS - 1 - kl;
                u = side == 1 ? m + k : 15 - m + k;
                v = side == 1 ? k - m : 15 + m + k;
            };
            bf16* dst = sP;
            for (int it = t; it < items; it += THREADS) {
                const int q = it & 3, px_ = it >> 2;
                int ty, tx, r1, r2, c1, c2;
                if (px_ < ncol) {                                             // extra column sl, real row ty
                    const int sl = px_ / PW;
                    ty = px_ - sl * PW;
                    tx = PW + sl;
                    r1 = r2 = ty;
                    pair(sl, col_side, c1, c2, 0);
                } else {                                                      // extra row sl, column tx (real, or an extra column)
                    const int e = px_ - ncol, sl = e / wrow;
                    tx = e - sl * wrow;
                    ty = PW + sl;
                    pair(sl, row_side, r1, r2, 0);
                    if (tx < PW) c1 = c2 = tx;
                    else pair(tx - PW, col_side, c1, c2, 0);
                }
                auto raw = [&](int r, int c) {
                    const int pp = r * PW + c;
                    const int ps = pp / PPT, tt = (pp - ps * PPT) * 4 + q;
                    return *reinterpret_cast<const f32x4*>(sR + (ps * THREADS + tt) * 8);
                };
                f32x4 v = raw(r1, c1);
                if (c2 != c1) v += raw(r1, c2);
                if (r2 != r1) {
                    v += raw(r2, c1);
                    if (c2 != c1) v += raw(r2, c2);
                }
                const Planes<NPL> pl2 = split_planes<NPL>(v, sx.s);
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<u32x2*>(dst + pl * P_PLANE + ty * PITCH + tx * CS + q * 4) = pl2.p[pl];
            }
        }
    };

    // acc: the leading products a0*b0; lo: the five correction products (2^-8 and 2^-16 of the leading one).  Kept apart,
    // the corrections are rounded at THEIR magnitude and the main accumulator sees one rounding per 16-channel step instead of
    // six; merged once in the epilogue.  (The 4x2-tile instantiation has no registers for a second set and adds all six in
    // place: measured error 2.7e-6 of the output scale on 6400-term sums against 6e-7 split -- both inside the 2e-5 fp32
    // tolerance of the parity suite, only the split form is used by default.)
    constexpr bool SPLIT = NPL == 3 && TM * TN <= 4;      // (two planes: the three products have one weight and share the accumulator)
    f32x16 acc[TM][TN], lo[SPLIT ? TM : 1][SPLIT ? TN : 1];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                acc[i][j][r] = 0.f;
                if (SPLIT) lo[SPLIT ? i : 0][SPLIT ? j : 0][r] = 0.f;
            }

    f32x16 big[FLUSH ? TM : 1][FLUSH ? TN : 1];
    if constexpr (FLUSH != 0) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) big[i][j][r] = 0.f;
    }

    // ---- prologue ----------------------------------------------------------------------------------------------------------
    load_patch(0);
    stage_w(0, 0);
    if (nsteps > 1) stage_w(1, 1);
    if constexpr (RAW) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the patch is read back from LDS
    write_patch(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (ring_tile) {                                    // (every wave's raw patch has landed: the barrier above)
        ring_presum();
        __syncthreads();
    }

    // ---- main loop: one step = one filter tap of one 16-channel slab, one barrier per step -----------------------------------
    // (r02 ablation of the unpipelined loop: kernel time = MFMA time + everything else, no overlap at all; r05 ablation of the
    // two-plane form, profiles/r05_h2_ablation.txt: still close to additive -- 5x5 256->128 at B=16: 273 us = ~131 us of loop without
    // MFMAs + 129 us of MFMAs at the nominal clock + ~13 us of launch, prologue and stores.)
    // PIPE: the barrier sits after two thirds of a step's MFMAs, right behind it the NEXT step's fragments are requested into the
    // second fragment set, and the last third of the MFMAs runs while they arrive (-3..8 % per kernel against reading them at the
    // top of their own step).  Requesting them a whole step early instead -- at the top of step s for step s + 1, or behind its
    // first MFMA group, with a four-slot ring filled three taps ahead so that the weights are in LDS one barrier earlier -- was built
    // and measured 3-6 % SLOWER per kernel (r05, ratios against the three-plane kernel of the same run: 1.59/1.62/1.52 against
    // 1.63/1.72/1.57); the four-slot ring alone made no difference (same box: 182/275/279 against 181/276/280 us).  Not kept.
    bf16x8 fa[PIPE ? 2 : 1][NPL][TM], fb[PIPE ? 2 : 1][NPL][TN];
    auto read_frags = [&](auto setc, int tap_, int pbuf_, int slot_) {
        constexpr int set = decltype(setc)::value;
        const int kh_ = tap_ / KS, kw_ = tap_ - kh_ * KS;
        const int d = kh_ * PITCH + kw_ * CS;
        const bf16* p = sP + pbuf_ * NPL * P_PLANE + d;
        const bf16* w = sW + slot_ * W_SLOT + b_row;
        int ring_d[TM];                                 // RING: ring pixels' lanes read their pre-summed pixel under this tap
#pragma unroll
        for (int i = 0; i < TM; ++i) ring_d[i] = 0;
        if constexpr (RING) {
            if (ring_tile) {
                const int dc = kw_ == kwA ? dCA : (kw_ == kwB ? dCB : 0);
#pragma unroll
                for (int i = 0; i < TM; ++i) ring_d[i] = dc + (kh_ == khA ? dRA[i] : (kh_ == khB ? dRB[i] : 0));
            }
        }
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
            for (int n = 0; n < TN; ++n) {
                if (DBG & 2) { for (int e = 0; e < 8; ++e) fb[set][pl][n][e] = (bf16)(float)(lane + tap_ + pl); }
                else fb[set][pl][n] = *reinterpret_cast<const bf16x8*>(w + (pl * BN + n * 32) * CS);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (DBG & 2) { for (int e = 0; e < 8; ++e) fa[set][pl][i][e] = (bf16)(float)(lane + i + pl); }
                else fa[set][pl][i] = *reinterpret_cast<const bf16x8*>(p + pl * P_PLANE + pp0[i] + ring_d[i]);
            }
        }
    };
    // terms [t0, t1) of the six products.  Split accumulators: in the order their fragments arrive from LDS (planes are read
    // 0, 1, 2), the leading product first -- its accumulator is separate, so the order costs no accuracy and the first MFMAs do
    // not wait for the last reads.  One accumulator (4x2 tiles): smallest terms first.
    auto mfma_terms = [&](auto setc, auto t0c, auto t1c) {
        constexpr int set = decltype(setc)::value, t0 = decltype(t0c)::value, t1 = decltype(t1c)::value;
        // (two planes: hi*hi, then lo*hi and hi*lo into the correction accumulator)
        constexpr int TA[6] = {NPL == 2 ? 0 : (SPLIT ? 0 : 2), 1, 0, SPLIT ? 2 : 1, SPLIT ? 1 : 0, 0};
        constexpr int TBp[6] = {0, NPL == 2 ? 0 : (SPLIT ? 0 : 1), NPL == 2 ? 1 : (SPLIT ? 1 : 2), 0, 1, SPLIT ? 2 : 0};
#pragma unroll
        for (int term = t0; term < t1; ++term)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int n = 0; n < TN; ++n) {
                    if (DBG & 1) {
                        asm volatile("" ::"v"(fa[set][TA[term]][i]), "v"(fb[set][TBp[term]][n]));
                        continue;
                    }
                    if (SPLIT && (TA[term] | TBp[term]) != 0)
                        lo[SPLIT ? i : 0][SPLIT ? n : 0] =
                            x3_mfma<NPL>(fb[set][TBp[term]][n], fa[set][TA[term]][i], lo[SPLIT ? i : 0][SPLIT ? n : 0]);
                    else
                        acc[i][n] = x3_mfma<NPL>(fb[set][TBp[term]][n], fa[set][TA[term]][i], acc[i][n]);
                }
    };
    typedef std::integral_constant<int, 0> I0;
    typedef std::integral_constant<int, 1> I1;
    // the products of a step in three groups (six products: 2 + 2 + 2, three: 1 + 1 + 1)
    typedef std::integral_constant<int, NPL == 3 ? 2 : 1> I2;
    typedef std::integral_constant<int, NPL == 3 ? 4 : 2> I4;
    typedef std::integral_constant<int, NPL == 3 ? 6 : 3> I6;
    constexpr float LO_W = 1.f;

    int pbuf = 0, tap = 0, cs = 0, slot = 0;
    auto step = [&](auto curc, auto nxtc, int s) {
        const bool fetch = tap == 0 && cs + 1 < ncs && !(DBG & 16) && !(DBG & 128);      // next slab's patch: registers now, LDS at tap 3
        if (fetch) load_patch(cs + 1);
        if constexpr (!PIPE) read_frags(curc, tap, pbuf, slot);
        mfma_terms(curc, I0{}, I2{});
        // (the compiler waits for the patch registers with a vmcnt that also covers every younger load: convert them BEFORE
        // this step's weight slab is issued, so that wait only sees loads that are at least a step old)
        // (waves w and w+4 share a SIMD: they convert one tap apart, so one of them is always free to feed the matrix pipe)
        if (PB == 2 && tap == 3 + (wave >> 2) && cs + 1 < ncs && !(DBG & 16)) write_patch(pbuf ^ 1);
        // AHEAD taps ahead, into the slot every wave left before the previous barrier
        if (s + AHEAD < nsteps && !(DBG & 4)) stage_w(s + AHEAD, slot >= 1 ? slot - 1 : NRING - 1);
        mfma_terms(curc, I2{}, I4{});
        if constexpr (!PIPE) mfma_terms(curc, I4{}, I6{});
        // the weights of step s + AHEAD - 1 (issued one step ago) must have landed; what this step issued (its weight slab and, at
        // tap 0, the PPASS loads of the next patch) may stay in flight
        if (s + AHEAD < nsteps) {
            if (fetch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W_INSTR + PPASS) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(W_INSTR) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        // raw barrier: __syncthreads() would add a fence that drains vmcnt to 0, i.e. wait for the slab issued a moment ago
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // this wave's patch writes and fragment reads are done
        if (!(DBG & 8)) __builtin_amdgcn_s_barrier();
        if (++tap == NTAP) {
            tap = 0;
            ++cs;
            if (PB == 2) {
                pbuf ^= 1;
            } else if (cs < ncs && !(DBG & 16)) {
                // one patch buffer: every wave is past its last read of the old slab (barrier above) -- convert in place
                // (RAW: the patch's DMA loads are NTAP steps old, the per-step vmcnt waits above have seen them land)
                if constexpr (FLUSH != 0 && !(DBG & 64)) {
                    if (cs % FLUSH == 0) {
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int n = 0; n < TN; ++n) {
                                big[i][n] += acc[i][n];
#pragma unroll
                                for (int r = 0; r < 16; ++r) acc[i][n][r] = 0.f;
                            }
                    }
                }
                if (!(DBG & 32)) write_patch(0);
                if (ring_tile) ring_presum();         // (reads the raw patch only: needs no barrier behind write_patch)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
        slot = slot == NRING - 1 ? 0 : slot + 1;
        if constexpr (PIPE) {
            if (s + 1 < nsteps) read_frags(nxtc, tap, pbuf, slot);
            mfma_terms(curc, I4{}, I6{});
        }
    };
    if constexpr (PIPE) {
        read_frags(I0{}, 0, 0, 0);
        for (int s = 0; s < nsteps; s += 2) {
            step(I0{}, I1{}, s);
            if (s + 1 < nsteps) step(I1{}, I0{}, s + 1);
        }
    } else {
        for (int s = 0; s < nsteps; ++s) step(I0{}, I0{}, s);
    }

    // ---- epilogue: bias + activation, fp32 stores of 4 channels per lane ---------------------------------------------------
    // (r04: the bias vectors of this lane's columns are loaded ONCE, in one batch, and the `add` operand of a pixel tile in one
    // batch in front of its stores.  Inside the per-chunk conditionals -- `if (col < N) if (bias) v += bias[col]` -- the compiler
    // can neither hoist nor batch a load: every chunk paid its own L2 round trip, one after the other.)
    if constexpr (KSP == 2) {
        // (whole tiles of a mixed launch skip the exchange; the correction accumulators are merged here either way)
        __shared__ unsigned s_role;
        if constexpr (SPLIT || FLUSH != 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int n = 0; n < TN; ++n) {
                    if constexpr (SPLIT) acc[i][n] += lo[SPLIT ? i : 0][SPLIT ? n : 0] * LO_W;
                    if constexpr (FLUSH != 0) acc[i][n] += big[i][n];
                }
        }
        // Both workgroups sit on one XCD: its L2 is the meeting point.  The first arriver's stores are complete (acknowledged by the
        // L2) before it raises the ticket, the second reads the ticket and the half sum past its L1 (sc1) -- no L2 write-back or
        // invalidate, which an agent-scope release / acquire pair would cost every workgroup (measured: 145 against 117 us).
      if (halved) {
        // ---- the two halves of the contraction meet (see the template comment) ------------------------------------------------
        unsigned* ticket = a.tickets + (tile_id - a.split_from);
        // The hand-off below is only coherent inside ONE XCD's L2.  That the two halves share one rests on the dispatcher placing
        // hardware workgroup h on XCD h % 8; each half therefore reads the XCD it really runs on (XCC_ID), the first arriver
        // publishes its id in bits 8.. of the ticket and the second poisons the tile when the ids differ (a CU mask, another
        // partition mode or a dispatch change must not turn into a silently stale half sum).
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        xcc &= 0xfu;
        if (t == 0) s_role = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        const unsigned role = s_role;
        // [tile][i][n][q4][thread] f32x4: 16 bytes per lane, consecutive lanes consecutive
        constexpr unsigned TILE_B = TM * TN * 4 * THREADS * 16;
        const __amdgpu_buffer_rsrc_t rsrc_p = __builtin_amdgcn_make_buffer_rsrc(
            reinterpret_cast<char*>(a.part) + (size_t)(tile_id - a.split_from) * TILE_B, 0, TILE_B, 0x00020000);
        typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
        if (role == 0) {
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int n = 0; n < TN; ++n)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4)
                        __builtin_amdgcn_raw_buffer_store_b128(
                            __builtin_bit_cast(u32x4_t, f32x4{acc[i][n][4 * q4], acc[i][n][4 * q4 + 1], acc[i][n][4 * q4 + 2], acc[i][n][4 * q4 + 3]}),
                            rsrc_p, (unsigned)(((i * TN + n) * 4 + q4) * THREADS + t) * 16u, 0, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (t == 0) __hip_atomic_fetch_add(ticket, 2u + ((xcc + 1u) << 8), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
        // second arriver: the low byte of the ticket reads 1 (first) + 1 (this one) + 2 (first done) once the other half is published.  The
        // wait is bounded (an aborted earlier launch may have left the ticket dirty): on expiry the tile is poisoned with NaN.
        __shared__ unsigned s_ok;
        if (t == 0) {
            unsigned ok = 0;
            for (int spin = 0; spin < (1 << 22); ++spin) {
                const unsigned tv = __hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((tv & 0xffu) >= 4u) { ok = (tv >> 8) == xcc + 1u; break; }      // published -- by a workgroup of THIS XCD, or the tile is poisoned
                __builtin_amdgcn_s_sleep(1);
            }
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // (sticky: the caller polls it -- hipdwc.ops.ksplit_status_poll -- raises and re-zeroes the ticket row, which a late first
            // arriver of this tile may still dirty)
            if (!ok && a.status) __hip_atomic_fetch_or(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_ok = ok;
        }
        __syncthreads();
        const float poison = s_ok ? 0.f : __builtin_nanf("");
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            f32x4 ov[TN][4];
#pragma unroll
            for (int n = 0; n < TN; ++n)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)       // aux 16 = sc1: served by the L2, never by this CU's L1
                    ov[n][q4] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                              rsrc_p, (unsigned)(((i * TN + n) * 4 + q4) * THREADS + t) * 16u, 0, 16));
#pragma unroll
            for (int n = 0; n < TN; ++n)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[i][n][4 * q4 + k] += ov[n][q4][k] + poison;
        }
      }
    }
    if constexpr (KSP == 1 && FLUSH != 0) {
        // (merged before the bias / add vectors are loaded: three accumulator sets and those would not fit the register file)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int n = 0; n < TN; ++n) acc[i][n] += big[i][n];
    }
    const float slope = dwc_act_slope(a.act);
    f32x4 bv[TN][4];
#pragma unroll
    for (int n = 0; n < TN; ++n)
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) bv[n][q4] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.bias) {
#pragma unroll
        for (int n = 0; n < TN; ++n)
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
                bv[n][q4] = *reinterpret_cast<const f32x4*>(a.bias + min(n0 + (wn * TN + n) * 32 + 8 * q4 + 4 * hi, a.N - 4));
    }
    unsigned y_am = 0;
    auto store = [&](auto general) {
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int pb = (wm * TM + i) * 32 + l31;
            float* dst = a.y + ((size_t)(n_img * OH + y0 + (pb >> 4)) * OW + x0 + (pb & 15)) * a.N;
            if constexpr (S2 == 2)
                dst = a.y + ((size_t)(n_img * 2 * OH + 2 * (y0 + (pb >> 4)) + ry) * (2 * OW) + 2 * (x0 + (pb & 15)) + rx) * a.N;
            f32x4 adv[TN][4];
            if (a.add) {
#pragma unroll
                for (int n = 0; n < TN; ++n)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4)
                        adv[n][q4] = *reinterpret_cast<const f32x4*>(a.add + (dst - a.y) + min(n0 + (wn * TN + n) * 32 + 8 * q4 + 4 * hi, a.N - 4));
            }
#pragma unroll
            for (int n = 0; n < TN; ++n)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const int col = n0 + (wn * TN + n) * 32 + 8 * q4 + 4 * hi;
                    if (col >= a.N) continue;
                    f32x4 v = {acc[i][n][4 * q4], acc[i][n][4 * q4 + 1], acc[i][n][4 * q4 + 2], acc[i][n][4 * q4 + 3]};
                    if (SPLIT && KSP == 1 && FLUSH == 0) {
                        const f32x16& c = lo[SPLIT ? i : 0][SPLIT ? n : 0];
                        v += f32x4{c[4 * q4], c[4 * q4 + 1], c[4 * q4 + 2], c[4 * q4 + 3]} * LO_W;
                    }
                    if constexpr (NPL == 2) v = v * sx.inv * sw.inv;      // (two exact multiplications: the product of the two could underflow)
                    v += bv[n][q4];
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        if constexpr (decltype(general)::value) v[k] = dwc_act_apply(v[k], a.act, col + k);
                        else v[k] = dwc_act_simple(v[k], slope);
                    }
                    if (a.add) v += adv[n][q4];
                    if constexpr (NPL == 2) y_am = max(max(y_am, max(dwc_abs_bits(v[0]), dwc_abs_bits(v[1]))), max(dwc_abs_bits(v[2]), dwc_abs_bits(v[3])));
                    if (!(DBG & 256)) *reinterpret_cast<f32x4*>(dst + col) = v;
                    else asm volatile("" ::"v"(v));
                }
        }
    };
    if (dwc_act_is_simple(a.act)) store(std::false_type{});
    else store(std::true_type{});
    if constexpr (NPL == 2) {
        __shared__ unsigned s_am[NW];
        dwc_amax_block_publish(a.ys, a.ys_epoch, y_am, s_am);      // (a.ys is uniform over the launch; a split tile's first arriver has left)
    }
#endif
}

// w: [Cout][Cin][K][K] fp32 (OIHW).  forward: rows = output channels, contraction over input channels;
// dgrad: rows = input channels, contraction over output channels, filter rotated by 180 degrees.
// out[((tap*ncs + cs)*3 + plane)*rows + row][16]
__global__ void x3_weight_prepare_kernel(const float* __restrict__ w, bf16* __restrict__ out, int Cout, int Cin, int K, int rows,
                                         int kdim, int dgrad) {
    const int ncs = (kdim + CS - 1) / CS;
    const size_t total = (size_t)K * K * ncs * rows * CS;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int j = idx % CS;
    size_t r = idx / CS;
    const int row = r % rows;
    r /= rows;
    const int cs = r % ncs, tap = r / ncs;
    const int kh = tap / K, kw = tap - kh * K;
    const int kc = cs * CS + j;
    float v = 0.f;
    if (!dgrad) {
        if (row < Cout && kc < Cin) v = w[(((size_t)row * Cin + kc) * K + kh) * K + kw];
    } else {
        if (row < Cin && kc < Cout) v = w[(((size_t)kc * Cin + row) * K + (K - 1 - kh)) * K + (K - 1 - kw)];
    }
    const unsigned hb = __float_as_uint(v) & 0xffff0000u;
    const float r1 = v - __uint_as_float(hb);
    const unsigned mb = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mb);
    unsigned short* o = reinterpret_cast<unsigned short*>(out);
    // halves of a row swapped when (row>>3)&1: bank-conflict-free fragment reads of the linear LDS copy (see the kernel)
    const size_t base = ((size_t)(tap * ncs + cs) * 3 * rows + row) * CS + (j ^ (((row >> 3) & 1) << 3));
    o[base] = (unsigned short)(hb >> 16);
    o[base + (size_t)rows * CS] = (unsigned short)(mb >> 16);
    o[base + 2 * (size_t)rows * CS] = (unsigned short)(__float_as_uint(r2) >> 16);
}

// Two-plane form of the same layout: out[((tap*ncs + cs)*2 + plane)*rows + row][16] f16 planes of s_w * w, s_w from the weight's
// absmax slot; {s_w, 1 / s_w} as two floats behind the planes (the convolution kernels read them from there).
__global__ void h2_weight_prepare_kernel(const float* __restrict__ w, unsigned short* __restrict__ out, int Cout, int Cin, int K,
                                         int rows, int kdim, int dgrad, const unsigned long long* slot, unsigned epoch) {
    const int ncs = (kdim + CS - 1) / CS;
    const size_t total = (size_t)K * K * ncs * rows * CS;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const H2Scale sw = h2_scale(slot, epoch);
    if (idx == 0) {
        float* tail = reinterpret_cast<float*>(out + 2 * total);
        tail[0] = sw.s;
        tail[1] = sw.inv;
    }
    if (idx >= total) return;
    const int j = idx % CS;
    size_t r = idx / CS;
    const int row = r % rows;
    r /= rows;
    const int cs = r % ncs, tap = r / ncs;
    const int kh = tap / K, kw = tap - kh * K;
    const int kc = cs * CS + j;
    float v = 0.f;
    if (!dgrad) {
        if (row < Cout && kc < Cin) v = w[(((size_t)row * Cin + kc) * K + kh) * K + kw];
    } else {
        if (row < Cin && kc < Cout) v = w[(((size_t)kc * Cin + row) * K + (K - 1 - kh)) * K + (K - 1 - kw)];
    }
    v *= sw.s;
    const _Float16 h = (_Float16)v;
    const _Float16 l = (_Float16)(v - (float)h);
    const size_t base = ((size_t)(tap * ncs + cs) * 2 * rows + row) * CS + (j ^ (((row >> 3) & 1) << 3));
    out[base] = __builtin_bit_cast(unsigned short, h);
    out[base + (size_t)rows * CS] = __builtin_bit_cast(unsigned short, l);
}

// max |x| over n floats -> slot (see dwc_amax_publish)
__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, size_t n, unsigned long long* slot, unsigned epoch) {
    const size_t n4 = n >> 2;
    unsigned m = 0;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x);
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {      // four 16-byte loads in flight per thread
        const f32x4 a = x4[i], b = x4[i + stride], c = x4[i + 2 * stride], d = x4[i + 3 * stride];
#pragma unroll
        for (int k = 0; k < 4; ++k) m = max(max(m, dwc_abs_bits(a[k])), max(dwc_abs_bits(b[k]), max(dwc_abs_bits(c[k]), dwc_abs_bits(d[k]))));
    }
    for (; i < n4; i += stride) {
        const f32x4 a = x4[i];
#pragma unroll
        for (int k = 0; k < 4; ++k) m = max(m, dwc_abs_bits(a[k]));
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) m = max(m, dwc_abs_bits(x[(n4 << 2) + threadIdx.x]));
    m = dwc_wave_max_u32(m);
    __shared__ unsigned sm[4];
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) dwc_amax_publish(slot, epoch, max(max(sm[0], sm[1]), max(sm[2], sm[3])));
}

// ------------------------------------------------------------------------------------------
// Weight gradient of the same layers as split products.  dW[kh][kw][ci][co] = sum_pixels x[pixel + tap][ci] * dY[pixel][co]:
// both operands are activations, so both are split on the fly.  A workgroup owns 64 channels of x, BN channels of dY and ONE
// filter column kw (all K rows kh), and walks 8x16-pixel units (wgrad_halo_kernel of conv_halo_bf16.hip is the bf16 form):
//  * the x patch (8+K-1 rows x 16 pixels, already shifted by kw) goes global -> registers -> three bf16 planes in LDS
//    ([pixel][64 ch], 128-byte rows, chunk swizzle of the transposing read), double buffered, written two passes at a time
//    between the rows of the previous unit;
//  * dY never enters LDS: a lane reads the 8 pixels x 1 channel of its B-fragment slot straight from global memory (128
//    contiguous bytes per pixel across 32 lanes), one row ahead, and splits them in registers;
//  * per dY row r and filter row kh the three A planes of patch row r + kh come through ds_read_b64_tr_b16 and meet the three
//    B planes in the six leading products; the leading product and the five corrections have separate accumulators.
// Pixel ranges are split over gridDim.y into fp32 slabs [split][tap*Cin + ci][co] summed in fixed order by x3_wgrad_reduce.
// HALVES == 2 (BN == 64): the two wave quartets take the upper / lower four rows of every unit and write their own slabs.
// ------------------------------------------------------------------------------------------
struct X3WgradArgs {
    const float* x;      // [B][H][W][Cin]
    const float* dy;     // [B][H][W][N]   (KS == 4, the stride-2 form: [B][H/2][W/2][N])
    float* slab;         // [splits * HALVES][K*K*Cin][N]
    int B, H, W, Cin, N;
    int units_x, units_per_img, total_units, units_per_split;
    int n_tiles, roles;
    const unsigned long long *xs = nullptr, *dys = nullptr;      // NPL == 2: absmax slots of x and dy and the epochs they must carry
    unsigned xs_epoch = 0, dys_epoch = 0;
};

// CIW = 64: 8 waves, one workgroup per CU.  CIW = 32: 4 waves, <= 74 KB of LDS, TWO independent workgroups per CU (see
// conv_halo_x3_kernel: they drift out of phase and overlap each other's MFMA and load phases).
template <int KS, int BN, int CIW = 64, int AHEAD = 2, int NPL = 3>
// (two planes, K = 3 / 4: two 8-wave workgroups per CU -- four waves per SIMD, at most 128 registers -- see x3_wgrad_plan)
__global__ __launch_bounds__(CIW == 64 ? 512 : 256, CIW == 64 ? (NPL == 2 && KS != 5 ? 4 : 1) : 2) void wgrad_x3_kernel(X3WgradArgs a) {
#if defined(__HIP_DEVICE_COMPILE__)
    static_assert((BN == 128 || BN == 64) && (CIW == 64 || CIW == 32), "tile shape");
    constexpr int HALVES = BN == 128 ? 1 : 2;
    constexpr int QPP = CIW / 4;                       // gather threads per pixel (4 channels each); 32 pixels per pass either way
    // KS == 4: the 4x4 stride-2 reflect-pad-1 layers (reference networks.py:90,94,437, networks_v2.py:107-111).  Tap (kh, kw) of
    // output pixel (oy, ox) reads input pixel (2 oy + kh - 1, 2 ox + kw - 1): for the workgroup's filter column kw the "patch" is
    // the stride-2 column set 2 (x0 + c) + kw - 1 of input rows 2 y0 - 1 .. 2 y0 + 2 UH, and (output row ks, filter row kh) reads
    // patch row 2 ks + kh.  Units are 4 rows high (10 patch rows: the LDS holds two buffers of three planes).
    constexpr bool S2 = KS == 4;
    constexpr int UH = S2 ? 4 : 8, UW = 16;
    constexpr int PH = S2 ? 2 * UH + 2 : UH + KS - 1, PPIX = PH * UW;
    constexpr int PPASS = (PPIX + 31) / 32;            // gather passes of 32 pixels (16 threads x 4 channels per pixel)
    constexpr int P_PLANE = PPASS * 32 * CIW;          // elements per plane
    constexpr int PAD = (KS - 1) / 2;
    constexpr int ROWS = UH / HALVES;                  // dY rows of a unit per wave
    __shared__ __attribute__((aligned(16))) bf16 smem[2 * NPL * P_PLANE];
    H2Scale sx = {1.f, 1.f}, sdy = {1.f, 1.f};
    if constexpr (NPL == 2) {
        sx = h2_scale(a.xs, a.xs_epoch);
        sdy = h2_scale(a.dys, a.dys_epoch);
    }

    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // 1-D grid, XCD-aware: hardware workgroup i runs on XCD i % 8; virtual ids are laid out so that each XCD gets a contiguous
    // range, and the roles (ci slab, dY tile, filter column) of one pixel split are consecutive virtual ids -- the workgroups
    // that re-read the same pixels of x and dY share an L2 (r02 PMC: 2.3 GB of HBM-side traffic per launch before this)
    int id = blockIdx.x;
    {
        const int nb = gridDim.x;
        if (nb >= 16) {
            const int q = nb >> 3, r = nb & 7, x = id & 7, y = id >> 3;
            id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
        }
    }
    const int split = id / a.roles;
    id -= split * a.roles;
    const int kw = id % KS;
    id /= KS;
    const int tn = id % a.n_tiles, cs = id / a.n_tiles;
    const int u0 = split * a.units_per_split, u1 = min(a.total_units, u0 + a.units_per_split);
    const int ci_tile = CIW == 32 ? 0 : (BN == 128 ? (wave >> 2) : (wave & 1));
    const int co_tile = CIW == 32 ? (BN == 128 ? wave : (wave & 1)) : (BN == 128 ? (wave & 3) : ((wave >> 1) & 1));
    const int half_id = BN == 128 ? 0 : (CIW == 32 ? (wave >> 1) : (wave >> 2));
    const int l31 = lane & 31, hi = lane >> 5;
    const int OH = S2 ? a.H >> 1 : a.H, OW = S2 ? a.W >> 1 : a.W;     // the dY grid (units tile it)

    // ---- unit walk ---------------------------------------------------------------------------------------------------------
    // (r04) Position (image, unit row, unit column) of the current and the next unit, advanced incrementally: wave-uniform, scalar
    // ALU only.  The gathers are buffer loads -- a 32-bit per-lane offset that is constant over the kernel (dY) or needs one
    // multiply-add per pass (x) plus a scalar offset for everything that moves with the unit.  Before that every dY row paid two
    // integer divisions and eight 64-bit address chains, every fragment read its swizzle: 4.7 vector instructions per MFMA.
    const int units_y = a.units_per_img / a.units_x;
    struct UnitPos { int n, uy, ux; };
    auto unit_pos = [&](int u) {
        UnitPos q;
        q.n = u / a.units_per_img;
        const int ur = u - q.n * a.units_per_img;
        q.uy = ur / a.units_x;
        q.ux = ur - q.uy * a.units_x;
        return q;
    };
    auto unit_next = [&](UnitPos q) {
        if (++q.ux == a.units_x) {
            q.ux = 0;
            if (++q.uy == units_y) q.uy = 0, ++q.n;
        }
        return q;
    };

    // ---- x patch: thread = (pixel t / QPP [+32 per pass], channel quad t % QPP); pass p covers patch rows 2p, 2p+1 ------------
    const int quad = t % QPP;
    // chunk swizzle of the transposing read: 128-byte rows need it, 64-byte rows (CIW 32) place 4 consecutive pixels on the
    // four quarters of the 256-byte bank window by themselves
    auto p_swz = [](int pp) { return CIW == 64 ? 4 * ((pp >> 1) & 1) : 0; };
    const __amdgpu_buffer_rsrc_t rsrc_x =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.x), 0, (unsigned)((size_t)a.B * a.H * a.W * a.Cin * 4), 0x00020000);
    const int prow = (t / QPP) >> 4, pcol = (t / QPP) & 15;
    const unsigned x_lane = (unsigned)(cs * CIW + quad * 4) * 4u;
    const unsigned img_bytes = (unsigned)a.H * a.W * a.Cin * 4u;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    f32x4 pv[2];
    auto load_patch = [&](UnitPos q, int first) {      // passes first, first+1 of the unit at q into pv[]
        const int y0 = q.uy * UH, x0 = q.ux * UW;
        const int w = min(reflect_idx(S2 ? 2 * (x0 + pcol) + kw - 1 : x0 - PAD + kw + pcol, a.W), a.W - 1);
        const unsigned img = (unsigned)q.n * img_bytes;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int h = min(reflect_idx((S2 ? 2 * y0 - 1 : y0 - PAD) + prow + 2 * (first + j), a.H), a.H - 1);
            const unsigned off = (unsigned)((h * a.W + w) * a.Cin) * 4u + x_lane;
            pv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, off, img, 0));
        }
    };
    auto write_patch = [&](int buf, int first) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int pp = t / QPP + 32 * (first + j);
            if (first + j >= PPASS) continue;
            const Planes<NPL> q = split_planes<NPL>(pv[j], sx.s);
            bf16* dst = smem + buf * NPL * P_PLANE + pp * CIW + (((quad >> 1) ^ p_swz(pp)) << 3) + (quad & 1) * 4;
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<u32x2*>(dst + pl * P_PLANE) = q.p[pl];
        }
    };

    // ---- dY fragments straight from global memory: lane (co = l31, pixels 8*hi .. 8*hi+7 of the row), AHEAD rows in flight ----
    const __amdgpu_buffer_rsrc_t rsrc_dy =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.dy), 0, (unsigned)((size_t)a.B * OH * OW * a.N * 4), 0x00020000);
    const unsigned n4 = (unsigned)a.N * 4u;
    const unsigned dy_lane = (unsigned)(8 * hi * a.N + tn * BN + co_tile * 32 + l31) * 4u;
    float raw[AHEAD][8];
    auto load_dy = [&](auto slot, UnitPos q, int row) {
        const unsigned base = (unsigned)((q.n * OH + q.uy * UH + row) * OW + q.ux * UW) * n4;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            raw[decltype(slot)::value][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_dy, dy_lane, base + j * n4, 0));
    };
    bf16x8 fb[NPL];
    auto split_dy = [&](auto slot) {
        if constexpr (NPL == 2) {
            u32x4 q0, q1;
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                u32x2 p0, p1;
                split2h(f32x4{raw[decltype(slot)::value][4 * k], raw[decltype(slot)::value][4 * k + 1], raw[decltype(slot)::value][4 * k + 2],
                              raw[decltype(slot)::value][4 * k + 3]}, sdy.s, p0, p1);
                q0[2 * k] = p0[0]; q0[2 * k + 1] = p0[1];
                q1[2 * k] = p1[0]; q1[2 * k + 1] = p1[1];
            }
            fb[0] = __builtin_bit_cast(bf16x8, q0);
            fb[1] = __builtin_bit_cast(bf16x8, q1);
            return;
        } else {
        // two values per instruction where the ISA has one: v_pk_add_f32 for the two subtractions (36 instead of 44 per row)
        u32x4 q0, q1, q2;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x2 v = {raw[decltype(slot)::value][2 * k], raw[decltype(slot)::value][2 * k + 1]};
            const f32x2 hb = {__uint_as_float(__float_as_uint(v[0]) & 0xffff0000u), __uint_as_float(__float_as_uint(v[1]) & 0xffff0000u)};
            const f32x2 r = v - hb;
            const f32x2 mb = {__uint_as_float(__float_as_uint(r[0]) & 0xffff0000u), __uint_as_float(__float_as_uint(r[1]) & 0xffff0000u)};
            const f32x2 l = r - mb;
            q0[k] = __builtin_amdgcn_perm(__float_as_uint(v[1]), __float_as_uint(v[0]), 0x07060302u);      // high halves: v's = hb's
            q1[k] = __builtin_amdgcn_perm(__float_as_uint(r[1]), __float_as_uint(r[0]), 0x07060302u);
            q2[k] = __builtin_amdgcn_perm(__float_as_uint(l[1]), __float_as_uint(l[0]), 0x07060302u);
        }
        fb[0] = __builtin_bit_cast(bf16x8, q0);
        fb[1] = __builtin_bit_cast(bf16x8, q1);
        fb[NPL - 1] = __builtin_bit_cast(bf16x8, q2);
        }
    };

    // ---- x fragments: transposing reads, lane 4q+p of a 16-lane group addresses pixel q, channels 4p..4p+3 ------------------
    // (the chunk swizzle of pixel r*16 + pxl + 4*hf depends on bit 1 of pxl only: one lane offset, the rest are immediates)
    const int li = lane & 15, gam = (lane >> 4) & 1;
    const int tq = li >> 2, tp = li & 3;
    const int pxl = 8 * hi + tq;
    const int a_col = ci_tile * 32 + 16 * gam + 4 * tp;
    const int a_lane = pxl * CIW + (((a_col >> 3) ^ p_swz(pxl)) << 3) + (a_col & 7);
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    typedef __attribute__((address_space(3))) bf16x4* lds4;
    auto a_frag = [&](const bf16* plane_lane, int r) {   // plane_lane = plane + a_lane
        bf16x4 v[2];
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) v[hf] = __builtin_amdgcn_ds_read_tr16_b64_v4bf16((lds4)(plane_lane + (r * UW + 4 * hf) * CIW));
        return __builtin_shufflevector(v[0], v[1], 0, 1, 2, 3, 4, 5, 6, 7);
    };

    f32x16 acc[KS], lo[NPL == 3 ? KS : 1];
#pragma unroll
    for (int j = 0; j < KS; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            acc[j][r] = 0.f;
            if (NPL == 3) lo[NPL == 3 ? j : 0][r] = 0.f;
        }

    typedef std::integral_constant<int, 0> S0;
    typedef std::integral_constant<int, AHEAD - 1> S1;
    static_assert(AHEAD == 1 || (AHEAD == 2 && ROWS % 2 == 0), "dY rows in flight");
    if (u0 < u1) {
        UnitPos cur = unit_pos(u0), nxt = unit_next(cur);
        // first unit's patch
        for (int f = 0; f < PPASS; f += 2) {
            load_patch(cur, f);
            write_patch(0, f);
        }
        load_dy(S0{}, cur, half_id * ROWS);
        if constexpr (AHEAD == 2) load_dy(S1{}, cur, half_id * ROWS + 1);
        __syncthreads();
        int buf = 0;
        for (int u = u0; u < u1; ++u) {
            const bool next = u + 1 < u1;
            const bf16* p = smem + buf * NPL * P_PLANE + a_lane;
#pragma unroll
            for (int rr = 0; rr < ROWS; ++rr) {
                const int ks = half_id * ROWS + rr;
                // next unit's patch: two gather passes in flight at a time, written two rows after they were issued
                constexpr int GROUPS = (PPASS + 1) / 2;
                constexpr int STEP = ROWS >= 2 * GROUPS + 2 ? 2 : 1;
                if (next) {
                    if (rr >= STEP && (rr - STEP) % STEP == 0 && (rr - STEP) / STEP < GROUPS) write_patch(buf ^ 1, 2 * ((rr - STEP) / STEP));
                    if (rr % STEP == 0 && rr / STEP < GROUPS) load_patch(nxt, 2 * (rr / STEP));
                }
                // this row's dY: split, then its registers take the row AHEAD rows further on (of the next unit at the end)
                auto refill = [&](auto slot) {
                    split_dy(slot);
                    if (rr + AHEAD < ROWS) load_dy(slot, cur, ks + AHEAD);
                    else if (next) load_dy(slot, nxt, half_id * ROWS + rr + AHEAD - ROWS);
                };
                if ((rr & (AHEAD - 1)) == 0) refill(S0{});
                else refill(S1{});
#pragma unroll
                for (int kh = 0; kh < KS; ++kh) {
                    bf16x8 fa[NPL];
#pragma unroll
                    for (int pl = 0; pl < NPL; ++pl) fa[pl] = a_frag(p + pl * P_PLANE, S2 ? 2 * ks + kh : ks + kh);
                    if constexpr (NPL == 3) {
                        lo[kh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[2], fb[0], lo[kh], 0, 0, 0);
                        lo[kh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[1], lo[kh], 0, 0, 0);
                        lo[kh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[NPL - 1], lo[kh], 0, 0, 0);
                        lo[kh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[1], fb[0], lo[kh], 0, 0, 0);
                        lo[kh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[1], lo[kh], 0, 0, 0);
                        acc[kh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[0], fb[0], acc[kh], 0, 0, 0);
                    } else {
                        acc[kh] = x3_mfma<2>(fa[1], fb[0], acc[kh]);      // (one weight, one accumulator: see split2h)
                        acc[kh] = x3_mfma<2>(fa[0], fb[1], acc[kh]);
                        acc[kh] = x3_mfma<2>(fa[0], fb[0], acc[kh]);
                    }
                }
            }
            if (next) {                                 // gather groups the row loop had no room for
                constexpr int GROUPS = (PPASS + 1) / 2;
                constexpr int STEP = ROWS >= 2 * GROUPS + 2 ? 2 : 1;
                constexpr int DONE_W = ROWS > STEP ? (ROWS - 1 - STEP) / STEP + 1 : 0;      // groups written inside the loop
                constexpr int DONE_L = (ROWS - 1) / STEP + 1;                                 // groups loaded inside the loop
#pragma unroll
                for (int gq = (DONE_W < GROUPS ? DONE_W : GROUPS); gq < GROUPS; ++gq) {
                    if (gq >= (DONE_L < GROUPS ? DONE_L : GROUPS)) load_patch(nxt, 2 * gq);
                    write_patch(buf ^ 1, 2 * gq);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // patch writes done; raw barrier: the next unit's first dY
            __builtin_amdgcn_s_barrier();                                // rows stay in flight (no vmcnt drain as in __syncthreads)
            buf ^= 1;
            cur = nxt;
            nxt = unit_next(nxt);
        }
    }
    // slab[split * HALVES + half][(tap*Cin + ci)][co]
    const int Ktot = KS * KS * a.Cin;
    float* out = a.slab + (size_t)(split * HALVES + half_id) * Ktot * a.N;
#pragma unroll
    for (int kh = 0; kh < KS; ++kh) {
        const int tap = kh * KS + kw;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = tap * a.Cin + cs * CIW + ci_tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            if constexpr (NPL == 2) out[(size_t)k * a.N + tn * BN + co_tile * 32 + l31] = acc[kh][r] * sx.inv * sdy.inv;
            else out[(size_t)k * a.N + tn * BN + co_tile * 32 + l31] = acc[kh][r] + lo[kh][r];
        }
    }
#endif
}

// slab[s][(kh,kw,ci)][co] summed over s -> dw[co][ci][kh][kw] (state_dict layout, fp32), real channels only
__global__ void x3_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int splits, int K, int N, int Cin, int KHW,
                                       int cin_real, int cout_real) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)K * N) return;
    const int co = idx % N;
    const int k = idx / N;
    const int ci = k % Cin, tap = k / Cin;
    if (co >= cout_real || ci >= cin_real) return;
    // (the slabs are added in order z = 0, 1, 2 ...; four loads in flight at a time -- one dependent load per slab made this launch
    // a chain of L2 latencies: 12 us for 24 MB)
    const float* p = slab + idx;
    const size_t stride = (size_t)K * N;
    float s = 0.f;
    int z = 0;
    for (; z + 4 <= splits; z += 4) {
        const float v0 = p[(size_t)z * stride], v1 = p[(size_t)(z + 1) * stride], v2 = p[(size_t)(z + 2) * stride],
                    v3 = p[(size_t)(z + 3) * stride];
        s += v0;
        s += v1;
        s += v2;
        s += v3;
    }
    for (; z < splits; ++z) s += p[(size_t)z * stride];
    dw[((size_t)co * cin_real + ci) * KHW + tap] = s;
}

// 64 channels of x per workgroup: one 8-wave workgroup per CU.  (The kernel's CIW = 32 form -- two 4-wave workgroups per CU -- was
// measured in r02-r04 at -3..+6 % by batch and is no longer instantiated.)
static int x3_wgrad_ciw() { return 64; }

// (K == 4 is the stride-2 form: H, W are the dimensions of x, the dY grid is H/2 x W/2 and is cut into 4x16-pixel units)
int x3_wgrad_bn(int B, int H, int W, int Cin, int Cout, int K) {
    if (K == 4) {
        if (B <= 0 || H < 8 || W < 32 || (H % 8) || (W % 32) || Cin < 64 || (Cin % 64)) return 0;
    } else if (B <= 0 || (K != 3 && K != 5) || H < 8 || W < 16 || (H % 8) || (W % 16) || Cin < 64 || (Cin % 64)) return 0;
    // (the kernel addresses x and dY through 32-bit buffer offsets)
    if ((size_t)B * H * W * Cin * 4 >= 0xffffffffull || (size_t)B * H * W * Cout * 4 >= 0xffffffffull) return 0;
    if (Cout >= 128 && !(Cout % 128)) return 128;
    return (Cout >= 64 && !(Cout % 64)) ? 64 : 0;
}

void x3_wgrad_plan(int B, int H, int W, int Cin, int Cout, int K, int bn, int npl, int* splits, int* ups) {
    const int ciw = x3_wgrad_ciw();
    const int roles = (Cin / ciw) * (Cout / bn) * K;
    const int units = K == 4 ? B * (H / 8) * (W / 32) : B * (H / 8) * (W / 16);
    // resident workgroups: whole rounds, see wgrad_halo_plan.  The two-plane kernels of K = 3 and K = 4 need 106-124 registers and
    // exactly half of the LDS (81,920 bytes): two workgroups per CU (r05, wgrad + reduce planned for 256 / 512 slots: 3x3 256->256
    // B=48 250 -> 223 us, 4x4 stride 2 200 -> 186 and 184 -> 179 us, B=16 unchanged; K = 5 holds 96 KB and stays at one per CU).
    // (... unless that leaves a workgroup fewer than 8 units -- 3x3 256->256 at B=16: the kernel takes the same time either way and
    // twice the slabs cost the reduce 13.5 instead of 9.9 us)
    const int cus = (npl == 2 && K != 5 && (long)units * roles >= 8L * 512) ? 512 : 256;
    const int smax = units / 4 > 0 ? units / 4 : 1;
    int s = 1;
    double best = 0.0;
    for (int c = 1; c <= smax && c * roles <= 2 * cus; ++c) {
        const int wgs = c * roles, rounds = (wgs + cus - 1) / cus;
        const double fill = (double)wgs / (rounds * cus) / (rounds > 1 ? 1.05 : 1.0);
        if (fill > best + 1e-9) best = fill, s = c;
    }
    *ups = (units + s - 1) / s;
    *splits = (units + *ups - 1) / *ups;
}

bool x3_ok(int B, int H, int W, int Cin, int N, int K) {
    return B > 0 && (K == 3 || K == 5) && H >= TB && W >= TB && !(H % TB) && !(W % TB) && Cin >= CS && !(Cin % CS) && N >= 32 &&
           !(N % 4);
}

template <int KS, int BN, int WM, int WN, int TM, int TN, int DBG = 0, int PB = 2, int S2 = 0, int KSP = 1, int NPL = 3, int RING = 0>
void x3_launch(const X3Args& a, dim3 grid, hipStream_t st) {
    hipLaunchKernelGGL((conv_halo_x3_kernel<KS, BN, WM, WN, TM, TN, DBG, PB, S2, KSP, NPL, RING>), grid, dim3(64 * WM * WN), 0, st, a);
}
// 16-bit elements of the two-plane prepared filter ({s_w, 1 / s_w} follow as two floats: + 4 elements)
size_t h2_w_elems(int rows, int kdim, int K) { return (size_t)K * K * ((kdim + CS - 1) / CS) * 2 * rows * CS; }

// Contraction split of the two-per-CU tile (conv_halo_x3_kernel, KSP == 2): launches of at most X3_KSPLIT_TILES tiles whose slab
// count is even and long enough to be worth halving.  DWC_X3_KSPLIT=0 switches it off.
constexpr int X3_KSPLIT_TILES = 256, X3_KSPLIT_TICKETS = 512;
constexpr size_t X3_KSPLIT_TILE_BYTES = 256 * 64 * sizeof(float);
// Number of tiles that run whole (a multiple of 8; the remaining tiles - split_from <= 256 are split), or -1: the launch is not
// split.  512 = the workgroups resident at once (two per CU): a last round of at most 256 tiles is the one worth halving.
// DWC_X3_KSPLIT=0 switches the split off.  (Splitting only the tail of larger launches -- 768 tiles on 512 slots -- was measured in
// r04: no gain, 330.6 against 323.8 us; workgroups are dispatched as slots free up, there is no "last round".  The kernel still
// understands split_from > 0, the launcher no longer asks for it.)
long x3_ksplit_from(long tiles, int slabs) {
    static const int on = getenv("DWC_X3_KSPLIT") ? atoi(getenv("DWC_X3_KSPLIT")) : 1;
    if (!on || slabs < 8 || (slabs & 1) || (tiles & 7)) return -1;                              // (tiles % 8: pairs share an XCD)
    // (launches of up to 512 tiles split whole were measured too: no gain at 5x5 batch 16, worse at 3x3 batch 32 and stride 2)
    return tiles <= X3_KSPLIT_TILES ? 0 : -1;
}
bool x3_ksplit_on(long tiles, int slabs) { return x3_ksplit_from(tiles, slabs) >= 0; }
size_t x3_ksplit_bytes(long tiles, int slabs) {
    const long from = x3_ksplit_from(tiles, slabs);
    return from < 0 ? 0 : (size_t)(tiles - from) * X3_KSPLIT_TILE_BYTES;
}

// (A second patch buffer in the stride-2 forms -- PB = 2 with the two-per-CU tile: 81.7 KB of LDS, still two per CU -- was measured
// equal to the conversion at the slab boundary, B = 48: 261.2 / 264.2 against 260.1 / 265.6 us; not instantiated.)
bool x3_s2_ok(int B, int H, int W, int Cin, int N) {
    return B > 0 && H >= 2 * TB && W >= 2 * TB && !(H % (2 * TB)) && !(W % (2 * TB)) && Cin >= CS && !(Cin % CS) && N >= 32 && !(N % 4) &&
           (size_t)B * H * W * Cin < 0x7fffffffull;
}

}  // namespace

extern "C" {

int dwc_x3_conv2d_same_ok(int B, int H, int W, int Cin, int Cout, int K) { return x3_ok(B, H, W, Cin, Cout, K) ? 1 : 0; }

size_t dwc_x3_weight_prepared_elems(int rows, int kdim, int K) {
    return (size_t)K * K * ((kdim + CS - 1) / CS) * 3 * rows * CS;
}

/* w (fp32 OIHW) -> three bf16 planes per (tap, 16-channel slab), exact split.  dgrad == 0: rows = Cout rounded up by the
 * caller to `rows`, contraction over Cin; dgrad != 0: rows >= Cin, contraction over Cout, taps rotated. */
int dwc_x3_weight_prepare(const float* w_oihw, void* out, int Cout, int Cin, int K, int rows, int dgrad, void* stream) {
    if (!w_oihw || !out || Cout <= 0 || Cin <= 0 || K <= 0 || rows < (dgrad ? Cin : Cout)) return DWC_EINVAL;
    const int kdim = dgrad ? Cout : Cin;
    const size_t total = dwc_x3_weight_prepared_elems(rows, kdim, K) / 3;
    hipLaunchKernelGGL(x3_weight_prepare_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oihw, (bf16*)out, Cout,
                       Cin, K, rows, kdim, dgrad);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

/* y = act(conv_KxK(pad(x)) + bias) for fp32 NHWC tensors, stride 1, pad (K-1)/2, K in {3,5}, H and W multiples of 16, Cin a
 * multiple of 16; N = channels (row stride) of y, a multiple of 4, rows = row count the weights were prepared with (>= N).
 * reflect != 0: reflect padding (forward); reflect == 0: zero padding (interior of the data gradient with dgrad-prepared
 * weights, to be followed by dwc_conv2d_bwd_data_ring). */
int dwc_x3_conv2d_same_add(const float* x, const void* w_prepared, const float* bias, const float* add, float* y, int B, int H, int W,
                           int Cin, int N, int rows, int K, int act, int reflect, void* stream);
int dwc_x3_conv2d_same_add_ws(const float* x, const void* w_prepared, const float* bias, const float* add, float* y, int B, int H, int W,
                              int Cin, int N, int rows, int K, int act, int reflect, void* ws, size_t ws_bytes, unsigned* tickets,
                              void* stream);
int dwc_x3_conv2d_same(const float* x, const void* w_prepared, const float* bias, float* y, int B, int H, int W, int Cin, int N,
                       int rows, int K, int act, int reflect, void* stream) {
    return dwc_x3_conv2d_same_add(x, w_prepared, bias, nullptr, y, B, H, W, Cin, N, rows, K, act, reflect, stream);
}

/* The same with `add` ([B,H,W,N] fp32 or NULL) added to the result behind bias and activation (see
 * dwc_bf16_conv2d_same_halo_add: the identity-branch gradient of a ResBlock rides on the data gradient of its first convolution). */
int dwc_x3_conv2d_same_add(const float* x, const void* w_prepared, const float* bias, const float* add, float* y, int B, int H, int W,
                           int Cin, int N, int rows, int K, int act, int reflect, void* stream) {
    return dwc_x3_conv2d_same_add_ws(x, w_prepared, bias, add, y, B, H, W, Cin, N, rows, K, act, reflect, nullptr, 0, nullptr, stream);
}

/* Scratch for the contraction split of small launches (see conv_halo_x3_kernel, KSP == 2): bytes of `ws` that
 * dwc_x3_conv2d_same_add_ws / dwc_x3_conv2d_s2_ws want for this shape (0: the launch is not split), and the number of 32-bit
 * ticket words the caller keeps per stream -- zero before the first call, left at zero by every call. */
size_t dwc_x3_conv2d_ksplit_ws_bytes(int B, int H, int W, int Cin, int N, int K, int stride) {
    if (stride == 2) {
        if (!x3_s2_ok(B, H, W, Cin, N)) return 0;
        const long tiles = (long)B * ((H / 2) / TB) * ((W / 2) / TB) * ((N + 63) / 64);
        return x3_ksplit_bytes(tiles, 4 * (Cin / CS));
    }
    if (!x3_ok(B, H, W, Cin, N, K)) return 0;
    const long tiles = (long)B * (H / TB) * (W / TB) * ((N + 63) / 64);
    return x3_ksplit_bytes(tiles, Cin / CS);
}
int dwc_x3_conv2d_ksplit_ticket_words(void) { return X3_KSPLIT_TICKETS + 1; }      // (+ the sticky status word behind the tickets)

/* dwc_x3_conv2d_same_add with the scratch of the contraction split: ws / tickets may be NULL (or ws_bytes too small), the launch
 * then runs unsplit.  Results do not depend on which form ran beyond fp32 summation order (two half sums instead of one). */
}  // extern "C"

template <int NPL>
static int x3_same_add_ws_impl(const float* x, const void* xs, unsigned xs_epoch, const void* w_prepared, const float* bias, const float* add,
                               float* y, int B, int H, int W, int Cin, int N, int rows, int K, int act, int reflect, void* ws,
                               size_t ws_bytes, unsigned* tickets, void* stream, void* ys = nullptr, unsigned ys_epoch = 0, bool ring = false) {
    // (two planes: x is addressed through 31-bit buffer offsets)
    if (!x || !w_prepared || !y || !x3_ok(B, H, W, Cin, N, K) || rows < N || (NPL == 2 && (!xs || (size_t)B * H * W * Cin * 4 >= 0x80000000ull)))
        return DWC_EINVAL;
    // (ring: the reflect adjoint's border ring inside the launch -- two-plane data gradients on images of at least two tiles per side)
    if (ring && (NPL != 2 || reflect || H < 2 * TB || W < 2 * TB)) return DWC_EINVAL;
    X3Args a;
    a.x = x; a.w = (const bf16*)w_prepared; a.bias = bias; a.add = add; a.y = y;
    a.xs = (const unsigned long long*)xs; a.xs_epoch = xs_epoch; a.w_elems = h2_w_elems(rows, Cin, K);
    a.ys = (unsigned long long*)ys; a.ys_epoch = ys_epoch;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.N = N; a.rows = rows; a.act = act; a.reflect = reflect;
    a.blocks_x = W / TB; a.blocks_per_img = (H / TB) * (W / TB);
    const int blocks = B * a.blocks_per_img;
    // Two 4-wave workgroups of 256 pixels x 64 channels per CU (single patch buffer, <= 67 KB of LDS each).  Two independent
    // workgroups drift out of phase, so one's MFMAs run beside the other's fragment reads, staging and barriers; measured in r02-r04
    // against one 8-wave workgroup per CU with 64- / 128- / 256-channel tiles: 5x5 128->64 +27 %, 5x5 256->128 +5 %, 3x3 256->256 at
    // B=48 +16 %, small launches equal -- the 8-wave tiles, the 32-channel tile for small launches and the phase offset between the
    // two workgroups (no effect at any offset) are no longer instantiated (r05; numbers in DESIGN.md sections 3.9 / 9).
    a.tiles_n = (N + 63) / 64;
    const dim3 g2(blocks * a.tiles_n);
    const size_t need = dwc_x3_conv2d_ksplit_ws_bytes(B, H, W, Cin, N, K, 1);
    if (need && ws && tickets && ws_bytes >= need) {
        a.part = (float*)ws;
        a.tickets = tickets;
        a.status = tickets + X3_KSPLIT_TICKETS;
        a.split_from = (int)x3_ksplit_from(g2.x, Cin / CS);
        const dim3 g4(2 * g2.x - a.split_from);
        if constexpr (NPL == 2) {
            if (ring && K == 3) x3_launch<3, 64, 4, 1, 2, 2, 0, 1, 0, 2, 2, 1>(a, g4, (hipStream_t)stream);
            else if (ring) x3_launch<5, 64, 4, 1, 2, 2, 0, 1, 0, 2, 2, 1>(a, g4, (hipStream_t)stream);
        }
        if (ring) {
        } else if (K == 3) x3_launch<3, 64, 4, 1, 2, 2, 0, 1, 0, 2, NPL>(a, g4, (hipStream_t)stream);
        else x3_launch<5, 64, 4, 1, 2, 2, 0, 1, 0, 2, NPL>(a, g4, (hipStream_t)stream);
    }
#ifdef DWC_DEV_ABLATIONS      // timing-only ablations (WRONG results): compiled only with -DDWC_DEV_ABLATIONS, never in the shipped .so
    // (benchmarks/h2_ablation_bench.py; DWC_X3_DBG: 1 no MFMA, 2 no fragment reads, 4 no weight staging, 8 no barrier, 16 no patch
    // refresh / flush, 32 no conversion at the slab boundary, 64 no flush, 128 no patch DMA, 256 no epilogue stores, 512 no main loop, sums of those, other: empty skeleton --
    // profiles/r05_h2_ablation.txt)
    else if (NPL == 2 && getenv("DWC_X3_DBG") && atoi(getenv("DWC_X3_DBG"))) {
        const int dbg = atoi(getenv("DWC_X3_DBG"));
        if constexpr (NPL == 2) {
            if (K == 3) {
                if (dbg == 1) x3_launch<3, 64, 4, 1, 2, 2, 1, 1, 0, 1, 2>(a, g2, (hipStream_t)stream);
                else if (dbg == 2) x3_launch<3, 64, 4, 1, 2, 2, 2, 1, 0, 1, 2>(a, g2, (hipStream_t)stream);
                else if (dbg == 4) x3_launch<3, 64, 4, 1, 2, 2, 4, 1, 0, 1, 2>(a, g2, (hipStream_t)stream);
                else if (dbg == 8) x3_launch<3, 64, 4, 1, 2, 2, 8, 1, 0, 1, 2>(a, g2, (hipStream_t)stream);
                else if (dbg == 16) x3_launch<3, 64, 4, 1, 2, 2, 16, 1, 0, 1, 2>(a, g2, (hipStream_t)stream);
                else if (dbg == 32) x3_launch<3, 64, 4, 1, 2, 2, 32, 1, 0, 1, 2>(a, g2, (hipStream_t)stream);
                else if (dbg == 64) x3_launch<3, 64, 4, 1, 2, 2, 64, 1, 0, 1, 2>(a, g2, (hipStream_t)stream);
                else if (dbg == 128) x3_launch<3, 64, 4, 1, 2, 2, 128, 1, 0, 1, 2>(a, g2, (hipStream_t)stream);
                else if (dbg == 3) x3_launch<3, 64, 4, 1, 2, 2, 3, 1, 0, 1, 2>(a, g2, (hipStream_t)stream);
                else if (dbg == 287) x3_launch<3, 64, 4, 1, 2, 2, 287, 1, 0, 1, 2>(a, g2, (hipStream_t)stream);
                else if (dbg == 543) x3_launch<3, 64, 4, 1, 2, 2, 543, 1, 0, 1, 2>(a, g2, (hipStream_t)stream);
                else if (dbg == 799) x3_launch<3, 64, 4, 1, 2, 2, 799, 1, 0, 1, 2>(a, g2, (hipStream_t)stream);
                else if (dbg == 256) x3_launch<3, 64, 4, 1, 2, 2, 256, 1, 0, 1, 2>(a, g2, (hipStream_t)stream);
                else x3_launch<3, 64, 4, 1, 2, 2, 31, 1, 0, 1, 2>(a, g2, (hipStream_t)stream);
            } else {
                if (dbg == 1) x3_launch<5, 64, 4, 1, 2, 2, 1, 1, 0, 1, 2>(a, g2, (hipStream_t)stream);
                else if (dbg == 2) x3_launch<5, 64, 4, 1, 2, 2, 2, 1, 0, 1, 2>(a, g2, (hipStream_t)stream);
                else if (dbg == 4) x3_launch<5, 64, 4, 1, 2, 2, 4, 1, 0, 1, 2>(a, g2, (hipStream_t)stream);
                else if (dbg == 8) x3_launch<5, 64, 4, 1, 2, 2, 8, 1, 0, 1, 2>(a, g2, (hipStream_t)stream);
                else if (dbg == 16) x3_launch<5, 64, 4, 1, 2, 2, 16, 1, 0, 1, 2>(a, g2, (hipStream_t)stream);
                else if (dbg == 32) x3_launch<5, 64, 4, 1, 2, 2, 32, 1, 0, 1, 2>(a, g2, (hipStream_t)stream);
                else if (dbg == 64) x3_launch<5, 64, 4, 1, 2, 2, 64, 1, 0, 1, 2>(a, g2, (hipStream_t)stream);
                else if (dbg == 128) x3_launch<5, 64, 4, 1, 2, 2, 128, 1, 0, 1, 2>(a, g2, (hipStream_t)stream);
                else if (dbg == 3) x3_launch<5, 64, 4, 1, 2, 2, 3, 1, 0, 1, 2>(a, g2, (hipStream_t)stream);
                else if (dbg == 287) x3_launch<5, 64, 4, 1, 2, 2, 287, 1, 0, 1, 2>(a, g2, (hipStream_t)stream);
                else if (dbg == 543) x3_launch<5, 64, 4, 1, 2, 2, 543, 1, 0, 1, 2>(a, g2, (hipStream_t)stream);
                else if (dbg == 799) x3_launch<5, 64, 4, 1, 2, 2, 799, 1, 0, 1, 2>(a, g2, (hipStream_t)stream);
                else if (dbg == 256) x3_launch<5, 64, 4, 1, 2, 2, 256, 1, 0, 1, 2>(a, g2, (hipStream_t)stream);
                else x3_launch<5, 64, 4, 1, 2, 2, 31, 1, 0, 1, 2>(a, g2, (hipStream_t)stream);
            }
        }
    }
#endif
    else if (ring) {
        if constexpr (NPL == 2) {
            if (K == 3) x3_launch<3, 64, 4, 1, 2, 2, 0, 1, 0, 1, 2, 1>(a, g2, (hipStream_t)stream);
            else x3_launch<5, 64, 4, 1, 2, 2, 0, 1, 0, 1, 2, 1>(a, g2, (hipStream_t)stream);
        }
    }
    else if (K == 3) x3_launch<3, 64, 4, 1, 2, 2, 0, 1, 0, 1, NPL>(a, g2, (hipStream_t)stream);
    else x3_launch<5, 64, 4, 1, 2, 2, 0, 1, 0, 1, NPL>(a, g2, (hipStream_t)stream);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

extern "C" {

/* (r06, ABI 8) DATA GRADIENT of a reflect-padded stride-1 "same" 3x3 / 5x5 convolution in ONE launch, two-plane split products:
 * dx[B,H,W,N] = interior (zero-rule convolution of dy[B,H,W,Cout] with w_prepared = dwc_h2_weight_prepare(dgrad = 1, `rows` rows)) +
 * the border ring of the padded gradient image folded back by the reflect rule (reference networks.py:579-585 through autograd) +
 * `add` (NULL or [B,H,W,N]).  The border tiles read pre-summed patch pixels for the rows / columns the ring folds onto
 * (conv_halo_x3_kernel, RING): replaces dwc_h2_conv2d_same_add_ws(reflect = 0) + dwc_conv2d_bwd_data_ring.  dy_amax / dy_epoch, ws,
 * ws_bytes, tickets as there.  H, W multiples of 16 and >= 32; otherwise as dwc_h2_conv2d_same_add_ws (DWC_EINVAL: use the two calls). */
int dwc_h2_conv2d_bwd_data_same_fused(const float* dy, const void* dy_amax, unsigned dy_epoch, const void* w_prepared, const float* add,
                                      float* dx, int B, int H, int W, int Cout, int N, int rows, int K, void* ws, size_t ws_bytes,
                                      unsigned* tickets, void* stream) {
    return x3_same_add_ws_impl<2>(dy, dy_amax, dy_epoch, w_prepared, nullptr, add, dx, B, H, W, Cout, N, rows, K, DWC_ACT_NONE, 0, ws, ws_bytes,
                                  tickets, stream, nullptr, 0, true);
}

int dwc_x3_conv2d_same_add_ws(const float* x, const void* w_prepared, const float* bias, const float* add, float* y, int B, int H, int W,
                              int Cin, int N, int rows, int K, int act, int reflect, void* ws, size_t ws_bytes, unsigned* tickets,
                              void* stream) {
    return x3_same_add_ws_impl<3>(x, nullptr, 0, w_prepared, bias, add, y, B, H, W, Cin, N, rows, K, act, reflect, ws, ws_bytes, tickets, stream);
}

/* ---- the same layers as TWO-plane f16 split products (r05; see split2h / h2_scale): x_amax = the absmax slot of x (dwc_absmax or a
 * producing kernel) carrying `x_epoch`, w_prepared = dwc_h2_weight_prepare.  Same tiles, scratch and tickets as the three-plane form. */
int dwc_h2_conv2d_same_add_ws(const float* x, const void* x_amax, unsigned x_epoch, const void* w_prepared, const float* bias,
                              const float* add, float* y, void* y_amax, unsigned y_epoch, int B, int H, int W, int Cin, int N, int rows,
                              int K, int act, int reflect, void* ws, size_t ws_bytes, unsigned* tickets, void* stream) {
    return x3_same_add_ws_impl<2>(x, x_amax, x_epoch, w_prepared, bias, add, y, B, H, W, Cin, N, rows, K, act, reflect, ws, ws_bytes, tickets,
                                  stream, y_amax, y_epoch);
}

size_t dwc_h2_weight_prepared_elems(int rows, int kdim, int K) { return h2_w_elems(rows, kdim, K) + 8; }

/* w (fp32 OIHW) -> two f16 planes of s_w * w per (tap, 16-channel slab) + {s_w, 1 / s_w}; w_amax: absmax slot of w carrying w_epoch
 * (dwc_absmax over the Cout*Cin*K*K floats).  Geometry arguments as dwc_x3_weight_prepare. */
int dwc_h2_weight_prepare(const float* w_oihw, void* out, int Cout, int Cin, int K, int rows, int dgrad, const void* w_amax,
                          unsigned w_epoch, void* stream) {
    if (!w_oihw || !out || !w_amax || Cout <= 0 || Cin <= 0 || K <= 0 || rows < (dgrad ? Cin : Cout)) return DWC_EINVAL;
    const int kdim = dgrad ? Cout : Cin;
    const size_t total = h2_w_elems(rows, kdim, K) / 2;
    hipLaunchKernelGGL(h2_weight_prepare_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, w_oihw, (unsigned short*)out,
                       Cout, Cin, K, rows, kdim, dgrad, (const unsigned long long*)w_amax, w_epoch);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

/* slot = max(slot, (epoch << 32) | bits of max |x[0..n)|): the largest-magnitude slot the two-plane kernels scale a tensor by.
 * x 16-byte aligned.  Slots are 8 bytes of caller memory, zero before their first use; epochs handed to one slot must not decrease. */
int dwc_absmax(const float* x, size_t n, void* slot, unsigned epoch, void* stream) {
    if (!x || !slot || ((uintptr_t)x & 15)) return DWC_EINVAL;
    if (n == 0) return DWC_OK;
    const size_t per_block = 256 * 4 * 4;      // floats one block covers per round of four loads
    size_t blocks = (n + per_block - 1) / per_block;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, n, (unsigned long long*)slot, epoch);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

int dwc_x3_conv2d_s2_ok(int B, int H, int W, int Cin, int Cout) { return x3_s2_ok(B, H, W, Cin, Cout) ? 1 : 0; }

/* y = act(conv4x4_stride2(reflect_pad1(x)) + bias) for fp32 NHWC tensors as split products (see conv_halo_x3_kernel, S2):
 * x:[B,H,W,Cin], y:[B,H/2,W/2,N], H and W multiples of 32, Cin a multiple of 16; w_prepared = dwc_x3_weight_prepare(K = 4, forward)
 * with `rows` >= N rows. */
int dwc_x3_conv2d_s2_ws(const float* x, const void* w_prepared, const float* bias, float* y, int B, int H, int W, int Cin, int N, int rows,
                        int act, void* ws, size_t ws_bytes, unsigned* tickets, void* stream);
int dwc_x3_conv2d_s2(const float* x, const void* w_prepared, const float* bias, float* y, int B, int H, int W, int Cin, int N, int rows,
                     int act, void* stream) {
    return dwc_x3_conv2d_s2_ws(x, w_prepared, bias, y, B, H, W, Cin, N, rows, act, nullptr, 0, nullptr, stream);
}

}  // extern "C"

template <int NPL>
static int x3_s2_ws_impl(const float* x, const void* xs, unsigned xs_epoch, const void* w_prepared, const float* bias, float* y, int B, int H,
                         int W, int Cin, int N, int rows, int act, void* ws, size_t ws_bytes, unsigned* tickets, void* stream,
                         void* ys = nullptr, unsigned ys_epoch = 0) {
    if (!x || !w_prepared || !y || !x3_s2_ok(B, H, W, Cin, N) || rows < N || (NPL == 2 && (!xs || (size_t)B * H * W * Cin * 4 >= 0x80000000ull)))
        return DWC_EINVAL;
    X3Args a;
    a.x = x; a.w = (const bf16*)w_prepared; a.bias = bias; a.add = nullptr; a.y = y;
    a.xs = (const unsigned long long*)xs; a.xs_epoch = xs_epoch; a.w_elems = h2_w_elems(rows, Cin, 4);
    a.ys = (unsigned long long*)ys; a.ys_epoch = ys_epoch;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.N = N; a.rows = rows; a.act = act; a.reflect = 1;
    a.blocks_x = (W / 2) / TB; a.blocks_per_img = ((H / 2) / TB) * ((W / 2) / TB);
    a.tiles_n = (N + 63) / 64;
    const size_t need = dwc_x3_conv2d_ksplit_ws_bytes(B, H, W, Cin, N, 4, 2);
    if (need && ws && tickets && ws_bytes >= need) {
        a.part = (float*)ws;
        a.tickets = tickets;
        a.status = tickets + X3_KSPLIT_TICKETS;
        const int tiles = B * a.blocks_per_img * a.tiles_n;
        a.split_from = (int)x3_ksplit_from(tiles, 4 * (Cin / CS));
        x3_launch<2, 64, 4, 1, 2, 2, 0, 1, 1, 2, NPL>(a, dim3(2 * tiles - a.split_from), (hipStream_t)stream);
    } else {
        x3_launch<2, 64, 4, 1, 2, 2, 0, 1, 1, 1, NPL>(a, dim3(B * a.blocks_per_img * a.tiles_n), (hipStream_t)stream);
    }
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

template <int NPL>
static int x3_s2_bwd_data_impl(const float* dy, const void* ds, unsigned ds_epoch, const void* w_prepared, float* dx, int B, int H, int W,
                               int Cin, int Cout, int rows, void* stream, bool ring = false) {
    if (!dy || !w_prepared || !dx || !dwc_x3_conv2d_s2_bwd_data_ok(B, H, W, Cin, Cout) || rows < Cin ||
        (NPL == 2 && (!ds || (size_t)B * (H / 2) * (W / 2) * Cout * 4 >= 0x80000000ull)))
        return DWC_EINVAL;
    X3Args a;
    a.x = dy; a.w = (const bf16*)w_prepared; a.bias = nullptr; a.add = nullptr; a.y = dx;
    a.xs = (const unsigned long long*)ds; a.xs_epoch = ds_epoch; a.w_elems = h2_w_elems(rows, Cout, 4);
    a.B = B; a.H = H / 2; a.W = W / 2; a.Cin = Cout; a.N = Cin; a.rows = rows; a.act = DWC_ACT_NONE; a.reflect = 0;
    a.blocks_x = (W / 2) / TB; a.blocks_per_img = ((H / 2) / TB) * ((W / 2) / TB);
    a.tiles_n = Cin / 64;
    if (ring) {
        if constexpr (NPL == 2) x3_launch<2, 64, 4, 1, 2, 2, 0, 1, 2, 1, 2, 1>(a, dim3(4 * B * a.blocks_per_img * a.tiles_n), (hipStream_t)stream);
        else return DWC_EINVAL;
    } else {
        x3_launch<2, 64, 4, 1, 2, 2, 0, 1, 2, 1, NPL>(a, dim3(4 * B * a.blocks_per_img * a.tiles_n), (hipStream_t)stream);
    }
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

extern "C" {

/* dwc_x3_conv2d_s2 with the scratch of the contraction split (see dwc_x3_conv2d_same_add_ws). */
int dwc_x3_conv2d_s2_ws(const float* x, const void* w_prepared, const float* bias, float* y, int B, int H, int W, int Cin, int N, int rows,
                        int act, void* ws, size_t ws_bytes, unsigned* tickets, void* stream) {
    return x3_s2_ws_impl<3>(x, nullptr, 0, w_prepared, bias, y, B, H, W, Cin, N, rows, act, ws, ws_bytes, tickets, stream);
}
/* two-plane f16 form (see dwc_h2_conv2d_same_add_ws); w_prepared = dwc_h2_weight_prepare(K = 4) */
int dwc_h2_conv2d_s2_ws(const float* x, const void* x_amax, unsigned x_epoch, const void* w_prepared, const float* bias, float* y,
                        void* y_amax, unsigned y_epoch, int B, int H, int W, int Cin, int N, int rows, int act, void* ws, size_t ws_bytes,
                        unsigned* tickets, void* stream) {
    return x3_s2_ws_impl<2>(x, x_amax, x_epoch, w_prepared, bias, y, B, H, W, Cin, N, rows, act, ws, ws_bytes, tickets, stream, y_amax, y_epoch);
}

/* INTERIOR of the data gradient of the same layers (conv_halo_x3_kernel, S2 == 2): dy:[B,H/2,W/2,Cout] fp32 -> the H x W pixels
 * of dx:[B,H,W,N] (N = the convolution's input channels), w_prepared = dwc_x3_weight_prepare(K = 4, dgrad = 1) with `rows` >= N
 * rows.  Every pixel of dx is written (no accumulation); the border ring of the padded gradient image still has to be folded
 * onto it: dwc_conv2d_bwd_data_s2_ring.  H, W multiples of 32, Cout a multiple of 16. */
int dwc_x3_conv2d_s2_bwd_data_ok(int B, int H, int W, int Cin, int Cout) {
    return (x3_s2_ok(B, H, W, Cout, Cin) && !(Cin % 64) && (size_t)B * H * W * Cin < 0x7fffffffull) ? 1 : 0;
}

int dwc_x3_conv2d_s2_bwd_data(const float* dy, const void* w_prepared, float* dx, int B, int H, int W, int Cin, int Cout, int rows,
                              void* stream) {
    return x3_s2_bwd_data_impl<3>(dy, nullptr, 0, w_prepared, dx, B, H, W, Cin, Cout, rows, stream);
}
/* two-plane f16 form; w_prepared = dwc_h2_weight_prepare(K = 4, dgrad = 1), dy_amax = absmax slot of dy */
int dwc_h2_conv2d_s2_bwd_data(const float* dy, const void* dy_amax, unsigned dy_epoch, const void* w_prepared, float* dx, int B, int H, int W,
                              int Cin, int Cout, int rows, void* stream) {
    return x3_s2_bwd_data_impl<2>(dy, dy_amax, dy_epoch, w_prepared, dx, B, H, W, Cin, Cout, rows, stream);
}

/* (r06, ABI 8) ... with the border ring of the padded gradient image folded in by the same launch (the reflect-pad-1 adjoint: padded row 0
 * onto dx row 1, row H+1 onto H-2, columns alike; conv_halo_x3_kernel RING, S2 == 2): the WHOLE data gradient of a 4x4 stride-2
 * reflect-pad-1 convolution, no dwc_conv2d_bwd_data_s2_ring behind it.  Arguments and shapes as dwc_h2_conv2d_s2_bwd_data. */
int dwc_h2_conv2d_s2_bwd_data_fused(const float* dy, const void* dy_amax, unsigned dy_epoch, const void* w_prepared, float* dx, int B, int H,
                                    int W, int Cin, int Cout, int rows, void* stream) {
    return x3_s2_bwd_data_impl<2>(dy, dy_amax, dy_epoch, w_prepared, dx, B, H, W, Cin, Cout, rows, stream, true);
}

size_t dwc_x3_conv2d_wgrad_ws_bytes(int B, int H, int W, int Cin, int Cout, int K) {
    const int bn = x3_wgrad_bn(B, H, W, Cin, Cout, K);
    if (!bn) return 0;
    int splits = 0;
    for (int npl = 2; npl <= 3; ++npl) {      // (one scratch size for both forms: the larger of the two plans)
        int s, ups;
        x3_wgrad_plan(B, H, W, Cin, Cout, K, bn, npl, &s, &ups);
        splits = s > splits ? s : splits;
    }
    return (size_t)splits * (bn == 128 ? 1 : 2) * K * K * Cin * Cout * sizeof(float);
}

/* dw (fp32 OIHW, [cout_real][cin_real][K][K]) of a reflect-padded stride-1 "same" K x K convolution (K in {3,5}) from the fp32 NHWC
 * tensors x:[B,H,W,Cin] and dy:[B,H,W,Cout], or of the 4x4 stride-2 reflect-pad-1 convolution (K = 4, dy:[B,H/2,W/2,Cout]), split
 * products (see wgrad_x3_kernel).  ws_bytes == 0: shape not handled (K in {3,5}: H % 8 == 0, W % 16 == 0; K = 4: H % 8 == 0,
 * W % 32 == 0; Cin and Cout multiples of 64) - use dwc_conv2d_bwd_weight. */
}  // extern "C"

template <int NPL>
static int x3_wgrad_impl(const float* x, const void* xs, unsigned xs_epoch, const float* dy, const void* dys, unsigned dys_epoch, float* dw_oihw,
                         int B, int H, int W, int Cin, int Cout, int K, int cin_real, int cout_real, void* ws, size_t ws_bytes, void* stream) {
    const int bn = x3_wgrad_bn(B, H, W, Cin, Cout, K);
    if (!x || !dy || !dw_oihw || !bn || cin_real > Cin || cout_real > Cout || (NPL == 2 && (!xs || !dys))) return DWC_EINVAL;
    int splits, ups;
    x3_wgrad_plan(B, H, W, Cin, Cout, K, bn, NPL, &splits, &ups);
    const int halves = bn == 128 ? 1 : 2;
    if (!ws || ws_bytes < (size_t)splits * halves * K * K * Cin * Cout * sizeof(float)) return DWC_EWORKSPACE;
    X3WgradArgs a;
    a.x = x; a.dy = dy; a.slab = (float*)ws;
    a.xs = (const unsigned long long*)xs; a.xs_epoch = xs_epoch; a.dys = (const unsigned long long*)dys; a.dys_epoch = dys_epoch;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.N = Cout;
    if (K == 4) {
        a.units_x = W / 32; a.units_per_img = (H / 8) * (W / 32);      // 4x16-pixel units of the H/2 x W/2 dY grid
    } else {
        a.units_x = W / 16; a.units_per_img = (H / 8) * (W / 16);
    }
    a.total_units = B * a.units_per_img; a.units_per_split = ups;
    a.n_tiles = Cout / bn;
    const int ciw = x3_wgrad_ciw();
    a.roles = (Cin / ciw) * a.n_tiles * K;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(a.roles * splits);
    if (K == 4) {
        if (bn == 128) hipLaunchKernelGGL((wgrad_x3_kernel<4, 128, 64, 2, NPL>), grid, dim3(512), 0, st, a);
        else hipLaunchKernelGGL((wgrad_x3_kernel<4, 64, 64, 2, NPL>), grid, dim3(512), 0, st, a);
    } else if (K == 3 && bn == 128) hipLaunchKernelGGL((wgrad_x3_kernel<3, 128, 64, 2, NPL>), grid, dim3(512), 0, st, a);
    else if (K == 3) hipLaunchKernelGGL((wgrad_x3_kernel<3, 64, 64, 2, NPL>), grid, dim3(512), 0, st, a);
    else if (bn == 128) hipLaunchKernelGGL((wgrad_x3_kernel<5, 128, 64, 2, NPL>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((wgrad_x3_kernel<5, 64, 64, 2, NPL>), grid, dim3(512), 0, st, a);
    DWC_LAUNCH_CHECK();
    const size_t total = (size_t)K * K * Cin * Cout;
    hipLaunchKernelGGL(x3_wgrad_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, (const float*)ws, dw_oihw, splits * halves,
                       K * K * Cin, Cout, Cin, K * K, cin_real, cout_real);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

extern "C" {

int dwc_x3_conv2d_wgrad(const float* x, const float* dy, float* dw_oihw, int B, int H, int W, int Cin, int Cout, int K, int cin_real,
                        int cout_real, void* ws, size_t ws_bytes, void* stream) {
    return x3_wgrad_impl<3>(x, nullptr, 0, dy, nullptr, 0, dw_oihw, B, H, W, Cin, Cout, K, cin_real, cout_real, ws, ws_bytes, stream);
}
/* two-plane f16 form: both operands scaled by their absmax slots (same scratch as dwc_x3_conv2d_wgrad) */
int dwc_h2_conv2d_wgrad(const float* x, const void* x_amax, unsigned x_epoch, const float* dy, const void* dy_amax, unsigned dy_epoch,
                        float* dw_oihw, int B, int H, int W, int Cin, int Cout, int K, int cin_real, int cout_real, void* ws, size_t ws_bytes,
                        void* stream) {
    return x3_wgrad_impl<2>(x, x_amax, x_epoch, dy, dy_amax, dy_epoch, dw_oihw, B, H, W, Cin, Cout, K, cin_real, cout_real, ws, ws_bytes, stream);
}

}  // extern "C"
