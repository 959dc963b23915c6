// Halo-tiled bf16 convolution for the stride-1 "same" layers (3x3 ResBlock and 5x5 upsampling convolutions: 11.5 of the
// decoder's 11.75 GMAC per image, reference networks.py:514-515, networks_v2.py:153-156), forward and data-gradient
// interior.
//
// Why a second formulation: at bf16 MFMA rates the im2col GEMM of conv_bf16.hip is bound by what one CU can pull out of L2
// into LDS (r02 measurements: ~50-60 GB/s per CU at every tile shape, 30-36 % of the MFMA peak), and im2col re-stages every
// input pixel once per filter tap.  Here a workgroup owns a 16x16 block of output pixels of ONE image and BN output
// channels; per 64-channel slab it stages the (16+K-1)^2 input patch ONCE (reflect or zero rule applied while staging,
// like everywhere else) and walks the K*K taps over it in LDS: the A operand of tap (kh,kw) is the same patch read at a
// pixel offset, so activation traffic drops by K*K and only the weight slab (BN x 64) is staged per tap.
//   L2->LDS bytes per 256x256x64 MFMA step: 32 KB weights + 41.5/9 KB patch = 37 KB  (im2col 128x128 tile: 128 KB)
// Layouts are those of conv_bf16.hip: 128-byte rows (64 bf16) with the 16-byte chunk index XOR-swizzled by (row>>1)&7 on
// the source side (patch rows: by (patch column >> 1) & 7), buffer_load ... lds staging with scalar slab offsets, D = W_tile . X_tile^T so a lane owns a pixel.
#include <type_traits>

#include "conv_geom.h"

namespace {

typedef __bf16 bf16;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int BK = 64;
constexpr int TB = 16;                 // block edge: 16x16 output pixels

__device__ __forceinline__ bf16x4 pack4h(float a, float b, float c, float d) {
    bf16x4 r;
    r[0] = (bf16)a; r[1] = (bf16)b; r[2] = (bf16)c; r[3] = (bf16)d;
    return r;
}

struct HaloArgs {
    const bf16* x;       // [B][H][W][Cin]
    const bf16* w;       // [N][Kp], k = (kh*K + kw)*Cin + ci   (dwc_bf16_weight_prepare_fwd / _dgrad layout)
    const float* bias;   // [N] or null
    const bf16* add;     // [B][H][W][N] or null: added to the result behind bias + activation (a second gradient w.r.t. the same tensor)
    bf16* y;             // [B][H][W][N]
    int B, H, W, Cin, logCin, N, K, Kp, act, reflect;
    int blocks_x, blocks_per_img, tiles_n;
};

// bf16 + bf16 as torch adds them: both to fp32, one rounding of the sum
__device__ __forceinline__ bf16x8 halo_add8(bf16x8 a, bf16x8 b) {
    bf16x8 r;
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = (bf16)((float)a[k] + (float)b[k]);
    return r;
}

// (The compiler-scheduled 32x32x16 kernel of round 2, conv_halo_kernel, and its knobs DWC_HALO16 / DWC_HALO_DUO / DWC_HALO_BN128 were
// removed in round 5: the hand-scheduled kernels below replaced it in round 3 -- measurements in DESIGN.md 3.8 / 3.8b.)
#include "conv_halo16_bf16.inc"

// ------------------------------------------------------------------------------------------
// Weight gradient of the same layers, halo form.  dW[tap][ci][co] = sum_pixels x[pixel + tap][ci] * dY[pixel][co]: the
// im2col form (conv_bf16.hip) re-stages x once per tap (r02 ablation: staging is 40-45 % of that kernel's time).  Here a
// workgroup owns CIW channels of x, BN channels of dY and a group of taps (all 9 of a 3x3; one filter COLUMN kw of a 5x5) and
// walks over 8x16-pixel units: per unit the x patch and the dY block are staged once into LDS.  Wave w keeps the 32(ci) x
// 32(co) accumulators of its (ci tile, co tile) for all TH x TW taps of the group.  Operand fragments come from LDS through
// the transposing read (4 consecutive pixels x 16 channels) and LDS read bandwidth is what bounds this kernel, so the unit
// is walked by PATCH row r: the TW fragments of x row r are read once and used for every kh (against dY row r - kh, kept in
// a rolling register window of TH fragments): (TW + 1) fragment reads per TH*TW MFMAs instead of TH*TW + 1.
// Pixel ranges are split over gridDim.y into fp32 slabs [split][tap*Cin + ci][co] that wgrad_halo_reduce sums in a fixed
// order.
// ------------------------------------------------------------------------------------------
struct WgradHaloArgs {
    const bf16* x;      // [B][H][W][Cin]
    const bf16* dy;     // [B][H][W][N]
    float* slab;        // [splits][K*K*Cin][N]
    int B, H, W, Cin, logCin, N;          // H, W: x; the dY grid is H x W (stride 1) or H/2 x W/2 (KS == 4: stride 2)
    int units_x, units_per_img, total_units, units_per_split;      // 8x16-pixel units of the dY grid
    int n_tiles, tap_groups;
};

// PROBE (the round-3 laboratory benchmarks/halo_lab.hip, removed in round 5, instantiated it): s_memtime stamps of wave 0 -- start / first unit staged / loop done / slabs stored.
// DBG (timing only, results wrong; lab and -DDWC_DEV_ABLATIONS builds): 1 no MFMA, 2 no fragment reads, 4 no staging in the
// loop, 8 no wait + barrier per unit.
// HALF (r06): the next unit's staging pieces are issued by ONE half of the waves (waves 0-3 for odd units, 4-7 for even ones: a wave and
// its SIMD partner are in different halves), each covering its partner's share too.  A wave issuing LDS-DMA instructions is held by the
// vector-memory path for ~140 cycles per instruction when all eight waves issue at once (7 pieces per thread and unit: ~1 100 cycles per unit
// with no MFMA running, by the probes of conv_narrow_persist_kernel); with one half issuing, the other half's MFMAs run meanwhile.  Worth
// 3-4 % on the 5x5 form (LDS read bandwidth still bounds it), nothing on the stride-2 one, and the 3x3 form has no register left for
// it (256 in use: 11 spilled, 166 -> 179 us): instantiated for KS == 5 only.
template <int KS, int BN, int DBG = 0, int PROBE = 0, int PFT = -1, int SPREAD = 0, int HALF = 0>
__global__ __launch_bounds__(512) void wgrad_halo_kernel(WgradHaloArgs a, unsigned long long* probe = nullptr) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned long long stamp[4] = {0, 0, 0, 0};
    if constexpr (PROBE) stamp[0] = __builtin_amdgcn_s_memtime();
    static_assert(BN == 128 || BN == 64, "dY tile width");
    // KS == 4: the 4x4 stride-2 reflect-pad-1 layers (reference networks.py:90,94,437, networks_v2.py:107-111).  Tap (kh, kw) =
    // (2 th + dy, 2 tw + dx) reads input pixel (2 (oy + th) + dy - 1, 2 (ox + tw) + dx - 1): per input-pixel parity (dy, dx) a
    // 2x2-tap stride-1 problem over the space-to-depth image, and the loader does the space-to-depth (one parity = one tap group).
    constexpr bool S2 = KS == 4;
    constexpr int TH = S2 ? 2 : KS, TW = S2 ? 2 : (KS == 3 ? 3 : 1);   // taps of one workgroup: TH filter rows x TW filter columns
    constexpr int CIW = BN == 128 ? 64 : 128;           // channels of x per workgroup (8 waves = CIW/32 x BN/32 tiles)
    constexpr int UH = 8, UW = 16;                      // unit: 8 rows x 16 columns = 128 pixels, one MFMA k-step per row
    constexpr int PW = UW + TW - 1, PH = UH + TH - 1, PPIX = PH * PW;
    constexpr int PCH = CIW / 8;                        // 16-byte chunks per patch pixel
    constexpr int P_RPP = 512 / PCH;                    // patch pixels per staging pass
    constexpr int PPASS = (PPIX + P_RPP - 1) / P_RPP;
    constexpr int P_TILE = PPASS * P_RPP * CIW;         // elements: [patch pixel][CIW channels]
    constexpr int D_TILE = 128 * BN;                    // [pixel][BN channels]
    constexpr int DCH = BN / 8;
    constexpr int D_RPP = 512 / DCH;
    constexpr int D_PASSES = 128 / D_RPP;
    constexpr int PAD = (KS - 1) / 2;
    constexpr int NT = TH * TW;
    __shared__ __attribute__((aligned(16))) bf16 smem[2 * (P_TILE + D_TILE)];
    bf16* sP = smem;
    bf16* sD = smem + 2 * P_TILE;

    const int t = threadIdx.x;
    const int lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
    // 1-D grid, XCD-aware (see wgrad_x3_kernel): the roles of one pixel split are consecutive virtual ids on one XCD and share
    // its L2 for the pixels of x and dY they all re-read
    int id = blockIdx.x;
    {
        const int nb = gridDim.x;
        if (nb >= 16) {
            const int q = nb >> 3, r = nb & 7, x = id & 7, y = id >> 3;
            id = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + y;
        }
    }
    const int roles = (a.Cin / (BN == 128 ? 64 : 128)) * a.n_tiles * a.tap_groups;
    const int split = id / roles;
    id -= split * roles;
    const int tgp = id % a.tap_groups;                  // 5x5: the filter column kw of this workgroup
    id /= a.tap_groups;
    const int tn = id % a.n_tiles, cs = id / a.n_tiles;
    const int u0 = split * a.units_per_split, u1 = min(a.total_units, u0 + a.units_per_split);
    const int ci_tile = BN == 128 ? (wave >> 2) : (wave >> 1);
    const int co_tile = BN == 128 ? (wave & 3) : (wave & 1);
    const int kw0 = TW == 1 ? tgp : 0;                  // column shift applied when the patch is staged
    const int par_y = S2 ? tgp >> 1 : 0, par_x = S2 ? tgp & 1 : 0;
    const int OH = S2 ? a.H >> 1 : a.H, OW = S2 ? a.W >> 1 : a.W;

    const unsigned x_bytes = (unsigned)a.B * a.H * a.W * a.Cin * 2u, dy_bytes = (unsigned)a.B * OH * OW * a.N * 2u;
    const __amdgpu_buffer_rsrc_t rsrc_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.x), 0, x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_d = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16*>(a.dy), 0, dy_bytes, 0x00020000);
    constexpr unsigned OOB = 0x80000000u;
    // 16-byte chunk swizzles (per LDS row) that make the 4-row x 32-byte blocks of the transposing read conflict free
    auto p_swz = [](int pp) { return CIW == 64 ? 4 * ((pp >> 1) & 1) : 4 * (pp & 3); };
    auto d_swz = [](int m) { return BN == 64 ? 4 * ((m >> 1) & 1) : 4 * (m & 3); };

    // staging of one unit = PPASS + D_PASSES LDS-DMA pieces per thread; their global offsets first (unit_offsets), the pieces
    // themselves one at a time (dma_piece) so the main loop can place them between the MFMAs of successive patch rows
    constexpr int NPIECE = PPASS + D_PASSES;
    unsigned s_off_all[1][NPIECE];                      // (HALF: the own share's offsets, then -- once those pieces are issued -- the partner's)
    typedef std::integral_constant<int, 0> Own;
    typedef std::integral_constant<int, 1> Partner;
    auto unit_offsets = [&](int u, auto whoc) {
        constexpr int who = HALF ? decltype(whoc)::value : 0;
        const int t = who ? (int)(threadIdx.x ^ 256u) : (int)threadIdx.x;      // (who = 1: the thread of wave ^ 4 at this lane)
        unsigned (&s_off)[NPIECE] = s_off_all[0];
        const int n = u / a.units_per_img, ur = u - n * a.units_per_img;
        const int uy = ur / a.units_x, ux = ur - uy * a.units_x;
        const int y0 = uy * UH, x0 = ux * UW;
#pragma unroll
        for (int i = 0; i < PPASS; ++i) {
            const int pp = t / PCH + P_RPP * i;
            const int py = pp / PW, px = pp - py * PW;
            const int h = min(reflect_idx(S2 ? 2 * (y0 + py) + par_y - 1 : y0 - PAD + py, a.H), a.H - 1);
            const int w = min(reflect_idx(S2 ? 2 * (x0 + px) + par_x - 1 : x0 - PAD + kw0 + px, a.W), a.W - 1);
            const int lc = (t % PCH) ^ p_swz(pp);
            const unsigned off = ((unsigned)(((n * a.H + h) * a.W + w) << a.logCin) + (unsigned)(cs * CIW + lc * 8)) * 2u;
            s_off[i] = pp < PPIX ? off : OOB;
        }
#pragma unroll
        for (int p = 0; p < D_PASSES; ++p) {
            const int dr = t / DCH + D_RPP * p;
            const int lc = (t % DCH) ^ d_swz(dr);
            const unsigned pix = (unsigned)((n * OH + y0 + (dr >> 4)) * OW + x0 + (dr & 15));
            s_off[PPASS + p] = (pix * a.N + tn * BN + lc * 8) * 2u;
        }
    };
    auto dma_piece = [&](auto ic, int buf, auto whoc) {
        constexpr int i = decltype(ic)::value;
        constexpr int who = HALF ? decltype(whoc)::value : 0;
        const int wv = who ? (wave ^ 4) : wave;
        const unsigned off = s_off_all[0][i < NPIECE ? i : 0];
        if constexpr (i < PPASS)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_x, (__attribute__((address_space(3))) void*)(sP + buf * P_TILE + wv * 512 + i * P_RPP * CIW),
                                                     16, off, 0, 0, 0);
        else if constexpr (i < NPIECE)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_d, (__attribute__((address_space(3))) void*)(sD + buf * D_TILE + wv * 512 + (i - PPASS) * D_RPP * BN),
                                                     16, off, 0, 0, 0);
    };

    f32x16 acc[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    // transposing reads: lane 4q+p of a 16-lane group addresses pixel row q, channels 4p..4p+3 of a 4-pixel x 16-channel block
    const int li = lane & 15, gam = (lane >> 4) & 1, hi = lane >> 5;
    const int tq = li >> 2, tp = li & 3;
    const int pxl = 8 * hi + tq;                         // pixel (column) inside the row's 16, first half (+4: second)
    const int a_col = ci_tile * 32 + 16 * gam + 4 * tp;  // channel inside the CIW-channel patch
    const int d_col = co_tile * 32 + 16 * gam + 4 * tp;  // channel inside the BN-wide dY tile
    // Fragment reads are asm statements, waited for by hand (lgkmcnt counted, tied to the fragment registers so the MFMAs
    // stay behind them).  Left to hipcc as builtins they cost the whole staging latency: an LDS-DMA is a vector-memory
    // operation that WRITES LDS, the compiler cannot prove that a later LDS read does not alias it, and puts `s_waitcnt
    // vmcnt(0)` in front of the next ds_read -- the "prefetch" of the next unit was drained before the first MFMA of this
    // one (r03 ablation: 4 900 cycles per 5x5 unit = 3 070 of staging alone + 2 740 of reads and MFMAs alone, no overlap).
    const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
    auto tr_read = [](bf16x4& dst, unsigned addr) { asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(dst) : "v"(addr)); };
    auto a_frag = [&](bf16x4 (&v)[2], unsigned pbase, int r, int kw) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int pp = r * PW + pxl + 4 * half + kw;
            tr_read(v[half], pbase + (unsigned)(pp * CIW + (((a_col >> 3) ^ p_swz(pp)) << 3) + (a_col & 7)) * 2u);
        }
    };
    auto d_frag = [&](bf16x4 (&v)[2], unsigned dbase, int ks) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const int m = ks * 16 + pxl + 4 * half;
            tr_read(v[half], dbase + (unsigned)(m * BN + (((d_col >> 3) ^ d_swz(m)) << 3) + (d_col & 7)) * 2u);
        }
    };

    if (u0 < u1) {
        unit_offsets(u0, Own{});
        h16_for<NPIECE>([&](auto ic) { dma_piece(ic, 0, Own{}); });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if constexpr (PROBE) stamp[1] = __builtin_amdgcn_s_memtime();
        int buf = 0;
        for (int u = u0; u < u1; ++u) {
            // the next unit's pieces, issued as one burst here (SPREAD, lab only: one piece behind each patch row's MFMAs -- measured
            // 5-30 % slower: the pieces then queue behind a busy LDS instead of in front of it)
            const bool stage_next = u + 1 < u1 && !(DBG & 4);
            if constexpr (HALF) {
                if (stage_next && (wave >> 2) == ((u - u0) & 1)) {       // (wave-uniform)
                    unit_offsets(u + 1, Own{});
                    h16_for<NPIECE>([&](auto ic) { dma_piece(ic, buf ^ 1, Own{}); });
                    unit_offsets(u + 1, Partner{});
                    h16_for<NPIECE>([&](auto ic) { dma_piece(ic, buf ^ 1, Partner{}); });
                }
            } else if (stage_next) {
                unit_offsets(u + 1, Own{});
                if constexpr (!SPREAD) h16_for<NPIECE>([&](auto ic) { dma_piece(ic, buf ^ 1, Own{}); });
            }
            const unsigned pbase = lds0 + (unsigned)(buf * P_TILE) * 2u, dbase = lds0 + (unsigned)(2 * P_TILE + buf * D_TILE) * 2u;
            // Fragment reads run PF patch rows ahead of the MFMAs that use them (register rings).
            constexpr int PF = PFT < 0 ? 1 : PFT;     // PFT: the round-3 laboratory compared depths
            constexpr int FA = PF + 1, FB = TH + PF;     // ring sizes: x rows r..r+PF, dY rows r-TH+1..r+PF
            bf16x4 fa[FA][TW][2], fb[FB][2];
            auto row_reads = [](int r) { return r < PH ? 2 * TW + (r < UH ? 2 : 0) : 0; };     // ds_read instructions of row r
            auto fetch_row = [&](auto rc) {
                constexpr int r = decltype(rc)::value;
                if constexpr (r < PH) {
#pragma unroll
                    for (int kw = 0; kw < TW; ++kw) {
                        if constexpr (DBG & 2) { for (int e = 0; e < 4; ++e) fa[r % FA][kw][0][e] = fa[r % FA][kw][1][e] = (bf16)(float)(lane + r + kw); }
                        else a_frag(fa[r % FA][kw], pbase, r, kw);
                    }
                    if constexpr (r < UH) {
                        if constexpr (DBG & 2) { for (int e = 0; e < 4; ++e) fb[r % FB][0][e] = fb[r % FB][1][e] = (bf16)(float)(lane - r); }
                        else d_frag(fb[r % FB], dbase, r);
                    }
                }
            };
            h16_for<PF>([&](auto rc) { fetch_row(rc); });
            h16_for<PH>([&](auto rc) {
                constexpr int r = decltype(rc)::value;
                fetch_row(std::integral_constant<int, r + PF>{});
                // rows <= r have landed when at most the reads of rows r+1 .. r+PF are still in flight
                constexpr int younger = [&] { int n = 0; for (int q = r + 1; q <= r + PF; ++q) n += row_reads(q); return n < 15 ? n : 15; }();   // (4-bit counter)
                if constexpr (!(DBG & 2)) {
#pragma unroll
                    for (int kw = 0; kw < TW; ++kw)
                        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(fa[r % FA][kw][0]), "+v"(fa[r % FA][kw][1]) : "n"(younger));
                    if constexpr (r < UH) asm volatile("" : "+v"(fb[r % FB][0]), "+v"(fb[r % FB][1]));
                }
#pragma unroll
                for (int kh = 0; kh < TH; ++kh) {
                    const int ks = r - kh;               // x row r is tap row kh of output row ks
                    if (ks < 0 || ks >= UH) continue;
#pragma unroll
                    for (int kw = 0; kw < TW; ++kw) {
                        const bf16x8 av = __builtin_shufflevector(fa[r % FA][kw][0], fa[r % FA][kw][1], 0, 1, 2, 3, 4, 5, 6, 7);
                        const bf16x8 bv = __builtin_shufflevector(fb[ks % FB][0], fb[ks % FB][1], 0, 1, 2, 3, 4, 5, 6, 7);
                        if constexpr (DBG & 1) asm volatile("" ::"v"(av), "v"(bv));
                        else acc[kh * TW + kw] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[kh * TW + kw], 0, 0, 0);
                    }
                }
                if constexpr (SPREAD) {
                    static_assert(NPIECE <= PH, "one staging piece per patch row");
                    __builtin_amdgcn_sched_barrier(0);
                    if (stage_next) dma_piece(rc, buf ^ 1, Own{});
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            // every fragment of this unit is in registers (the last row's wait was lgkmcnt(0)): the barrier below releases the buffer
            if constexpr (!(DBG & 8)) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
            }
            buf ^= 1;
        }
    }
    if constexpr (PROBE) stamp[2] = __builtin_amdgcn_s_memtime();
    // slab[split][(tap*Cin + ci)][co]
    const int l31 = lane & 31;
    const int Ktot = KS * KS * a.Cin;
    float* out = a.slab + (size_t)split * Ktot * a.N;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int tap = S2 ? (2 * (j / TW) + par_y) * 4 + 2 * (j % TW) + par_x : (j / TW) * KS + kw0 + (j % TW);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = tap * a.Cin + cs * CIW + ci_tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
            out[(size_t)k * a.N + tn * BN + co_tile * 32 + l31] = acc[j][r];
        }
    }
    if constexpr (PROBE) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp[3] = __builtin_amdgcn_s_memtime();
        if (t == 0 && probe) {
            for (int k = 0; k < 4; ++k) probe[(size_t)blockIdx.x * 8 + k] = stamp[k];
            probe[(size_t)blockIdx.x * 8 + 4] = (unsigned long long)(u1 - u0);
        }
    }
#endif
}

// slab[s][(kh,kw,ci)][co] summed over s -> dw[co][ci][kh][kw] (state_dict layout, fp32), real channels only
__global__ void wgrad_halo_reduce_kernel(const float* __restrict__ slab, float* __restrict__ dw, int splits, int K, int N, int Cin,
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

// BN (channels of dY per workgroup) of a handled shape, 0 otherwise
// (K == 4 is the stride-2 form: H, W are the dimensions of x, the dY grid is H/2 x W/2)
int wgrad_halo_bn(int B, int H, int W, int Cin, int Cout, int K) {
    const int OH = K == 4 ? H / 2 : H, OW = K == 4 ? W / 2 : W;
    if (B <= 0 || (K != 3 && K != 5 && K != 4) || (K == 4 && ((H | W) & 1)) || OH < 8 || OW < 16 || (OH % 8) || (OW % 16) || Cin < 64 ||
        dwc_ilog2_exact(Cin) < 6 || (size_t)B * H * W * Cin * 2 >= 0x80000000ull || (size_t)B * OH * OW * Cout * 2 >= 0x80000000ull)
        return 0;
    if (Cout >= 128 && !(Cout % 128)) return 128;
    return (K == 5 && Cout >= 64 && !(Cout % 64) && !(Cin % 128)) ? 64 : 0;
}

void wgrad_halo_plan(int B, int H, int W, int Cin, int Cout, int K, int bn, int* splits, int* ups) {
    const int ciw = bn == 128 ? 64 : 128;
    const int roles = (Cin / ciw) * (Cout / bn) * (K == 3 ? 1 : K);        // (K == 4: the four input-pixel parities)
    const int units = K == 4 ? B * (H / 16) * (W / 32) : B * (H / 8) * (W / 16);
    // pixel splits: fill the 256 CUs (one workgroup each: LDS) in whole rounds - 260 workgroups cost two rounds, 240 one -
    // with at least 4 units per workgroup; among equally full launches the fewest splits (slab traffic)
    const int cus = 256;
    const int smax = units / 4 > 0 ? units / 4 : 1;
    int s = 1;
    double best = 0.0;
    for (int c = 1; c <= smax && c * roles <= 2 * cus; ++c) {
        const int wgs = c * roles, rounds = (wgs + cus - 1) / cus;
        const double fill = (double)wgs / (rounds * cus) / (rounds > 1 ? 1.05 : 1.0);   // a second round also pays a second prologue
        if (fill > best + 1e-9) best = fill, s = c;
    }
    *ups = (units + s - 1) / s;
    *splits = (units + *ups - 1) / *ups;
}

bool halo_ok(int B, int H, int W, int Cin, int Cout, int K) {
    return B > 0 && (K == 3 || K == 5) && H >= TB && W >= TB && !(H % TB) && !(W % TB) && Cin >= 64 && dwc_ilog2_exact(Cin) >= 6 &&
           Cout >= 64 && !(Cout & 7) && (size_t)B * H * W * Cin * 2 < 0x80000000ull && (size_t)B * H * W * Cout * 2 < 0x80000000ull;
}

}  // namespace

extern "C" {

/* 1 when dwc_bf16_conv2d_same_halo handles this stride-1, pad=(K-1)/2 convolution (else use dwc_bf16_conv2d_fwd /
 * dwc_bf16_conv2d_bwd_data_same): K in {3,5}, H and W multiples of 16, Cin a power of two >= 64, Cout a multiple of 8 >= 64 */
int dwc_bf16_conv2d_same_halo_ok(int B, int H, int W, int Cin, int Cout, int K) { return halo_ok(B, H, W, Cin, Cout, K) ? 1 : 0; }

/* y = act(conv(pad(x), w) + bias) for a stride-1 "same" convolution, pad rule reflect (forward, reference
 * networks.py:579-585) or zero (reflect == 0: the interior of the data gradient with w in dgrad layout, x := dY,
 * Cin := channels of dY, Cout := channels of dx; bias NULL, act NONE).  w prepared by dwc_bf16_weight_prepare_fwd / _dgrad
 * with cout_pad = Cout, cin_pad = Cin.  No scratch. */
int dwc_bf16_conv2d_same_halo_add(const void* x, const void* w_prepared, const float* bias, const void* add, void* y, int B, int H,
                                  int W, int Cin, int Cout, int K, int act, int reflect, void* stream);
int dwc_bf16_conv2d_same_halo(const void* x, const void* w_prepared, const float* bias, void* y, int B, int H, int W, int Cin,
                              int Cout, int K, int act, int reflect, void* stream) {
    return dwc_bf16_conv2d_same_halo_add(x, w_prepared, bias, nullptr, y, B, H, W, Cin, Cout, K, act, reflect, stream);
}

/* The same with `add` ([B,H,W,Cout] bf16 or NULL) added to the result behind bias and activation, as one more bf16 addition
 * (both operands to fp32, one rounding): the data gradient of a ResBlock's first convolution takes the gradient of the identity
 * branch (reference networks.py:521) here instead of in a separate pass over the tensor. */
int dwc_bf16_conv2d_same_halo_add(const void* x, const void* w_prepared, const float* bias, const void* add, void* y, int B, int H,
                                  int W, int Cin, int Cout, int K, int act, int reflect, void* stream) {
    if (!halo_ok(B, H, W, Cin, Cout, K)) return DWC_EINVAL;
    HaloArgs a;
    a.x = (const bf16*)x; a.w = (const bf16*)w_prepared; a.bias = bias; a.add = (const bf16*)add; a.y = (bf16*)y;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.logCin = dwc_ilog2_exact(Cin); a.N = Cout; a.K = K;
    a.Kp = (K * K * Cin + BK - 1) / BK * BK; a.act = act; a.reflect = reflect;
    a.blocks_x = W / TB; a.blocks_per_img = (H / TB) * (W / TB);
    hipStream_t st = (hipStream_t)stream;
    const int nblk = B * a.blocks_per_img;
    // the hand-scheduled 16x16x32 kernels (conv_halo16_bf16.inc)
#define HALO16_LAUNCH(KS, BN, WM, WN, PB)                                                                                  \
    do {                                                                                                                  \
        a.tiles_n = (Cout + BN - 1) / BN;                                                                                 \
        hipLaunchKernelGGL((conv_halo16_kernel<KS, BN, WM, WN, PB>), dim3(nblk * a.tiles_n), dim3(64 * WM * WN), 0, st, a, nullptr); \
    } while (0)
    // Two 4-wave workgroups per CU (256 pixels x 128 / x 64 channels, single patch buffer, <= 80 KB of LDS) where a full round
    // of them exists.  r03, same operands, 8-wave tile -> two workgroups per CU (bit-identical results): 5x5 256->128 B=384
    // 1950 -> 1698 us (52.8 -> 60.7 % of 2.5 PF), B=128 688 -> 603; 5x5 64->128 (data gradient) 758 -> 689; 3x3 256->256 B=384
    // 401 -> 387, B=128 139 -> 138; 5x5 128->64: 3 % slower (stays 8-wave).
    const bool duo = (long)nblk * (Cout / 64) >= 512;
    if (K == 3) {
        if (Cout > 128 && duo && Cout % 128 == 0) HALO16_LAUNCH(3, 128, 2, 2, 1);
        else if (Cout > 128) HALO16_LAUNCH(3, 256, 2, 4, 2);
        else if (Cout > 64) HALO16_LAUNCH(3, 128, 4, 2, 2);
        else HALO16_LAUNCH(3, 64, 4, 2, 2);
    } else {
        // (r06: the two-per-CU tile as 4 x 1 waves -- 4 block rows x 64 channels per wave, 8 fragment reads per 16 MFMAs instead of the 10 of
        // the 2 x 2 form / 6 per 8 of the 8-wave tile -- for the 64-output-channel layer, which ran the 8-wave tile: 128->64 at 128 x 128
        // B=128 698-705 -> 650-671 us, B=384 1852-1855 -> 1802-1820 (profiles/r06_halo5_tile_bench.txt); on the wider layers within
        // +-1.5 % of the 2 x 2 form, which stays.  DWC_H16_WM4=1: everywhere, =0: nowhere.)
        static const int wm4 = getenv("DWC_H16_WM4") ? atoi(getenv("DWC_H16_WM4")) : -1;
        if (duo && Cout % 64 == 0 && (wm4 == 1 || (wm4 < 0 && Cout == 64))) HALO16_LAUNCH(5, 64, 4, 1, 1);
        else if (Cout > 64 && duo && Cout % 64 == 0) HALO16_LAUNCH(5, 64, 2, 2, 1);
        else if (Cout > 128) HALO16_LAUNCH(5, 256, 2, 4, 1);
        else if (Cout > 64) HALO16_LAUNCH(5, 128, 4, 2, 1);
        else HALO16_LAUNCH(5, 64, 4, 2, 1);
    }
#undef HALO16_LAUNCH
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

/* DATA GRADIENT of a reflect-padded stride-1 "same" convolution in ONE launch (r06; reference networks.py:579-585 through autograd):
 * dx[B,H,W,Cin] = interior of the padded gradient image (zero-rule convolution of dy[B,H,W,Cout] with the rotated filter, w_dgrad =
 * dwc_bf16_weight_prepare_dgrad layout) + its border ring folded back by the reflect rule, the ring computed by the border tiles
 * themselves (conv_halo16_bf16.inc, RING) -- replaces dwc_bf16_conv2d_same_halo_add(reflect = 0) + dwc_bf16_conv2d_bwd_data_ring.
 * `add` ([B,H,W,Cin] bf16 or NULL) is added behind it as in dwc_bf16_conv2d_same_halo_add.  Handled: K = 3 with Cin a multiple of 128
 * above 128, K = 5 with Cin a multiple of 64 above 64, H and W multiples of 16 and >= 32, launches of at least 512 four-wave
 * workgroups (the forms two of which share a CU); _ok == 0: use the two-call form. */
int dwc_bf16_conv2d_bwd_data_same_fused_ok(int B, int H, int W, int Cin, int Cout, int K) {
    if (!halo_ok(B, H, W, Cout, Cin, K) || H < 2 * TB || W < 2 * TB) return 0;
    const long nblk = (long)B * (H / TB) * (W / TB);
    if (nblk * (Cin / 64) < 512) return 0;
    return (K == 3 ? (Cin > 128 && Cin % 128 == 0) : (Cin > 64 && Cin % 64 == 0)) ? 1 : 0;
}

int dwc_bf16_conv2d_bwd_data_same_fused(const void* dy, const void* w_dgrad, const void* add, void* dx, int B, int H, int W, int Cin,
                                        int Cout, int K, void* stream) {
    if (!dy || !w_dgrad || !dx || !dwc_bf16_conv2d_bwd_data_same_fused_ok(B, H, W, Cin, Cout, K)) return DWC_EINVAL;
    HaloArgs a;
    a.x = (const bf16*)dy; a.w = (const bf16*)w_dgrad; a.bias = nullptr; a.add = (const bf16*)add; a.y = (bf16*)dx;
    a.B = B; a.H = H; a.W = W; a.Cin = Cout; a.logCin = dwc_ilog2_exact(Cout); a.N = Cin; a.K = K;
    a.Kp = (K * K * Cout + BK - 1) / BK * BK; a.act = DWC_ACT_NONE; a.reflect = 0;
    a.blocks_x = W / TB; a.blocks_per_img = (H / TB) * (W / TB);
    hipStream_t st = (hipStream_t)stream;
    const int nblk = B * a.blocks_per_img;
    if (K == 3) {
        a.tiles_n = Cin / 128;
        hipLaunchKernelGGL((conv_halo16_kernel<3, 128, 2, 2, 1, 0, 0, 1>), dim3(nblk * a.tiles_n), dim3(256), 0, st, a, nullptr);
    } else {
        a.tiles_n = Cin / 64;
        hipLaunchKernelGGL((conv_halo16_kernel<5, 64, 2, 2, 1, 0, 0, 1>), dim3(nblk * a.tiles_n), dim3(256), 0, st, a, nullptr);
    }
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

/* INTERIOR of the data gradient of the 4x4, stride-2, reflect-pad-1 convolutions (conv_halo16_bf16.inc, S2 == 2): dy:[B,H/2,W/2,Cout]
 * bf16 -> the H x W pixels of dx:[B,H,W,Cin] (every pixel written, no accumulation); w_dgrad = dwc_bf16_weight_prepare_dgrad(KH = KW = 4,
 * stride 2, cout_pad = Cout, cin_pad = Cin): four class matrices [Cin][Kp].  The border ring of the padded gradient image still has to
 * be folded onto dx: dwc_bf16_conv2d_bwd_data_s2_ring.  H, W multiples of 32, Cout a power of two >= 64, Cin a multiple of 64. */
int dwc_bf16_conv2d_s2_halo_bwd_data_ok(int B, int H, int W, int Cin, int Cout) {
    return (B > 0 && H >= 32 && W >= 32 && !(H % 32) && !(W % 32) && Cout >= 64 && dwc_ilog2_exact(Cout) >= 6 && Cin >= 64 && !(Cin % 64) &&
            (size_t)B * H * W * Cin * 2 < 0x80000000ull && (size_t)B * (H / 2) * (W / 2) * Cout * 2 < 0x80000000ull &&
            (size_t)4 * Cin * 4 * Cout * 2 < 0x80000000ull) ? 1 : 0;
}

int dwc_bf16_conv2d_s2_halo_bwd_data(const void* dy, const void* w_dgrad, void* dx, int B, int H, int W, int Cin, int Cout, void* stream) {
    if (!dy || !w_dgrad || !dx || !dwc_bf16_conv2d_s2_halo_bwd_data_ok(B, H, W, Cin, Cout)) return DWC_EINVAL;
    HaloArgs a;
    a.x = (const bf16*)dy; a.w = (const bf16*)w_dgrad; a.bias = nullptr; a.add = nullptr; a.y = (bf16*)dx;
    a.B = B; a.H = H / 2; a.W = W / 2; a.Cin = Cout; a.logCin = dwc_ilog2_exact(Cout); a.N = Cin; a.K = 4;
    a.Kp = (4 * Cout + BK - 1) / BK * BK; a.act = DWC_ACT_NONE; a.reflect = 0;
    a.blocks_x = (W / 2) / TB; a.blocks_per_img = ((H / 2) / TB) * ((W / 2) / TB);
    hipStream_t st = (hipStream_t)stream;
    const int nblk = B * a.blocks_per_img;
    if (Cin % 128 == 0) {
        a.tiles_n = Cin / 128;
        hipLaunchKernelGGL((conv_halo16_kernel<2, 128, 2, 2, 1, 0, 2>), dim3(4 * nblk * a.tiles_n), dim3(256), 0, st, a, nullptr);
    } else {
        a.tiles_n = Cin / 64;
        hipLaunchKernelGGL((conv_halo16_kernel<2, 64, 2, 2, 1, 0, 2>), dim3(4 * nblk * a.tiles_n), dim3(256), 0, st, a, nullptr);
    }
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

/* (r06, ABI 8) dwc_bf16_conv2d_s2_halo_bwd_data with the border ring of the padded gradient image folded in by the launch itself (the
 * reflect-pad-1 adjoint: padded row 0 onto dx row 1, row H+1 onto H-2, columns alike; conv_halo16_bf16.inc RING, S2 == 2): the WHOLE data
 * gradient, no dwc_bf16_conv2d_bwd_data_s2_ring behind it.  Same shapes as dwc_bf16_conv2d_s2_halo_bwd_data_ok. */
int dwc_bf16_conv2d_s2_halo_bwd_data_fused(const void* dy, const void* w_dgrad, void* dx, int B, int H, int W, int Cin, int Cout, void* stream) {
    if (!dy || !w_dgrad || !dx || !dwc_bf16_conv2d_s2_halo_bwd_data_ok(B, H, W, Cin, Cout)) return DWC_EINVAL;
    HaloArgs a;
    a.x = (const bf16*)dy; a.w = (const bf16*)w_dgrad; a.bias = nullptr; a.add = nullptr; a.y = (bf16*)dx;
    a.B = B; a.H = H / 2; a.W = W / 2; a.Cin = Cout; a.logCin = dwc_ilog2_exact(Cout); a.N = Cin; a.K = 4;
    a.Kp = (4 * Cout + BK - 1) / BK * BK; a.act = DWC_ACT_NONE; a.reflect = 0;
    a.blocks_x = (W / 2) / TB; a.blocks_per_img = ((H / 2) / TB) * ((W / 2) / TB);
    hipStream_t st = (hipStream_t)stream;
    const int nblk = B * a.blocks_per_img;
    if (Cin % 128 == 0) {
        a.tiles_n = Cin / 128;
        hipLaunchKernelGGL((conv_halo16_kernel<2, 128, 2, 2, 1, 0, 2, 1>), dim3(4 * nblk * a.tiles_n), dim3(256), 0, st, a, nullptr);
    } else {
        a.tiles_n = Cin / 64;
        hipLaunchKernelGGL((conv_halo16_kernel<2, 64, 2, 2, 1, 0, 2, 1>), dim3(4 * nblk * a.tiles_n), dim3(256), 0, st, a, nullptr);
    }
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

/* 1 when dwc_bf16_conv2d_s2_halo handles this 4x4, stride-2, reflect-pad-1 convolution (else dwc_bf16_conv2d_fwd): H and W
 * multiples of 32 (16x16 output blocks), Cin a power of two >= 64, Cout a multiple of 64 */
int dwc_bf16_conv2d_s2_halo_ok(int B, int H, int W, int Cin, int Cout) {
    return (B > 0 && H >= 32 && W >= 32 && !(H % 32) && !(W % 32) && Cin >= 64 && dwc_ilog2_exact(Cin) >= 6 && Cout >= 64 && !(Cout % 64) &&
            (size_t)B * H * W * Cin * 2 < 0x80000000ull && (size_t)B * (H / 2) * (W / 2) * Cout * 2 < 0x80000000ull) ? 1 : 0;
}

/* y = act(conv4x4_stride2(reflect_pad1(x)) + bias) on bf16 NHWC tensors (reference networks.py:90,94,437, networks_v2.py:107-111):
 * x [B,H,W,Cin] -> y [B,H/2,W/2,Cout]; w prepared by dwc_bf16_weight_prepare_fwd (KH = KW = 4, cout_pad = Cout, cin_pad = Cin).
 * Halo form over the space-to-depth image (conv_halo16_bf16.inc, S2), two 4-wave workgroups per CU.  No scratch. */
int dwc_bf16_conv2d_s2_halo(const void* x, const void* w_prepared, const float* bias, void* y, int B, int H, int W, int Cin, int Cout,
                            int act, void* stream) {
    if (!x || !w_prepared || !y || !dwc_bf16_conv2d_s2_halo_ok(B, H, W, Cin, Cout)) return DWC_EINVAL;
    HaloArgs a;
    a.x = (const bf16*)x; a.w = (const bf16*)w_prepared; a.bias = bias; a.add = nullptr; a.y = (bf16*)y;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.logCin = dwc_ilog2_exact(Cin); a.N = Cout; a.K = 4;
    a.Kp = (16 * Cin + BK - 1) / BK * BK; a.act = act; a.reflect = 1;
    a.blocks_x = (W / 2) / TB; a.blocks_per_img = ((H / 2) / TB) * ((W / 2) / TB);
    hipStream_t st = (hipStream_t)stream;
    const int nblk = B * a.blocks_per_img;
    if (Cout % 128 == 0) {
        a.tiles_n = Cout / 128;
        hipLaunchKernelGGL((conv_halo16_kernel<2, 128, 2, 2, 1, 0, 1>), dim3(nblk * a.tiles_n), dim3(256), 0, st, a, nullptr);
    } else {
        a.tiles_n = Cout / 64;
        hipLaunchKernelGGL((conv_halo16_kernel<2, 64, 2, 2, 1, 0, 1>), dim3(nblk * a.tiles_n), dim3(256), 0, st, a, nullptr);
    }
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

size_t dwc_bf16_conv2d_wgrad_halo_ws_bytes(int B, int H, int W, int Cin, int Cout, int K) {
    const int bn = wgrad_halo_bn(B, H, W, Cin, Cout, K);
    if (!bn) return 0;
    int splits, ups;
    wgrad_halo_plan(B, H, W, Cin, Cout, K, bn, &splits, &ups);
    return (size_t)splits * K * K * Cin * Cout * sizeof(float);
}

/* dw (fp32 OIHW, [cout_real][cin_real][K][K]) of a reflect-padded stride-1 "same" K x K convolution (K = 3, 5) from x:[B,H,W,Cin]
 * and dy:[B,H,W,Cout], or of the 4x4 stride-2 reflect-pad-1 convolution (K = 4; dy:[B,H/2,W/2,Cout]) -- both bf16, halo form (see
 * wgrad_halo_kernel); dwc_bf16_conv2d_wgrad_halo_ws_bytes == 0 means the shape is not handled (use dwc_bf16_conv2d_bwd_weight). */
int dwc_bf16_conv2d_wgrad_halo(const void* x, const void* dy, float* dw_oihw, int B, int H, int W, int Cin, int Cout, int K,
                               int cin_real, int cout_real, void* ws, size_t ws_bytes, void* stream) {
    const int bn = wgrad_halo_bn(B, H, W, Cin, Cout, K);
    if (!bn || cin_real > Cin || cout_real > Cout) return DWC_EINVAL;
    int splits, ups;
    wgrad_halo_plan(B, H, W, Cin, Cout, K, bn, &splits, &ups);
    if (!ws || ws_bytes < (size_t)splits * K * K * Cin * Cout * sizeof(float)) return DWC_EWORKSPACE;
    WgradHaloArgs a;
    a.x = (const bf16*)x; a.dy = (const bf16*)dy; a.slab = (float*)ws;
    a.B = B; a.H = H; a.W = W; a.Cin = Cin; a.logCin = dwc_ilog2_exact(Cin); a.N = Cout;
    const int OH = K == 4 ? H / 2 : H, OW = K == 4 ? W / 2 : W;
    a.units_x = OW / 16; a.units_per_img = (OH / 8) * (OW / 16); a.total_units = B * a.units_per_img; a.units_per_split = ups;
    a.n_tiles = Cout / bn; a.tap_groups = K == 3 ? 1 : K;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid((Cin / (bn == 128 ? 64 : 128)) * a.n_tiles * a.tap_groups * splits);
#ifdef DWC_DEV_ABLATIONS      // timing-only ablation (WRONG results)
    static const int dbg = getenv("DWC_WGRAD_HALO_DBG") ? atoi(getenv("DWC_WGRAD_HALO_DBG")) : 0;
    if (K == 3 && dbg == 4) hipLaunchKernelGGL((wgrad_halo_kernel<3, 128, 4>), grid, dim3(512), 0, st, a);
    else
#endif
    // (HALF: measured at B=128 on one box, alternating: 5x5 256->128 735 / 739 -> 711 / 704 us, 128->64 742 / 750 -> 733 / 714 us; the
    // stride-2 form 160 -> 166 us and stays as it was, profiles/r06_wgrad_half_ab.txt.  DWC_WGRAD_HALF=0 restores the all-waves burst.)
    const char* he = getenv("DWC_WGRAD_HALF");
    const int half = he ? atoi(he) : 1;
    if (K == 3) hipLaunchKernelGGL((wgrad_halo_kernel<3, 128>), grid, dim3(512), 0, st, a);
    else if (K == 4) hipLaunchKernelGGL((wgrad_halo_kernel<4, 128>), grid, dim3(512), 0, st, a);
    else if (bn == 128 && half) hipLaunchKernelGGL((wgrad_halo_kernel<5, 128, 0, 0, -1, 0, 1>), grid, dim3(512), 0, st, a);
    else if (bn == 128) hipLaunchKernelGGL((wgrad_halo_kernel<5, 128>), grid, dim3(512), 0, st, a);
    else if (half) hipLaunchKernelGGL((wgrad_halo_kernel<5, 64, 0, 0, -1, 0, 1>), grid, dim3(512), 0, st, a);
    else hipLaunchKernelGGL((wgrad_halo_kernel<5, 64>), grid, dim3(512), 0, st, a);
    DWC_LAUNCH_CHECK();
    const size_t total = (size_t)K * K * Cin * Cout;
    hipLaunchKernelGGL(wgrad_halo_reduce_kernel, dim3((total + 255) / 256), dim3(256), 0, st, (const float*)ws, dw_oihw, splits,
                       K * K * Cin, Cout, Cin, K * K, cin_real, cout_real);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

}  // extern "C"
