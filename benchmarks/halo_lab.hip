// Laboratory for the bf16 halo-tiled convolution kernels (development aid, not part of libdwcgan_hip.so).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dwc-gan_amd/csrc benchmarks/halo_lab.hip -o benchmarks/bin/halo_lab
//   benchmarks/bin/halo_lab [rounds]
// Runs the layer shapes of BASELINE configs[2] through the compiler-scheduled 32x32x16 kernels (DWC_HALO16=0) and the
// hand-scheduled 16x16x32 kernels (DWC_HALO16=1, the default) of dwc_bf16_conv2d_same_halo on the same RANDOM bf16 operands
// (zero-filled operands clock higher: cdna_hip_programming.md rule 25), variants interleaved in ONE process (rule 24), checks
// that the two agree, and prints median / min microseconds and TFLOP/s against the 2.5 PF dense bf16 roof.
#include "../dwc-gan_amd/csrc/conv_halo_bf16.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static unsigned short f2bf(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    return (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
}
static float bf2f(unsigned short h) {
    unsigned u = (unsigned)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

struct Shape { const char* name; int B, H, Cin, Cout, K; };

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 12;
    const Shape shapes[] = {
        {"3x3 256>256 @32 B128", 128, 32, 256, 256, 3}, {"3x3 256>256 @32 B384", 384, 32, 256, 256, 3},
        {"5x5 256>128 @64 B128", 128, 64, 256, 128, 5}, {"5x5 256>128 @64 B384", 384, 64, 256, 128, 5},
        {"5x5 128>64 @128 B128", 128, 128, 128, 64, 5}, {"5x5 128>64 @128 B384", 384, 128, 128, 64, 5},
        {"5x5 128>256 @64 B128 (dgrad)", 128, 64, 128, 256, 5}, {"5x5 64>128 @128 B128 (dgrad)", 128, 128, 64, 128, 5},
        {"3x3 64>128 @32 B8", 8, 32, 64, 128, 3}, {"3x3 128>64 @48 B3", 3, 48, 128, 64, 3},
    };
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    srand(1234);
    for (const Shape& s : shapes) {
        const size_t nx = (size_t)s.B * s.H * s.H * s.Cin, ny = (size_t)s.B * s.H * s.H * s.Cout;
        const int Kp = (s.K * s.K * s.Cin + 63) / 64 * 64;
        const size_t nw = (size_t)s.Cout * Kp;
        std::vector<unsigned short> hx(nx), hw(nw);
        for (auto& v : hx) v = f2bf((float)rand() / RAND_MAX * 2.f - 1.f);
        for (auto& v : hw) v = f2bf(((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f);
        std::vector<float> hb(s.Cout);
        for (auto& v : hb) v = (float)rand() / RAND_MAX - 0.5f;
        void *dx, *dw, *dy[3];
        float* db;
        CK(hipMalloc(&dx, nx * 2)); CK(hipMalloc(&dw, nw * 2)); CK(hipMalloc(&dy[0], ny * 2)); CK(hipMalloc(&dy[1], ny * 2)); CK(hipMalloc(&dy[2], ny * 2));
        CK(hipMalloc(&db, s.Cout * 4));
        CK(hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, hb.data(), s.Cout * 4, hipMemcpyHostToDevice));
        const double flops = 2.0 * s.B * s.H * s.H * (double)s.Cout * s.Cin * s.K * s.K;
        std::vector<float> tms[3];
        double maxdiff[2] = {0, 0}, maxdiff_p[2] = {0, 0};
        for (int reflect = 1; reflect >= 0; --reflect) {
            for (int v = 0; v < 3; ++v) CK(hipMemset(dy[v], 0xFF, ny * 2));
            // the dispatcher reads DWC_HALO16 once (static): call the kernels directly instead
            for (int r = 0; r < (reflect ? rounds : 1); ++r)
                for (int v = 0; v < 3; ++v) {
                    HaloArgs a;
                    a.x = (const bf16*)dx; a.w = (const bf16*)dw; a.bias = reflect ? db : nullptr; a.add = nullptr; a.y = (bf16*)dy[v];
                    a.B = s.B; a.H = s.H; a.W = s.H; a.Cin = s.Cin; a.logCin = dwc_ilog2_exact(s.Cin); a.N = s.Cout; a.K = s.K;
                    a.Kp = Kp; a.act = reflect ? DWC_ACT_RELU : DWC_ACT_NONE; a.reflect = reflect;
                    a.blocks_x = s.H / 16; a.blocks_per_img = (s.H / 16) * (s.H / 16);
                    const int nblk = s.B * a.blocks_per_img;
                    CK(hipEventRecord(e0, st));
#define L32(KS, BN, WM, WN, TM, TN) do { a.tiles_n = (s.Cout + BN - 1) / BN; hipLaunchKernelGGL((conv_halo_kernel<KS, BN, WM, WN, TM, TN, (KS == 3 ? 2 : 1)>), dim3(nblk * a.tiles_n), dim3(512), 0, st, a); } while (0)
#define L16(KS, BN, WM, WN, PB) do { a.tiles_n = (s.Cout + BN - 1) / BN; hipLaunchKernelGGL((conv_halo16_kernel<KS, BN, WM, WN, PB>), dim3(nblk * a.tiles_n), dim3(512), 0, st, a); } while (0)
                    if (v == 0) {
                        if (s.K == 3) { if (s.Cout > 128) L32(3, 256, 2, 4, 4, 2); else if (s.Cout > 64) L32(3, 128, 4, 2, 2, 2); else L32(3, 64, 4, 2, 2, 1); }
                        else { if (s.Cout > 128) L32(5, 256, 2, 4, 4, 2); else if (s.Cout > 64) L32(5, 128, 4, 2, 2, 2); else L32(5, 64, 4, 2, 2, 1); }
                    } else if (v == 1) {
                        if (s.K == 3) { if (s.Cout > 128) L16(3, 256, 2, 4, 2); else if (s.Cout > 64) L16(3, 128, 4, 2, 2); else L16(3, 64, 4, 2, 2); }
                        else { if (s.Cout > 128) L16(5, 256, 2, 4, 1); else if (s.Cout > 64) L16(5, 128, 4, 2, 1); else L16(5, 64, 4, 2, 1); }
                    } else {
#define L16D(KS, BN, WM, WN) do { a.tiles_n = (s.Cout + BN - 1) / BN; hipLaunchKernelGGL((conv_halo16_kernel<KS, BN, WM, WN, 1>), dim3(nblk * a.tiles_n), dim3(256), 0, st, a, nullptr); } while (0)
                        // two 4-wave workgroups per CU: 256 pixels x 128 channels (wave tile 128 x 64) / x 64 channels (128 x 32)
                        if (s.K == 3) { if (s.Cout > 64) L16D(3, 128, 2, 2); else L16D(3, 64, 2, 2); }
                        else L16D(5, 64, 2, 2);      // (5x5 x 128 channels: 85 KB of LDS, one workgroup per CU only)
                    }
                    CK(hipGetLastError());
                    CK(hipEventRecord(e1, st));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (reflect && r >= 2) tms[v].push_back(ms);
                }
            std::vector<unsigned short> y0(ny), y1(ny), y2(ny);
            CK(hipMemcpy(y0.data(), dy[0], ny * 2, hipMemcpyDeviceToHost));
            CK(hipMemcpy(y1.data(), dy[1], ny * 2, hipMemcpyDeviceToHost));
            CK(hipMemcpy(y2.data(), dy[2], ny * 2, hipMemcpyDeviceToHost));
            double md = 0, md2 = 0, scale = 0;
            size_t nbad = 0;
            for (size_t i = 0; i < ny; ++i) {
                const double a0 = bf2f(y0[i]), a1 = bf2f(y1[i]), a2 = bf2f(y2[i]);
                scale = std::max(scale, std::fabs(a0));
                const double d = std::fabs(a0 - a1), d2 = std::fabs(a0 - a2);
                if (!(d <= 1e30) || !(d2 <= 1e30)) ++nbad;
                md = std::max(md, d);
                md2 = std::max(md2, d2);
            }
            maxdiff[reflect] = md / (scale > 0 ? scale : 1);
            maxdiff_p[reflect] = md2 / (scale > 0 ? scale : 1);
            if (nbad) printf("  !! %zu non-finite differences (reflect=%d)\n", nbad, reflect);
        }
        auto stat = [&](std::vector<float>& t, double& med, double& mn) {
            std::sort(t.begin(), t.end());
            med = t[t.size() / 2];
            mn = t[0];
        };
        double m0, n0, m1, n1, m2, n2;
        stat(tms[0], m0, n0);
        stat(tms[1], m1, n1);
        stat(tms[2], m2, n2);
        printf("%-30s %8.1f GFLOP | 32x32x16: med %8.1f us (%5.1f%% of 2.5PF) | 16x16x32 hand: med %8.1f us min %8.1f (%5.1f%%) x%.3f | duo (2 WG/CU): med %8.1f us min %8.1f (%5.1f%%) x%.3f | maxdiff/scale hand %.1e/%.1e duo %.1e/%.1e\n",
               s.name, flops / 1e9, m0 * 1e3, flops / (m0 * 1e-3) / 2.5e15 * 100, m1 * 1e3, n1 * 1e3,
               flops / (m1 * 1e-3) / 2.5e15 * 100, m0 / m1, m2 * 1e3, n2 * 1e3, flops / (m2 * 1e-3) / 2.5e15 * 100, m0 / m2, maxdiff[1], maxdiff[0],
               maxdiff_p[1], maxdiff_p[0]);
        fflush(stdout);
        // timeline of the hand-scheduled kernel (diagnostic instantiation; shader cycles of wave 0, median over workgroups)
        if (s.B >= 128 && (s.K == 3 || s.Cout <= 128)) {
            HaloArgs a;
            a.x = (const bf16*)dx; a.w = (const bf16*)dw; a.bias = db; a.add = nullptr; a.y = (bf16*)dy[1];
            a.B = s.B; a.H = s.H; a.W = s.H; a.Cin = s.Cin; a.logCin = dwc_ilog2_exact(s.Cin); a.N = s.Cout; a.K = s.K;
            a.Kp = Kp; a.act = DWC_ACT_RELU; a.reflect = 1;
            a.blocks_x = s.H / 16; a.blocks_per_img = (s.H / 16) * (s.H / 16);
            const int nblk = s.B * a.blocks_per_img;
            int tiles = 1;
            unsigned long long* dp;
#define LP(KS, BN, WM, WN, PB) do { a.tiles_n = tiles = (s.Cout + BN - 1) / BN; CK(hipMalloc(&dp, (size_t)nblk * tiles * 64)); CK(hipMemset(dp, 0, (size_t)nblk * tiles * 64)); for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL((conv_halo16_kernel<KS, BN, WM, WN, PB, 1>), dim3(nblk * tiles), dim3(512), 0, st, a, dp); } while (0)
            if (s.K == 3) { if (s.Cout > 128) LP(3, 256, 2, 4, 2); else if (s.Cout > 64) LP(3, 128, 4, 2, 2); else LP(3, 64, 4, 2, 2); }
            else { if (s.Cout > 64) LP(5, 128, 4, 2, 1); else LP(5, 64, 4, 2, 1); }
            CK(hipStreamSynchronize(st));
            const size_t nb = (size_t)nblk * tiles;
            std::vector<unsigned long long> hp(nb * 8);
            CK(hipMemcpy(hp.data(), dp, nb * 64, hipMemcpyDeviceToHost));
            std::vector<double> pro, loop, epi, clk;
            unsigned long long tmin = ~0ull, tmax = 0;
            for (size_t b = 0; b < nb; ++b) {
                const unsigned long long* q = &hp[b * 8];
                pro.push_back((double)(q[1] - q[0])); loop.push_back((double)(q[2] - q[1])); epi.push_back((double)(q[3] - q[2]));
                if (q[5] > q[4]) clk.push_back((double)(q[3] - q[0]) / (double)(q[5] - q[4]) * 100.0);     // MHz (100 MHz realtime)
                tmin = std::min(tmin, q[4]); tmax = std::max(tmax, q[5]);
            }
            auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
            const int nsteps = (s.Cin / 64) * s.K * s.K;
            printf("    timeline (cycles, median over %zu workgroups): prologue %.0f | main loop %.0f = %.0f per (tap,slab) step | epilogue %.0f | "
                   "shader clock %.0f MHz | launch span %.1f us\n", nb, med(pro), med(loop), med(loop) / nsteps, med(epi), med(clk),
                   (double)(tmax - tmin) / 100.0);
            CK(hipFree(dp));
        }
        // weight gradient of the same layer (compiler-scheduled halo form): time + timeline
        if (s.B >= 128 && wgrad_halo_bn(s.B, s.H, s.H, s.Cin, s.Cout, s.K)) {
            const int bn = wgrad_halo_bn(s.B, s.H, s.H, s.Cin, s.Cout, s.K);
            int splits, ups;
            wgrad_halo_plan(s.B, s.H, s.H, s.Cin, s.Cout, s.K, bn, &splits, &ups);
            float* slab;
            CK(hipMalloc(&slab, (size_t)splits * s.K * s.K * s.Cin * s.Cout * 4));
            WgradHaloArgs wa;
            wa.x = (const bf16*)dx; wa.dy = (const bf16*)dy[1]; wa.slab = slab;
            wa.B = s.B; wa.H = s.H; wa.W = s.H; wa.Cin = s.Cin; wa.logCin = dwc_ilog2_exact(s.Cin); wa.N = s.Cout;
            wa.units_x = s.H / 16; wa.units_per_img = (s.H / 8) * (s.H / 16); wa.total_units = s.B * wa.units_per_img; wa.units_per_split = ups;
            wa.n_tiles = s.Cout / bn; wa.tap_groups = s.K == 3 ? 1 : s.K;
            const int grid = (s.Cin / (bn == 128 ? 64 : 128)) * wa.n_tiles * wa.tap_groups * splits;
            unsigned long long* dp;
            CK(hipMalloc(&dp, (size_t)grid * 64));
            const size_t slab_n = (size_t)splits * s.K * s.K * s.Cin * s.Cout;
            std::vector<float> ref_slab;
            auto run_pf = [&](auto pfc, auto dbgc, auto sprc) {
                constexpr int PF = decltype(pfc)::value, DBGV = decltype(dbgc)::value, SPR = decltype(sprc)::value;
                CK(hipMemset(dp, 0, (size_t)grid * 64));
                std::vector<float> tw;
                for (int rep = 0; rep < 8; ++rep) {
                    CK(hipEventRecord(e0, st));
                    if (s.K == 3) hipLaunchKernelGGL((wgrad_halo_kernel<3, 128, DBGV, 1, PF, SPR>), dim3(grid), dim3(512), 0, st, wa, dp);
                    else if (bn == 128) hipLaunchKernelGGL((wgrad_halo_kernel<5, 128, DBGV, 1, PF, SPR>), dim3(grid), dim3(512), 0, st, wa, dp);
                    else hipLaunchKernelGGL((wgrad_halo_kernel<5, 64, DBGV, 1, PF, SPR>), dim3(grid), dim3(512), 0, st, wa, dp);
                    CK(hipEventRecord(e1, st));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (rep >= 2) tw.push_back(ms);
                }
                std::sort(tw.begin(), tw.end());
                std::vector<unsigned long long> hp((size_t)grid * 8);
                CK(hipMemcpy(hp.data(), dp, (size_t)grid * 64, hipMemcpyDeviceToHost));
                std::vector<float> hs(slab_n);
                CK(hipMemcpy(hs.data(), slab, slab_n * 4, hipMemcpyDeviceToHost));
                double md = 0.0;
                if (DBGV) md = -1.0;
                else if (ref_slab.empty()) ref_slab = hs;
                else for (size_t i = 0; i < slab_n; ++i) md = std::max(md, (double)fabsf(hs[i] - ref_slab[i]));
                std::vector<double> pro, loop, epi, per;
                for (int b = 0; b < grid; ++b) {
                    const unsigned long long* q = &hp[(size_t)b * 8];
                    pro.push_back((double)(q[1] - q[0])); loop.push_back((double)(q[2] - q[1])); epi.push_back((double)(q[3] - q[2]));
                    if (q[4]) per.push_back((double)(q[2] - q[1]) / (double)q[4]);
                }
                auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
                const int taps_wg = s.K == 3 ? 9 : 5;
                printf("    wgrad (halo form, ablation %d, fragment prefetch %d rows, staging %s, %d workgroups, %d splits): med %.1f us (%.1f%% of 2.5PF) | cycles: prologue %.0f | "
                       "loop %.0f = %.0f per 128-pixel unit (MFMA issue per SIMD: %d) | store %.0f | maxdiff vs prefetch 0: %.1e\n", DBGV, PF, SPR ? "spread" : "burst", grid, splits,
                       tw[tw.size() / 2] * 1e3, flops / (tw[tw.size() / 2] * 1e-3) / 2.5e15 * 100, med(pro), med(loop), med(per), 8 * taps_wg * 32 * 2,
                       med(epi), md);
            };
            typedef std::integral_constant<int, 0> Z;
            typedef std::integral_constant<int, 1> One;
            run_pf(Z{}, Z{}, Z{});
            run_pf(One{}, Z{}, Z{});
            run_pf(Z{}, Z{}, One{});
            run_pf(One{}, Z{}, One{});
            run_pf(std::integral_constant<int, 2>{}, Z{}, One{});
            run_pf(One{}, One{}, One{});                                // no MFMA
            run_pf(One{}, std::integral_constant<int, 4>{}, One{});     // no staging inside the loop
            CK(hipFree(dp)); CK(hipFree(slab));
        }
        CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(dy[0])); CK(hipFree(dy[1])); CK(hipFree(dy[2])); CK(hipFree(db));
    }
    return 0;
}
