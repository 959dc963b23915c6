// Laboratory: phase offset between the two workgroups that share a CU in the two-workgroups-per-CU ("duo") forms of the bf16
// halo kernel (conv_halo16_kernel<.., WM*WN = 4, PB = 1>).  Development aid, not part of libdwcgan_hip.so.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I dwc-gan_amd/csrc benchmarks/stagger_lab.hip -o benchmarks/bin/stagger_lab
//   benchmarks/bin/stagger_lab [rounds]
// For each layer shape of BASELINE configs[2]: the duo kernel with a.stagger = 0 and a sweep of delays (10 ns ticks) applied once
// to the CU's second workgroup in the first round, variants interleaved in ONE process on the same RANDOM bf16 operands
// (cdna_hip_programming.md rules 24/25), results compared bit for bit with stagger 0; then the timeline probe (prologue / main
// loop / epilogue cycles of wave 0, shader clock) with and without the offset, and how HW_ID's TG_ID / WAVE_ID pair up the
// first-round workgroups on a CU.
#include "../dwc-gan_amd/csrc/conv_halo_bf16.hip"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static unsigned short f2bf(float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    return (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16);
}

struct Shape { const char* name; int B, H, Cin, Cout, K; };

int main(int argc, char** argv) {
    const int rounds = argc > 1 ? atoi(argv[1]) : 10;
    const Shape shapes[] = {
        {"3x3 256>256 @32 B128", 128, 32, 256, 256, 3}, {"3x3 256>256 @32 B256", 256, 32, 256, 256, 3}, {"3x3 256>256 @32 B384", 384, 32, 256, 256, 3},
        {"5x5 256>128 @64 B128", 128, 64, 256, 128, 5}, {"5x5 256>128 @64 B384", 384, 64, 256, 128, 5},
        {"5x5 64>128 @128 B128 (dgrad)", 128, 128, 64, 128, 5},
    };
    const int sweeps[] = {0, 2000, 0};            // (r04: a phase offset changes nothing measurable -- kept as a two-point check)
    const int NS = sizeof(sweeps) / sizeof(int);
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
        void *dx, *dw, *dy0, *dy1;
        float* db;
        CK(hipMalloc(&dx, nx * 2)); CK(hipMalloc(&dw, nw * 2)); CK(hipMalloc(&dy0, ny * 2)); CK(hipMalloc(&dy1, ny * 2));
        CK(hipMalloc(&db, s.Cout * 4));
        CK(hipMemcpy(dx, hx.data(), nx * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(dw, hw.data(), nw * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(db, hb.data(), s.Cout * 4, hipMemcpyHostToDevice));
        const double flops = 2.0 * s.B * s.H * s.H * (double)s.Cout * s.Cin * s.K * s.K;
        HaloArgs a;
        a.x = (const bf16*)dx; a.w = (const bf16*)dw; a.bias = db; a.add = nullptr;
        a.B = s.B; a.H = s.H; a.W = s.H; a.Cin = s.Cin; a.logCin = dwc_ilog2_exact(s.Cin); a.N = s.Cout; a.K = s.K;
        a.Kp = Kp; a.act = DWC_ACT_RELU; a.reflect = 1;
        a.blocks_x = s.H / 16; a.blocks_per_img = (s.H / 16) * (s.H / 16);
        const int nblk = s.B * a.blocks_per_img;
        const int BN = s.K == 3 ? 128 : 64;
        a.tiles_n = s.Cout / BN;
        const int grid = nblk * a.tiles_n;
        auto launch_abl = [&](void* y, auto PR, unsigned long long* probe) {      // PROBE bits 2 (no stores) / 4 (non-temporal stores) [+1: timeline]
            constexpr int P = decltype(PR)::value;
            a.y = (bf16*)y;
            a.stagger = 0;
            if (s.K == 3) hipLaunchKernelGGL((conv_halo16_kernel<3, 128, 2, 2, 1, P>), dim3(grid), dim3(256), 0, st, a, probe);
            else hipLaunchKernelGGL((conv_halo16_kernel<5, 64, 2, 2, 1, P>), dim3(grid), dim3(256), 0, st, a, probe);
            CK(hipGetLastError());
        };
        auto launch = [&](void* y, int stagger, unsigned long long* probe) {
            a.y = (bf16*)y;
            a.stagger = stagger;
            if (probe) {
                if (s.K == 3) hipLaunchKernelGGL((conv_halo16_kernel<3, 128, 2, 2, 1, 1>), dim3(grid), dim3(256), 0, st, a, probe);
                else hipLaunchKernelGGL((conv_halo16_kernel<5, 64, 2, 2, 1, 1>), dim3(grid), dim3(256), 0, st, a, probe);
            } else {
                if (s.K == 3) hipLaunchKernelGGL((conv_halo16_kernel<3, 128, 2, 2, 1>), dim3(grid), dim3(256), 0, st, a, nullptr);
                else hipLaunchKernelGGL((conv_halo16_kernel<5, 64, 2, 2, 1>), dim3(grid), dim3(256), 0, st, a, nullptr);
            }
            CK(hipGetLastError());
        };
        std::vector<float> tms[NS];
        for (int r = 0; r < rounds; ++r)
            for (int v = 0; v < NS; ++v) {
                CK(hipEventRecord(e0, st));
                launch(v == 0 ? dy0 : dy1, sweeps[v], nullptr);
                CK(hipEventRecord(e1, st));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (r >= 2) tms[v].push_back(ms);
            }
        std::vector<unsigned short> y0(ny), y1(ny);
        CK(hipMemcpy(y0.data(), dy0, ny * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(y1.data(), dy1, ny * 2, hipMemcpyDeviceToHost));
        size_t ndiff = 0;
        for (size_t i = 0; i < ny; ++i) ndiff += y0[i] != y1[i];
        printf("%-30s %8.1f GFLOP, %5d workgroups (%.2f rounds of 512) | results differ in %zu elements\n", s.name, flops / 1e9, grid, grid / 512.0, ndiff);
        for (int v = 0; v < NS; ++v) {
            std::sort(tms[v].begin(), tms[v].end());
            const double med = tms[v][tms[v].size() / 2], mn = tms[v][0];
            printf("    stagger %5d ticks (%5.1f us): med %8.1f us  min %8.1f us  (%5.1f%% of 2.5PF)\n", sweeps[v], sweeps[v] / 100.0, med * 1e3, mn * 1e3,
                   flops / (med * 1e-3) / 2.5e15 * 100);
        }
        // what the result stores cost: the same kernel without them / with non-temporal stores (interleaved with the plain one)
        {
            std::vector<float> ta[3];
            for (int r = 0; r < rounds; ++r)
                for (int v = 0; v < 3; ++v) {
                    CK(hipEventRecord(e0, st));
                    if (v == 0) launch(dy1, 0, nullptr);
                    else if (v == 1) launch_abl(dy1, std::integral_constant<int, 2>{}, nullptr);
                    else launch_abl(dy1, std::integral_constant<int, 4>{}, nullptr);
                    CK(hipEventRecord(e1, st));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    if (r >= 2) ta[v].push_back(ms);
                }
            const char* nm[3] = {"plain stores", "NO global stores (ablation)", "non-temporal stores"};
            for (int v = 0; v < 3; ++v) {
                std::sort(ta[v].begin(), ta[v].end());
                const double med = ta[v][ta[v].size() / 2];
                printf("    %-30s med %8.1f us  min %8.1f us  (%5.1f%% of 2.5PF)\n", nm[v], med * 1e3, ta[v][0] * 1e3, flops / (med * 1e-3) / 2.5e15 * 100);
            }
            CK(hipMemcpy(y1.data(), dy1, ny * 2, hipMemcpyDeviceToHost));
            size_t nd = 0;
            for (size_t i = 0; i < ny; ++i) nd += y0[i] != y1[i];
            printf("    non-temporal result differs from the plain one in %zu elements\n", nd);
        }
        // timeline with / without the offset
        for (int stg : {0, 2000}) {
            unsigned long long* dp;
            CK(hipMalloc(&dp, (size_t)grid * 64));
            CK(hipMemset(dp, 0, (size_t)grid * 64));
            for (int rep = 0; rep < 3; ++rep) launch(dy1, stg, dp);
            CK(hipStreamSynchronize(st));
            std::vector<unsigned long long> hp((size_t)grid * 8);
            CK(hipMemcpy(hp.data(), dp, (size_t)grid * 64, hipMemcpyDeviceToHost));
            std::vector<double> pro, loop, epi, clk, issue, drain;
            unsigned long long tmin = ~0ull, tmax = 0;
            std::map<unsigned, int> tg_hist, wave_hist;
            std::map<unsigned, std::vector<unsigned>> cu_tgs;     // (se, sh, cu) -> TG_IDs of the first-round workgroups seen there
            for (int b = 0; b < grid; ++b) {
                const unsigned long long* q = &hp[(size_t)b * 8];
                pro.push_back((double)(q[1] - q[0])); loop.push_back((double)(q[2] - q[1])); epi.push_back((double)(q[3] - q[2]));
                issue.push_back((double)(q[7] - q[2])); drain.push_back((double)(q[3] - q[7]));
                if (q[5] > q[4]) clk.push_back((double)(q[3] - q[0]) / (double)(q[5] - q[4]) * 100.0);
                tmin = std::min(tmin, q[4]); tmax = std::max(tmax, q[5]);
                const unsigned hw = (unsigned)q[6];
                if (b < 512) {
                    tg_hist[(hw >> 16) & 15]++;
                    wave_hist[hw & 15]++;
                }
            }
            auto med = [](std::vector<double>& v) { std::sort(v.begin(), v.end()); return v.empty() ? 0.0 : v[v.size() / 2]; };
            const int nsteps = (s.Cin / 64) * s.K * s.K;
            printf("    timeline stagger %d (cycles, median over %d workgroups): prologue %.0f | main loop %.0f = %.0f per step | epilogue %.0f = %.0f until the last store is issued + %.0f drain | shader clock %.0f MHz | span %.1f us\n",
                   stg, grid, med(pro), med(loop), med(loop) / nsteps, med(epi), med(issue), med(drain), med(clk), (double)(tmax - tmin) / 100.0);
            if (stg == 0) {
                printf("    first 512 workgroups: TG_ID histogram");
                for (auto& kv : tg_hist) printf(" %u:%d", kv.first, kv.second);
                printf(" | WAVE_ID (wave 0) histogram");
                for (auto& kv : wave_hist) printf(" %u:%d", kv.first, kv.second);
                printf("\n");
            }
            CK(hipFree(dp));
        }
        fflush(stdout);
        CK(hipFree(dx)); CK(hipFree(dw)); CK(hipFree(dy0)); CK(hipFree(dy1)); CK(hipFree(db));
    }
    return 0;
}
