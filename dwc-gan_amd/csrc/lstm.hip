// Recurrent part of the text encoder's packed bi-LSTM (reference networks_v2.py:199-203,226-233: nn.LSTM over a
// pack_padded_sequence) on the fp32 matrix cores.
//
// The input projections of all time steps are one library GEMM done by the caller; what is sequential is
//     gates_t = xproj_t + h_{t-1} W_hh^T ;  c_t = f*c_{t-1} + i*g ;  h_t = o*tanh(c_t)
// Stock ROCm runs this as two tiny launches per step per direction (a rocBLAS GEMM and a cell kernel, ~11 us).
// Here one launch per step serves BOTH directions: workgroup (g, d) owns 16 hidden units of direction d and computes
// their four gate tiles with v_mfma_f32_16x16x4_f32 ([B x H] . [H x 16], batch rows on M; each wave a quarter of the
// contraction), the cell update is fused behind an LDS exchange, and the kernel boundary is the only synchronisation
// (no spin barriers).
//
// Packed-sequence semantics by masking: sample b is active at step t iff t < len[b].  Inactive positions of out/c are
// written as zeros, so "previous state" is simply the neighbouring time slot (t-1 forward, t+1 reverse): a reverse
// sequence starts from zeros at its own last token exactly as nn.LSTM does on packed input.
#include "dwc_common.h"

namespace {

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// K-labelling shared by both operands: lane (r = lane & 15, j = lane >> 4) takes k = 16*q + 4*j + i for the i-th MFMA
// of block q, so that each lane fetches ONE 16-byte vector per block and operand.  Blocks [q0, q1) are processed in
// chunks whose loads are all issued before the first MFMA (addresses clamped, out-of-range operands zeroed afterwards):
// a step kernel is a chain of L2 latencies, so what matters is how many loads are in flight, not how many are issued.
// NB operand sets: acc[n][m] += A[row0 + 16m + r][k] * Bn[k] for NB "B" rows (the four gates in the forward step).
template <int MT, int NB>
__device__ __forceinline__ void mfma_rows(const float* __restrict__ a_base, int row0, int a_rows, int a_pitch,
                                          const float* const (&b_rows)[NB], bool b_valid, int K, int q0, int q1,
                                          f32x4 (&acc)[NB][MT]) {
    constexpr int CH = (MT * NB <= 2) ? 8 : ((MT + NB <= 6) ? 5 : 3);
    const int lane = threadIdx.x & 63, r = lane & 15, j = lane >> 4;
    const float* a_row[MT];
    bool a_ok[MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int row = row0 + m * 16 + r;
        a_ok[m] = row < a_rows;
        a_row[m] = a_base + (size_t)min(row, a_rows - 1) * a_pitch;
    }
    for (int qc = q0; qc < q1; qc += CH) {
        f32x4 av[CH][MT], bv[CH][NB];
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const int kc = min(16 * (qc + u) + 4 * j, K - 4);
#pragma unroll
            for (int n = 0; n < NB; ++n) bv[u][n] = *reinterpret_cast<const f32x4*>(b_rows[n] + kc);
#pragma unroll
            for (int m = 0; m < MT; ++m) av[u][m] = *reinterpret_cast<const f32x4*>(a_row[m] + kc);
        }
#pragma unroll
        for (int u = 0; u < CH; ++u) {
            const bool in = (qc + u < q1) && (16 * (qc + u) + 4 * j < K);   // K % 4 == 0: a vector is inside or outside
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const f32x4 b = (in && b_valid) ? bv[u][n] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const f32x4 a = (in && a_ok[m]) ? av[u][m] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[i], acc[n][m], 0, 0, 0);
                }
            }
        }
    }
}

// One time step of both directions.  grid (ceil(H/16), dirs, row chunks), 256 threads.  Each wave takes a quarter of the
// contraction for all four gates (so the whole operand fetch is one round of loads); the partial gate tiles meet in LDS.
//   xproj [dirs][T][B][4H] (biases included)   w_hh [dirs][4H][H]   lens [B]
//   out, c [dirs][T][B][H]   gates [dirs][T][B][4H] (activated i,f,g,o; kept for the backward)
template <int MT>
__global__ __launch_bounds__(256) void lstm_step_fwd(const float* __restrict__ xproj, const float* __restrict__ w_hh,
                                                     const int* __restrict__ lens, float* __restrict__ out, float* __restrict__ c,
                                                     float* __restrict__ gates, int T, int B, int H, int s) {
    __shared__ float ex[4][4][MT * 16][17];                  // [wave][gate][batch row][unit]
    const int d = blockIdx.y, u0 = blockIdx.x * 16, b0 = blockIdx.z * (MT * 16);
    const int t = d == 0 ? s : T - 1 - s;
    const int tp = d == 0 ? t - 1 : t + 1;                 // slot holding the previous state
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, j = lane >> 4;
    const size_t dTB = (size_t)d * T * B;
    const bool has_prev = tp >= 0 && tp < T;
    const int unit = u0 + r;
    // operands of the cell update, fetched BEFORE the matrix phase so that their latency hides behind it
    float xpv[MT][4], cpv[MT];
    bool act[MT], mine[MT];
#pragma unroll
    for (int k = 0; k < MT; ++k) {
        const int p = threadIdx.x + 256 * k, b = b0 + (p >> 4), u = u0 + (p & 15);
        mine[k] = b < B && u < H;
        act[k] = mine[k] && t < lens[min(b, B - 1)];
        cpv[k] = 0.f;
#pragma unroll
        for (int n = 0; n < 4; ++n) xpv[k][n] = 0.f;
        if (act[k]) {
            const size_t row = dTB + (size_t)t * B + b;
#pragma unroll
            for (int n = 0; n < 4; ++n) xpv[k][n] = xproj[row * 4 * H + (size_t)n * H + u];
            if (has_prev) cpv[k] = c[(dTB + (size_t)tp * B + b) * H + u];
        }
    }
    f32x4 acc[4][MT];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (has_prev) {
        const float* w0 = w_hh + ((size_t)d * 4 * H + min(unit, H - 1)) * H;
        const float* const w_rows[4] = {w0, w0 + (size_t)H * H, w0 + (size_t)2 * H * H, w0 + (size_t)3 * H * H};
        const int nq = (H + 15) / 16, per = (nq + 3) / 4;
        mfma_rows<MT, 4>(out + (dTB + (size_t)tp * B) * H, b0, B, H, w_rows, unit < H, H, wave * per, min(nq, (wave + 1) * per), acc);
    }
    // accumulator layout: column = lane & 15 (unit), rows 4*j + i (batch)
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) ex[wave][n][m * 16 + 4 * j + i][r] = acc[n][m][i];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MT; ++k) {
        if (!mine[k]) continue;
        const int p = threadIdx.x + 256 * k, bl = p >> 4, ul = p & 15, b = b0 + bl, u = u0 + ul;
        const size_t row = dTB + (size_t)t * B + b;
        float hv = 0.f, cv = 0.f, gi = 0.f, gf = 0.f, gg = 0.f, go = 0.f;
        if (act[k]) {
            float pre[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) pre[n] = xpv[k][n] + ((ex[0][n][bl][ul] + ex[1][n][bl][ul]) + (ex[2][n][bl][ul] + ex[3][n][bl][ul]));
            gi = sigmoidf_(pre[0]);
            gf = sigmoidf_(pre[1]);
            gg = tanhf(pre[2]);
            go = sigmoidf_(pre[3]);
            cv = gf * cpv[k] + gi * gg;
            hv = go * tanhf(cv);
        }
        out[row * H + u] = hv;
        c[row * H + u] = cv;
        float* gs = gates + row * 4 * H + u;
        gs[0] = gi; gs[H] = gf; gs[2 * H] = gg; gs[3 * H] = go;
    }
}

// One backward time step of both directions (time runs against the forward order).
//   d_out, d_c [dirs][T][B][H] or NULL: gradients arriving at h_t / c_t from outside the recurrence
//   w_hh_t [dirs][H][4H] (transposed)   dgates [dirs][T][B][4H] (output; the step processed before is read back)
//   dc_carry [dirs][B][H] scratch, zero before the first step
template <int MT>
__global__ __launch_bounds__(256) void lstm_step_bwd(const float* __restrict__ d_out, const float* __restrict__ d_c,
                                                     const float* __restrict__ w_hh_t,
                                                     const int* __restrict__ lens, const float* __restrict__ c,
                                                     const float* __restrict__ gates, float* __restrict__ dgates,
                                                     float* __restrict__ dc_carry, int T, int B, int H, int s) {
    __shared__ float ex[4][MT * 16][17];
    const int d = blockIdx.y, u0 = blockIdx.x * 16, b0 = blockIdx.z * (MT * 16);
    const int t = d == 0 ? T - 1 - s : s;                   // forward direction walks back from the end
    const int tn = d == 0 ? t + 1 : t - 1;                  // the step processed just before (later in forward order)
    const int tp = d == 0 ? t - 1 : t + 1;                  // forward-order predecessor (its c enters f's gradient)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, j = lane >> 4;
    const size_t dTB = (size_t)d * T * B;
    const int unit = u0 + r;
    // operands of the cell gradient, fetched BEFORE the matrix phase so that their latency hides behind it
    float gv[MT][4], cvv[MT], cpv[MT], dhv[MT], dcv[MT];
    bool act[MT], mine[MT];
#pragma unroll
    for (int k = 0; k < MT; ++k) {
        const int p = threadIdx.x + 256 * k, b = b0 + (p >> 4), u = u0 + (p & 15);
        mine[k] = b < B && u < H;
        const int len = lens[min(b, B - 1)];
        act[k] = mine[k] && t < len;
        cvv[k] = cpv[k] = dhv[k] = dcv[k] = 0.f;
#pragma unroll
        for (int n = 0; n < 4; ++n) gv[k][n] = 0.f;
        if (act[k]) {
            const size_t row = dTB + (size_t)t * B + b;
#pragma unroll
            for (int n = 0; n < 4; ++n) gv[k][n] = gates[row * 4 * H + (size_t)n * H + u];
            cvv[k] = c[row * H + u];
            const bool has_prev = d == 0 ? tp >= 0 : tp < len;   // reverse: the state before the first step is zero
            if (has_prev) cpv[k] = c[(dTB + (size_t)tp * B + b) * H + u];
            dcv[k] = dc_carry[((size_t)d * B + b) * H + u];
            if (d_out) dhv[k] = d_out[row * H + u];
            if (d_c) dcv[k] += d_c[row * H + u];
        }
    }
    // dh_rec[b][u] = sum_n dgates[tn][b][n] * W_hh[n][u]: contraction over 4H, split over the four waves
    f32x4 acc[1][MT];
#pragma unroll
    for (int m = 0; m < MT; ++m) acc[0][m] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tn >= 0 && tn < T) {
        const int nq = (4 * H + 15) / 16, per = (nq + 3) / 4;
        const float* a = dgates + (dTB + (size_t)tn * B) * 4 * H;
        const float* const w_rows[1] = {w_hh_t + ((size_t)d * H + min(unit, H - 1)) * 4 * H};
        mfma_rows<MT, 1>(a, b0, B, 4 * H, w_rows, unit < H, 4 * H, wave * per, min(nq, (wave + 1) * per), acc);
    }
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int i = 0; i < 4; ++i) ex[wave][m * 16 + 4 * j + i][r] = acc[0][m][i];
    __syncthreads();
#pragma unroll
    for (int k = 0; k < MT; ++k) {
        if (!mine[k]) continue;
        const int p = threadIdx.x + 256 * k, bl = p >> 4, ul = p & 15, b = b0 + bl, u = u0 + ul;
        const size_t row = dTB + (size_t)t * B + b;
        float* dg = dgates + row * 4 * H + u;
        if (!act[k]) {                                       // inactive: nothing flows
            dg[0] = 0.f; dg[H] = 0.f; dg[2 * H] = 0.f; dg[3 * H] = 0.f;
            continue;
        }
        const float dh = dhv[k] + ((ex[0][bl][ul] + ex[1][bl][ul]) + (ex[2][bl][ul] + ex[3][bl][ul]));
        const float gi = gv[k][0], gf = gv[k][1], gg = gv[k][2], go = gv[k][3];
        const float th = tanhf(cvv[k]);
        const float dc = dcv[k] + dh * go * (1.f - th * th);
        dg[0] = dc * gg * gi * (1.f - gi);
        dg[H] = dc * cpv[k] * gf * (1.f - gf);
        dg[2 * H] = dc * gi * (1.f - gg * gg);
        dg[3 * H] = dh * th * go * (1.f - go);
        dc_carry[((size_t)d * B + b) * H + u] = dc * gf;
    }
}

}  // namespace

extern "C" {

int dwc_lstm_fwd(const float* xproj, const float* w_hh, const int* lens, float* out, float* c, float* gates, int T, int B, int H,
                 int dirs, void* stream) {
    if (T <= 0 || B <= 0 || H <= 0 || (H & 3) || dirs < 1 || dirs > 2) return DWC_EINVAL;
    const int mt = min(4, (B + 15) / 16);
    const dim3 grid((H + 15) / 16, dirs, (B + 16 * mt - 1) / (16 * mt));
    hipStream_t st = (hipStream_t)stream;
    for (int s = 0; s < T; ++s) {
        switch (mt) {
            case 1: hipLaunchKernelGGL(lstm_step_fwd<1>, grid, dim3(256), 0, st, xproj, w_hh, lens, out, c, gates, T, B, H, s); break;
            case 2: hipLaunchKernelGGL(lstm_step_fwd<2>, grid, dim3(256), 0, st, xproj, w_hh, lens, out, c, gates, T, B, H, s); break;
            case 3: hipLaunchKernelGGL(lstm_step_fwd<3>, grid, dim3(256), 0, st, xproj, w_hh, lens, out, c, gates, T, B, H, s); break;
            default: hipLaunchKernelGGL(lstm_step_fwd<4>, grid, dim3(256), 0, st, xproj, w_hh, lens, out, c, gates, T, B, H, s); break;
        }
    }
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

int dwc_lstm_bwd(const float* d_out, const float* d_c, const float* w_hh_t, const int* lens, const float* c, const float* gates,
                 float* dgates, float* dc_carry, int T, int B, int H, int dirs, void* stream) {
    if (T <= 0 || B <= 0 || H <= 0 || (H & 3) || dirs < 1 || dirs > 2) return DWC_EINVAL;
    const int mt = min(4, (B + 15) / 16);
    const dim3 grid((H + 15) / 16, dirs, (B + 16 * mt - 1) / (16 * mt));
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(dc_carry, 0, (size_t)dirs * B * H * sizeof(float), st) != hipSuccess) return DWC_ELAUNCH;
    for (int s = 0; s < T; ++s) {
        switch (mt) {
            case 1: hipLaunchKernelGGL(lstm_step_bwd<1>, grid, dim3(256), 0, st, d_out, d_c, w_hh_t, lens, c, gates, dgates, dc_carry, T, B, H, s); break;
            case 2: hipLaunchKernelGGL(lstm_step_bwd<2>, grid, dim3(256), 0, st, d_out, d_c, w_hh_t, lens, c, gates, dgates, dc_carry, T, B, H, s); break;
            case 3: hipLaunchKernelGGL(lstm_step_bwd<3>, grid, dim3(256), 0, st, d_out, d_c, w_hh_t, lens, c, gates, dgates, dc_carry, T, B, H, s); break;
            default: hipLaunchKernelGGL(lstm_step_bwd<4>, grid, dim3(256), 0, st, d_out, d_c, w_hh_t, lens, c, gates, dgates, dc_carry, T, B, H, s); break;
        }
    }
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

}  // extern "C"
