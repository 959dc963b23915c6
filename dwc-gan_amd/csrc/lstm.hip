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

// ------------------------------------------------------------------------------------------
// Persistent forward: ALL time steps of both directions in ONE launch ("a persistent bi-LSTM
// kernel").  Same decomposition as lstm_step_fwd -- workgroup (g, d, z) owns 16 hidden units of direction d for batch chunk z,
// each wave a quarter of the contraction -- but the workgroup stays resident: its W_hh slice (4 gates x 16 units x H) sits in
// REGISTERS for the whole sequence (80 fp32 per lane at H = 300; the step kernels re-fetched 1.4 MB per direction from L2
// every step), its cell state c stays in registers, and only h_t crosses workgroups.  Hand-off per step (the recipe of
// cdna_hip_programming.md Guideline 16 / MI355X_MICROARCH.md "Valid forms", counter row): every h_t element is stored
// WRITE-THROUGH (sc1), every storing wave drains vmcnt, the workgroup's barrier, then ONE lane adds 1 to the counter of its
// (direction, batch chunk) group with an agent-scope atomic; a consumer's first wave polls that counter with relaxed sc1
// loads (s_sleep between polls, bounded: a missed rendez-vous sets the timeout word and ends the kernel instead of hanging),
// the workgroup barrier releases the other waves, and EVERY load of h_{t-1} is an sc1 load (no L1 copy can be stale, no
// acquire needed).  The groups are independent (19 workgroups each at H = 300); the grid (<= 76 workgroups at B <= 128) is
// far below the 256 CUs and each workgroup declares > 80 KB of LDS, so all of them are resident, one per CU.
//   ws: [0] timeout word, [16 + d * chunks + z] arrival counters (zeroed by the launcher).
// ------------------------------------------------------------------------------------------
typedef unsigned __attribute__((address_space(1))) lstm_gu32;

__device__ __forceinline__ f32x4 lstm_ld_sc1(const __amdgpu_buffer_rsrc_t rsrc, unsigned byte_off) {
    // 16-byte load that bypasses this CU's L1 (aux 16 = sc1)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, byte_off, 0, 16);
    return f32x4{__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
}

template <int MT, int NQ>      // NQ: 16-deep contraction blocks per wave (ceil(ceil(H/16)/4))
__global__ __launch_bounds__(256) void lstm_seq_fwd(const float* __restrict__ xproj, const float* __restrict__ w_hh,
                                                    const int* __restrict__ lens, float* __restrict__ out, float* __restrict__ c,
                                                    float* __restrict__ gates, unsigned* __restrict__ ws, unsigned* __restrict__ status,
                                                    int T, int B, int H) {
    __shared__ float ex[4][4][MT * 16][17];                  // [wave][gate][batch row][unit]
    __shared__ int s_ok;                                     // (+ dynamic LDS requested by the launcher: > 80 KB in all, one workgroup per CU)
    const int d = blockIdx.y, u0 = blockIdx.x * 16, z = blockIdx.z, b0 = z * (MT * 16);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, j = lane >> 4;
    const int groups = gridDim.x;
    const size_t dTB = (size_t)d * T * B;
    const int unit = u0 + r;
    lstm_gu32* counter = (lstm_gu32*)(ws + 16 + d * gridDim.z + z);
    lstm_gu32* timeout = (lstm_gu32*)ws;

    // W_hh slice of this wave's contraction quarter, in registers for the whole sequence
    const int nq = (H + 15) / 16;
    const int q0 = wave * NQ;
    f32x4 wreg[NQ][4];
    {
        const float* w0 = w_hh + ((size_t)d * 4 * H + min(unit, H - 1)) * H;
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int k = 16 * (q0 + u) + 4 * j;
            const bool in = (q0 + u < nq) && k < H && unit < H;
#pragma unroll
            for (int n = 0; n < 4; ++n)
                wreg[u][n] = in ? *reinterpret_cast<const f32x4*>(w0 + (size_t)n * H * H + min(k, H - 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const __amdgpu_buffer_rsrc_t rsrc_h = __builtin_amdgcn_make_buffer_rsrc(out, 0, (unsigned)((size_t)gridDim.y * T * B * H * 4), 0x00020000);
    float cprev[MT];                                         // cell state of this thread's (batch row, unit) pairs
    int len_k[MT];
    bool mine[MT];
#pragma unroll
    for (int k = 0; k < MT; ++k) {
        const int p = threadIdx.x + 256 * k, b = b0 + (p >> 4), u = u0 + (p & 15);
        mine[k] = b < B && u < H;
        len_k[k] = lens[min(b, B - 1)];
        cprev[k] = 0.f;
    }
    bool alive = true;
    for (int s = 0; s < T; ++s) {
        const int t = d == 0 ? s : T - 1 - s;
        const int tp = d == 0 ? t - 1 : t + 1;
        // operands of the cell update that nobody hands off: fetched before the rendez-vous
        float xpv[MT][4];
        bool act[MT];
#pragma unroll
        for (int k = 0; k < MT; ++k) {
            const int p = threadIdx.x + 256 * k, b = b0 + (p >> 4), u = u0 + (p & 15);
            act[k] = mine[k] && t < len_k[k];
#pragma unroll
            for (int n = 0; n < 4; ++n) xpv[k][n] = act[k] ? xproj[(dTB + (size_t)t * B + b) * 4 * H + (size_t)n * H + u] : 0.f;
        }
        f32x4 acc[4][MT];
#pragma unroll
        for (int n = 0; n < 4; ++n)
#pragma unroll
            for (int m = 0; m < MT; ++m) acc[n][m] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
            // rendez-vous: every workgroup of this (direction, batch chunk) group has published h of step s-1
            if (threadIdx.x < 64) {
                bool ok = true;
                if (threadIdx.x == 0) {
                    const unsigned want = (unsigned)groups * (unsigned)s;
                    unsigned spins = 0;
                    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > (1u << 20)) {                        // ~ a second: give up instead of hanging the GPU
                            __hip_atomic_store(timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            if (status) __hip_atomic_fetch_or((lstm_gu32*)status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                            ok = false;
                            break;
                        }
                    }
                    s_ok = ok ? 1 : 0;
                }
            }
            __syncthreads();
            alive = s_ok != 0;
            if (!alive) break;                                             // uniform: s_ok is workgroup-wide
            // h_{t-1}: [B][H] rows of this batch chunk, EVERY load sc1
            const unsigned row_bytes = (unsigned)H * 4u;
            const unsigned base = (unsigned)((dTB + (size_t)tp * B) * H * 4);
#pragma unroll
            for (int u = 0; u < NQ; ++u) {
                const int k = 16 * (q0 + u) + 4 * j;
                const bool kin = (q0 + u < nq) && k < H;
                f32x4 av[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const int row = b0 + m * 16 + r;
                    av[m] = lstm_ld_sc1(rsrc_h, base + (unsigned)min(row, B - 1) * row_bytes + (unsigned)min(k, H - 4) * 4u);
                    if (!(kin && row < B)) av[m] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int i = 0; i < 4; ++i)
                            acc[n][m] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][i], wreg[u][n][i], acc[n][m], 0, 0, 0);
            }
        }
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
                cv = gf * cprev[k] + gi * gg;
                hv = go * tanhf(cv);
            }
            cprev[k] = cv;
            // h crosses workgroups: write-through store (sc1).  c and the gates are only read after the launch.
            __hip_atomic_store((lstm_gu32*)(out + row * H + u), __float_as_uint(hv), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            c[row * H + u] = cv;
            float* gs = gates + row * 4 * H + u;
            gs[0] = gi; gs[H] = gf; gs[2 * H] = gg; gs[3 * H] = go;
        }
        // publish: every storing wave drains its stores, the workgroup meets, ONE lane signals
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0 && s + 1 < T) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!alive) {
        // a rendez-vous was missed (a workgroup of the group never became resident): the results are INVALID.  Fail loudly: every
        // h / c / gate slot this workgroup owns becomes NaN (losses downstream turn NaN) and `status` keeps the sticky flag.
        const float nan = __uint_as_float(0x7fc00000u);
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int k = 0; k < MT; ++k) {
                if (!mine[k]) continue;
                const int p = threadIdx.x + 256 * k, b = b0 + (p >> 4), u = u0 + (p & 15);
                const size_t row = dTB + (size_t)t * B + b;
                out[row * H + u] = nan;
                c[row * H + u] = nan;
                float* gs = gates + row * 4 * H + u;
                gs[0] = nan; gs[H] = nan; gs[2 * H] = nan; gs[3 * H] = nan;
            }
    }
}

// Persistent backward: all time steps of both directions in one launch, same hand-off as lstm_seq_fwd.  What crosses
// workgroups is dgates of the step processed before (each workgroup writes the four gate slices of its 16 units; every
// workgroup of the group contracts over all 4H of them); the transposed W_hh slice of the workgroup's units (4H values per
// unit, NQ 16-deep blocks per wave = 76 fp32 per lane at H = 300) and the cell-gradient carry stay in registers.
template <int MT, int NQ>
__global__ __launch_bounds__(256) void lstm_seq_bwd(const float* __restrict__ d_out, const float* __restrict__ d_c,
                                                    const float* __restrict__ w_hh_t, const int* __restrict__ lens,
                                                    const float* __restrict__ c, const float* __restrict__ gates, float* __restrict__ dgates,
                                                    unsigned* __restrict__ ws, unsigned* __restrict__ status, int T, int B, int H) {
    __shared__ float ex[4][MT * 16][17];
    __shared__ int s_ok;
    const int d = blockIdx.y, u0 = blockIdx.x * 16, z = blockIdx.z, b0 = z * (MT * 16);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, j = lane >> 4;
    const int groups = gridDim.x;
    const size_t dTB = (size_t)d * T * B;
    const int unit = u0 + r;
    const int K = 4 * H;
    lstm_gu32* counter = (lstm_gu32*)(ws + 16 + d * gridDim.z + z);
    lstm_gu32* timeout = (lstm_gu32*)ws;
    const int nq = (K + 15) / 16;
    const int q0 = wave * NQ;
    f32x4 wreg[NQ];
    {
        const float* w0 = w_hh_t + ((size_t)d * H + min(unit, H - 1)) * K;
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int k = 16 * (q0 + u) + 4 * j;
            const bool in = (q0 + u < nq) && k < K && unit < H;
            wreg[u] = in ? *reinterpret_cast<const f32x4*>(w0 + min(k, K - 4)) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
    }
    const __amdgpu_buffer_rsrc_t rsrc_g = __builtin_amdgcn_make_buffer_rsrc(dgates, 0, (unsigned)((size_t)gridDim.y * T * B * K * 4), 0x00020000);
    float carry[MT];
    int len_k[MT];
    bool mine[MT];
#pragma unroll
    for (int k = 0; k < MT; ++k) {
        const int p = threadIdx.x + 256 * k, b = b0 + (p >> 4), u = u0 + (p & 15);
        mine[k] = b < B && u < H;
        len_k[k] = lens[min(b, B - 1)];
        carry[k] = 0.f;
    }
    bool alive = true;
    for (int s = 0; s < T; ++s) {
        const int t = d == 0 ? T - 1 - s : s;                   // forward direction walks back from the end
        const int tn = d == 0 ? t + 1 : t - 1;                  // the step processed just before
        const int tp = d == 0 ? t - 1 : t + 1;                  // forward-order predecessor (its c enters f's gradient)
        float gv[MT][4], cvv[MT], cpv[MT], dhv[MT], dcv[MT];
        bool act[MT];
#pragma unroll
        for (int k = 0; k < MT; ++k) {
            const int p = threadIdx.x + 256 * k, b = b0 + (p >> 4), u = u0 + (p & 15);
            act[k] = mine[k] && t < len_k[k];
            cvv[k] = cpv[k] = dhv[k] = 0.f;
            dcv[k] = carry[k];
#pragma unroll
            for (int n = 0; n < 4; ++n) gv[k][n] = 0.f;
            if (act[k]) {
                const size_t row = dTB + (size_t)t * B + b;
#pragma unroll
                for (int n = 0; n < 4; ++n) gv[k][n] = gates[row * K + (size_t)n * H + u];
                cvv[k] = c[row * H + u];
                const bool has_prev = d == 0 ? tp >= 0 : tp < len_k[k];   // reverse: the state before the first step is zero
                if (has_prev) cpv[k] = c[(dTB + (size_t)tp * B + b) * H + u];
                if (d_out) dhv[k] = d_out[row * H + u];
                if (d_c) dcv[k] += d_c[row * H + u];
            }
        }
        f32x4 acc[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc[m] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (s > 0) {
            if (threadIdx.x == 0) {
                bool ok = true;
                const unsigned want = (unsigned)groups * (unsigned)s;
                unsigned spins = 0;
                while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                    __builtin_amdgcn_s_sleep(2);
                    if (++spins > (1u << 20)) {
                        __hip_atomic_store(timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (status) __hip_atomic_fetch_or((lstm_gu32*)status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        ok = false;
                        break;
                    }
                }
                s_ok = ok ? 1 : 0;
            }
            __syncthreads();
            if (s_ok == 0) { alive = false; break; }
            const unsigned row_bytes = (unsigned)K * 4u;
            const unsigned base = (unsigned)((dTB + (size_t)tn * B) * K * 4);
            constexpr int CH = 5;
#pragma unroll
            for (int uc = 0; uc < NQ; uc += CH) {
                f32x4 av[CH][MT];
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    if (uc + u >= NQ) continue;
                    const int k = 16 * (q0 + uc + u) + 4 * j;
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const int row = b0 + m * 16 + r;
                        av[u][m] = lstm_ld_sc1(rsrc_g, base + (unsigned)min(row, B - 1) * row_bytes + (unsigned)min(k, K - 4) * 4u);
                    }
                }
#pragma unroll
                for (int u = 0; u < CH; ++u) {
                    if (uc + u >= NQ) continue;
                    const int k = 16 * (q0 + uc + u) + 4 * j;
                    const bool kin = (q0 + uc + u < nq) && k < K;
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        const bool ok = kin && (b0 + m * 16 + r) < B;
                        const f32x4 a = ok ? av[u][m] : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int i = 0; i < 4; ++i) acc[m] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], wreg[uc + u][i], acc[m], 0, 0, 0);
                    }
                }
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) ex[wave][m * 16 + 4 * j + i][r] = acc[m][i];
        __syncthreads();
#pragma unroll
        for (int k = 0; k < MT; ++k) {
            if (!mine[k]) continue;
            const int p = threadIdx.x + 256 * k, bl = p >> 4, ul = p & 15, b = b0 + bl, u = u0 + ul;
            const size_t row = dTB + (size_t)t * B + b;
            lstm_gu32* dg = (lstm_gu32*)(dgates + row * K + u);
            float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
            if (act[k]) {
                const float dh = dhv[k] + ((ex[0][bl][ul] + ex[1][bl][ul]) + (ex[2][bl][ul] + ex[3][bl][ul]));
                const float gi = gv[k][0], gf = gv[k][1], gg = gv[k][2], go = gv[k][3];
                const float th = tanhf(cvv[k]);
                const float dc = dcv[k] + dh * go * (1.f - th * th);
                o0 = dc * gg * gi * (1.f - gi);
                o1 = dc * cpv[k] * gf * (1.f - gf);
                o2 = dc * gi * (1.f - gg * gg);
                o3 = dh * th * go * (1.f - go);
                carry[k] = dc * gf;
            }
            // (inactive: nothing flows, zeros are stored and the carry is kept -- it is only read again while active)
            __hip_atomic_store(dg, __float_as_uint(o0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dg + H, __float_as_uint(o1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dg + 2 * H, __float_as_uint(o2), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dg + 3 * H, __float_as_uint(o3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0 && s + 1 < T) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!alive) {                                                // missed rendez-vous: poison every gate gradient this workgroup owns
        const float nan = __uint_as_float(0x7fc00000u);
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int k = 0; k < MT; ++k) {
                if (!mine[k]) continue;
                const int p = threadIdx.x + 256 * k, b = b0 + (p >> 4), u = u0 + (p & 15);
                float* dg = dgates + (dTB + (size_t)t * B + b) * K + u;
                dg[0] = nan; dg[H] = nan; dg[2 * H] = nan; dg[3 * H] = nan;
            }
    }
}

// Workgroups that can be resident at once for a persistent kernel (CUs x occupancy at its LDS request), per device and
// kernel variant.  The hand-off inside lstm_seq_* spins on values other workgroups of the launch produce: every workgroup
// of the grid MUST be resident, so the launchers refuse (DWC_EINVAL -> the caller runs the per-step kernels) any grid above
// this capacity -- a partitioned device (CPX), a CU mask or a smaller part lowers it.
template <typename K>
int lstm_resident_capacity(K kernel, unsigned dyn_lds, int variant) {
    static int cache[16][16];                                   // [device][variant]; racing writers store the same value
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 0;
    int* slot = (dev >= 0 && dev < 16 && variant >= 0 && variant < 16) ? &cache[dev][variant] : nullptr;
    if (slot && *slot > 0) return *slot;
    int cus = 0, occ = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kernel, 256, dyn_lds) != hipSuccess) return 0;
    const int cap = cus * occ;
    if (slot) *slot = cap;
    return cap;
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

/* All T steps of both directions in one launch (lstm_seq_fwd); same tensors as dwc_lstm_fwd plus ws >= dwc_lstm_seq_ws_bytes
 * (arrival counters + a timeout word: after the launch ws[0] != 0 means a rendez-vous was missed and the results are invalid).
 * Returns DWC_EINVAL for shapes the persistent form does not take (H > 320, more workgroups than CUs): use dwc_lstm_fwd. */
size_t dwc_lstm_seq_ws_bytes(int B, int dirs) { return ((size_t)(16 + dirs * ((B + 63) / 16 + 1)) * sizeof(unsigned) + 64 + 15) / 16 * 16; }

int dwc_lstm_seq_fwd(const float* xproj, const float* w_hh, const int* lens, float* out, float* c, float* gates, int T, int B, int H,
                     int dirs, void* ws, size_t ws_bytes, unsigned* status, int max_workgroups, void* stream) {
    if (T <= 0 || B <= 0 || H <= 0 || (H & 3) || dirs < 1 || dirs > 2) return DWC_EINVAL;
    const int nq = (H + 15) / 16, per = (nq + 3) / 4;
    if (per > 5 || (size_t)dirs * T * B * H * 4 >= 0x80000000ull) return DWC_EINVAL;
    // Batch rows per workgroup (16 * mt): the SMALLEST chunk whose grid is still resident at once -- a step is a chain of MFMAs over
    // the chunk's rows (320 fp32 MFMAs per wave at 64 rows: 5 of the 11 us of a step at B = 128), so more, smaller chunks on the
    // idle CUs shorten every step (r04: B = 128 runs 152 workgroups of 32 rows instead of 76 of 64).
    int mt = 0;
    unsigned dyn = 0;
    dim3 grid;
    for (int m = 1; m <= 4 && !mt; ++m) {
        // dynamic LDS on top of the exchange buffer so that a workgroup needs > 80 KB: ONE workgroup per CU (the hand-off form used
        // is measured for one workgroup per CU, and the residency argument counts CUs)
        const size_t ex_bytes = (size_t)4 * 4 * m * 16 * 17 * 4;
        const unsigned dy = ex_bytes < 84 * 1024 ? (unsigned)(84 * 1024 - ex_bytes) : 0u;
        int cp = 0;
        switch (m) {
            case 1: cp = lstm_resident_capacity(lstm_seq_fwd<1, 5>, dy, 0); break;
            case 2: cp = lstm_resident_capacity(lstm_seq_fwd<2, 5>, dy, 1); break;
            case 3: cp = lstm_resident_capacity(lstm_seq_fwd<3, 5>, dy, 2); break;
            default: cp = lstm_resident_capacity(lstm_seq_fwd<4, 5>, dy, 3); break;
        }
        if (max_workgroups > 0) cp = min(cp, max_workgroups);
        const dim3 g((H + 15) / 16, dirs, (B + 16 * m - 1) / (16 * m));
        if ((size_t)g.x * g.y * g.z <= (size_t)max(cp, 0)) mt = m, dyn = dy, grid = g;
    }
    if (!mt) return DWC_EINVAL;                                  // not all workgroups would be resident at any chunk size
    if (!ws || ws_bytes < dwc_lstm_seq_ws_bytes(B, dirs)) return DWC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(ws, 0, dwc_lstm_seq_ws_bytes(B, dirs), st) != hipSuccess) return DWC_ELAUNCH;
    unsigned* w = (unsigned*)ws;
    switch (mt) {
        case 1: hipLaunchKernelGGL((lstm_seq_fwd<1, 5>), grid, dim3(256), dyn, st, xproj, w_hh, lens, out, c, gates, w, status, T, B, H); break;
        case 2: hipLaunchKernelGGL((lstm_seq_fwd<2, 5>), grid, dim3(256), dyn, st, xproj, w_hh, lens, out, c, gates, w, status, T, B, H); break;
        case 3: hipLaunchKernelGGL((lstm_seq_fwd<3, 5>), grid, dim3(256), dyn, st, xproj, w_hh, lens, out, c, gates, w, status, T, B, H); break;
        default: hipLaunchKernelGGL((lstm_seq_fwd<4, 5>), grid, dim3(256), dyn, st, xproj, w_hh, lens, out, c, gates, w, status, T, B, H); break;
    }
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

int dwc_lstm_seq_bwd(const float* d_out, const float* d_c, const float* w_hh_t, const int* lens, const float* c, const float* gates,
                     float* dgates, int T, int B, int H, int dirs, void* ws, size_t ws_bytes, unsigned* status, int max_workgroups,
                     void* stream) {
    if (T <= 0 || B <= 0 || H <= 0 || (H & 3) || dirs < 1 || dirs > 2) return DWC_EINVAL;
    const int nq = (4 * H + 15) / 16, per = (nq + 3) / 4;
    if (per > 19 || (size_t)dirs * T * B * 4 * H * 4 >= 0x80000000ull) return DWC_EINVAL;
    int mt = 0;                                                  // smallest resident chunk, see dwc_lstm_seq_fwd
    unsigned dyn = 0;
    dim3 grid;
    for (int m = 1; m <= 4 && !mt; ++m) {
        const size_t ex_bytes = (size_t)4 * m * 16 * 17 * 4;
        const unsigned dy = (unsigned)(84 * 1024 - ex_bytes);
        int cp = 0;
        switch (m) {
            case 1: cp = lstm_resident_capacity(lstm_seq_bwd<1, 19>, dy, 4); break;
            case 2: cp = lstm_resident_capacity(lstm_seq_bwd<2, 19>, dy, 5); break;
            case 3: cp = lstm_resident_capacity(lstm_seq_bwd<3, 19>, dy, 6); break;
            default: cp = lstm_resident_capacity(lstm_seq_bwd<4, 19>, dy, 7); break;
        }
        if (max_workgroups > 0) cp = min(cp, max_workgroups);
        const dim3 g((H + 15) / 16, dirs, (B + 16 * m - 1) / (16 * m));
        if ((size_t)g.x * g.y * g.z <= (size_t)max(cp, 0)) mt = m, dyn = dy, grid = g;
    }
    if (!mt) return DWC_EINVAL;
    if (!ws || ws_bytes < dwc_lstm_seq_ws_bytes(B, dirs)) return DWC_EWORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(ws, 0, dwc_lstm_seq_ws_bytes(B, dirs), st) != hipSuccess) return DWC_ELAUNCH;
    unsigned* w = (unsigned*)ws;
    switch (mt) {
        case 1: hipLaunchKernelGGL((lstm_seq_bwd<1, 19>), grid, dim3(256), dyn, st, d_out, d_c, w_hh_t, lens, c, gates, dgates, w, status, T, B, H); break;
        case 2: hipLaunchKernelGGL((lstm_seq_bwd<2, 19>), grid, dim3(256), dyn, st, d_out, d_c, w_hh_t, lens, c, gates, dgates, w, status, T, B, H); break;
        case 3: hipLaunchKernelGGL((lstm_seq_bwd<3, 19>), grid, dim3(256), dyn, st, d_out, d_c, w_hh_t, lens, c, gates, dgates, w, status, T, B, H); break;
        default: hipLaunchKernelGGL((lstm_seq_bwd<4, 19>), grid, dim3(256), dyn, st, d_out, d_c, w_hh_t, lens, c, gates, dgates, w, status, T, B, H); break;
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
