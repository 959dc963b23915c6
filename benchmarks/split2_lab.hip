// Numerics laboratory for the fp32-on-matrix-core split products.
//   x3 : a = a0 + a1 + a2 (three bf16 planes, exact), six v_mfma_f32_32x32x16_bf16 per k-step (csrc/conv_halo_x3.hip as shipped in r04)
//   h2 : s*a = hi + lo * 2^-11 (two f16 planes, round-to-nearest: hi = f16(s*a), lo = f16((s*a - hi) * 2^11)), THREE
//        v_mfma_f32_32x32x16_f16 per k-step: hi*hi into the main accumulator, hi*lo + lo*hi into a correction accumulator that is
//        merged once (x 2^-11) at the end; s = a per-operand power of two that puts the operand's largest magnitude at 2^13..2^14
//   f32: v_mfma_f32_32x32x2_f32 (the native instruction)
// One wave computes C[32][32] = A[32][K] . B[32][K]^T each way; the host compares with a float64 product and prints max / rms error
// relative to the output scale (max |C|), the measure tests/test_x3_parity.py uses.
// Build: hipcc --offload-arch=gfx950 -O3 -o split2_lab split2_lab.hip ; run: ./split2_lab
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split3(float v, __bf16& p0, __bf16& p1, __bf16& p2) {
    const unsigned hb = __float_as_uint(v) & 0xffff0000u;
    const float r1 = v - __uint_as_float(hb);
    const unsigned mb = __float_as_uint(r1) & 0xffff0000u;
    const float r2 = r1 - __uint_as_float(mb);
    p0 = __builtin_bit_cast(__bf16, (unsigned short)(hb >> 16));
    p1 = __builtin_bit_cast(__bf16, (unsigned short)(mb >> 16));
    p2 = __builtin_bit_cast(__bf16, (unsigned short)(__float_as_uint(r2) >> 16));
}

// mode 0: x3, 1: h2 (two accumulators), 2: fp32 MFMA, 3: h2 without scales (plain f16 range), 7 / 8: unscaled lo + one accumulator,
// 4 / 5 / 6: h2 whose MFMA accumulators
// are flushed into fp32 VALU accumulators (round to nearest) every 4 / 9 / 25 k-steps
__global__ void lab_kernel(const float* A, const float* B, float* C, int K, int mode, float sa, float sb) {
    const int lane = threadIdx.x, l31 = lane & 31, hi = lane >> 5;
    f32x16 acc, lo;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f, lo[r] = 0.f;
    if (mode == 2) {
        for (int k = 0; k < K; k += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[l31 * K + k + hi], B[l31 * K + k + hi], acc, 0, 0, 0);
    } else if (mode == 0) {
        for (int k = 0; k < K; k += 16) {
            bf16x8 a[3], b[3];
            for (int e = 0; e < 8; ++e) {
                __bf16 p0, p1, p2;
                split3(A[l31 * K + k + 8 * hi + e], p0, p1, p2);
                a[0][e] = p0; a[1][e] = p1; a[2][e] = p2;
                split3(B[l31 * K + k + 8 * hi + e], p0, p1, p2);
                b[0][e] = p0; b[1][e] = p1; b[2][e] = p2;
            }
            lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], lo, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], acc, 0, 0, 0);
        }
        for (int r = 0; r < 16; ++r) acc[r] += lo[r];
    } else if (mode >= 7) {
        // UNSCALED lo plane (lo = f16(s a - hi): full precision down to 2^-16 of the tensor's largest magnitude instead of 2^-27) -- all
        // three products carry the same weight and go into ONE MFMA accumulator, flushed every F k-steps
        const int F = mode == 7 ? 9 : 25;
        f32x16 big;
        for (int r = 0; r < 16; ++r) big[r] = 0.f;
        int cnt = 0;
        for (int k = 0; k < K; k += 16) {
            f16x8 a[2], b[2];
            for (int e = 0; e < 8; ++e) {
                const float va = A[l31 * K + k + 8 * hi + e] * sa, vb = B[l31 * K + k + 8 * hi + e] * sb;
                a[0][e] = (_Float16)va;
                a[1][e] = (_Float16)(va - (float)a[0][e]);
                b[0][e] = (_Float16)vb;
                b[1][e] = (_Float16)(vb - (float)b[0][e]);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc, 0, 0, 0);
            if (++cnt == F) {
                cnt = 0;
                for (int r = 0; r < 16; ++r) big[r] += acc[r], acc[r] = 0.f;
            }
        }
        const float inv = 1.f / (sa * sb);
        for (int r = 0; r < 16; ++r) acc[r] = (big[r] + acc[r]) * inv;
    } else if (mode >= 4) {
        const int F = mode == 4 ? 4 : (mode == 5 ? 9 : 25);
        f32x16 big;
        for (int r = 0; r < 16; ++r) big[r] = 0.f;
        int cnt = 0;
        for (int k = 0; k < K; k += 16) {
            f16x8 a[2], b[2];
            for (int e = 0; e < 8; ++e) {
                const float va = A[l31 * K + k + 8 * hi + e] * sa, vb = B[l31 * K + k + 8 * hi + e] * sb;
                a[0][e] = (_Float16)va;
                a[1][e] = (_Float16)((va - (float)a[0][e]) * 2048.f);
                b[0][e] = (_Float16)vb;
                b[1][e] = (_Float16)((vb - (float)b[0][e]) * 2048.f);
            }
            lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], lo, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc, 0, 0, 0);
            if (++cnt == F) {
                cnt = 0;
                for (int r = 0; r < 16; ++r) big[r] += acc[r] + lo[r] * (1.f / 2048.f), acc[r] = 0.f, lo[r] = 0.f;
            }
        }
        const float inv = 1.f / (sa * sb);
        for (int r = 0; r < 16; ++r) acc[r] = (big[r] + acc[r] + lo[r] * (1.f / 2048.f)) * inv;
    } else {
        for (int k = 0; k < K; k += 16) {
            f16x8 a[2], b[2];
            for (int e = 0; e < 8; ++e) {
                const float va = A[l31 * K + k + 8 * hi + e] * sa, vb = B[l31 * K + k + 8 * hi + e] * sb;
                a[0][e] = (_Float16)va;
                a[1][e] = (_Float16)((va - (float)a[0][e]) * 2048.f);
                b[0][e] = (_Float16)vb;
                b[1][e] = (_Float16)((vb - (float)b[0][e]) * 2048.f);
            }
            lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], lo, 0, 0, 0);
            lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], lo, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], acc, 0, 0, 0);
        }
        const float inv = 1.f / (sa * sb);
        for (int r = 0; r < 16; ++r) acc[r] = (acc[r] + lo[r] * (1.f / 2048.f)) * inv;
    }
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * hi) * 32 + l31] = acc[r];
}

// denormal probe: one product of two f16 numbers that are subnormal / tiny, through the MFMA
__global__ void denorm_kernel(float* out) {
    const int lane = threadIdx.x;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) a[e] = (_Float16)0.f, b[e] = (_Float16)0.f;
    if (lane == 0) { a[0] = __builtin_bit_cast(_Float16, (unsigned short)0x0001); }      // 2^-24, smallest subnormal
    if (lane == 0) { b[0] = (_Float16)16384.f; }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];      // expect 2^-24 * 2^14 = 2^-10 if subnormal inputs are honoured, 0 if flushed
    // conversion of a tiny fp32 to f16: subnormal result or flushed?
    if (lane == 0) out[1] = (float)(_Float16)(out[2]);
}

static float pow2_scale(const std::vector<float>& v) {      // s = 2^k with 2^13 <= s * max|v| < 2^14
    float m = 0.f;
    for (float x : v) m = std::fmax(m, std::fabs(x));
    if (!(m > 0.f) || !std::isfinite(m)) return 1.f;
    int e;
    std::frexp(m, &e);      // m = f * 2^e, f in [0.5, 1)
    return std::ldexp(1.f, 14 - e);
}

int main() {
    const int Ks[] = {64, 2304, 6400, 32768};
    const char* dist_names[] = {"uniform(-1,1) x uniform(-1,1)", "relu(normal) x normal*0.02", "normal*1e-6 x normal*0.02", "uniform*1e4 x normal",
                                "lognormal(sigma 4) x normal", "positive uniform(0,1) x uniform(0,1) (no cancellation)"};
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.f, 1.f);
    std::uniform_real_distribution<float> ud(-1.f, 1.f);
    float *dA, *dB, *dC, *dD;
    const int KM = 32768;
    hipMalloc(&dA, 32 * KM * 4); hipMalloc(&dB, 32 * KM * 4); hipMalloc(&dC, 32 * 32 * 4); hipMalloc(&dD, 16);
    float tiny[4] = {0.f, 0.f, 1.0e-7f, 0.f};
    hipMemcpy(dD, tiny, 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(denorm_kernel, dim3(1), dim3(64), 0, 0, dD);
    hipMemcpy(tiny, dD, 16, hipMemcpyDeviceToHost);
    printf("f16 MFMA subnormal input: 2^-24 * 2^14 -> %g (2^-10 = %g if honoured); (float)(f16)1e-7 -> %g\n", tiny[0], 1.0 / 1024, tiny[1]);
    for (int d = 0; d < 6; ++d) {
        for (int K : Ks) {
            std::vector<float> A(32 * K), B(32 * K), C(1024);
            for (int i = 0; i < 32 * K; ++i) {
                float a, b;
                switch (d) {
                    case 0: a = ud(rng); b = ud(rng); break;
                    case 1: a = std::fmax(nd(rng), 0.f); b = 0.02f * nd(rng); break;
                    case 2: a = 1e-6f * nd(rng); b = 0.02f * nd(rng); break;
                    case 3: a = 1e4f * ud(rng); b = nd(rng); break;
                    case 4: a = std::exp(4.f * nd(rng)) * (ud(rng) < 0 ? -1.f : 1.f); b = nd(rng); break;
                    default: a = 0.5f * (ud(rng) + 1.f); b = 0.5f * (ud(rng) + 1.f); break;
                }
                A[i] = a; B[i] = b;
            }
            std::vector<double> ref(1024);
            double scale = 0;
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) {
                    double s = 0;
                    for (int k = 0; k < K; ++k) s += (double)A[i * K + k] * (double)B[j * K + k];
                    ref[i * 32 + j] = s;
                    scale = std::fmax(scale, std::fabs(s));
                }
            // float32 sequential CPU sum (what a scalar fp32 loop commits)
            double cpu_max = 0;
            for (int i = 0; i < 32; ++i)
                for (int j = 0; j < 32; ++j) {
                    float s = 0;
                    for (int k = 0; k < K; ++k) s = fmaf(A[i * K + k], B[j * K + k], s);
                    cpu_max = std::fmax(cpu_max, std::fabs((double)s - ref[i * 32 + j]));
                }
            hipMemcpy(dA, A.data(), 32 * K * 4, hipMemcpyHostToDevice);
            hipMemcpy(dB, B.data(), 32 * K * 4, hipMemcpyHostToDevice);
            const float sa = pow2_scale(A), sb = pow2_scale(B);
            printf("%-52s K=%5d scale %.3e sa 2^%d sb 2^%d | fp32-fma-cpu %.2e", dist_names[d], K, scale, (int)std::log2(sa), (int)std::log2(sb), cpu_max / scale);
            const char* mn[] = {"x3", "h2", "mfma32", "h2-unscaled", "h2-flush4", "h2-flush9", "h2-flush25", "h2u1acc-flush9", "h2u1acc-flush25"};
            for (int mode = 0; mode < 9; ++mode) {
                const int m = mode == 3 ? 1 : mode;
                hipLaunchKernelGGL(lab_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, K, m, mode == 3 ? 1.f : sa, mode == 3 ? 1.f : sb);
                hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
                double mx = 0, ss = 0;
                for (int i = 0; i < 1024; ++i) {
                    const double e = std::fabs((double)C[i] - ref[i]);
                    mx = std::fmax(mx, std::isfinite(e) ? e : 1e300);
                    ss += e * e;
                }
                printf(" | %s max %.2e rms %.2e", mn[mode], mx / scale, std::sqrt(ss / 1024) / scale);
            }
            printf("\n");
        }
    }
    return 0;
}
