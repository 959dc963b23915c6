// Shared helpers for the gfx950 kernels of libdwcgan_hip.so (wave64, CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/dwcgan_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 dwc_bf16;
typedef __bf16 dwc_bf16x4 __attribute__((ext_vector_type(4)));
typedef __bf16 dwc_bf16x8 __attribute__((ext_vector_type(8)));

// Element-type generic group-of-4 accessors (index in units of 4 elements): fp32 tensors move 16 bytes, bf16 tensors
// 8 bytes per access; all arithmetic stays fp32 and a value is rounded to bf16 once, at its store.
#ifdef __HIPCC__
__device__ __forceinline__ f32x4 ld4(const float* p, size_t i4) { return reinterpret_cast<const f32x4*>(p)[i4]; }
__device__ __forceinline__ f32x4 ld4(const dwc_bf16* p, size_t i4) {
    const dwc_bf16x4 v = reinterpret_cast<const dwc_bf16x4*>(p)[i4];
    return f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
}
__device__ __forceinline__ void st4(float* p, size_t i4, f32x4 v) { reinterpret_cast<f32x4*>(p)[i4] = v; }
__device__ __forceinline__ void st4(dwc_bf16* p, size_t i4, f32x4 v) {
    dwc_bf16x4 r;
    r[0] = (dwc_bf16)v[0]; r[1] = (dwc_bf16)v[1]; r[2] = (dwc_bf16)v[2]; r[3] = (dwc_bf16)v[3];
    reinterpret_cast<dwc_bf16x4*>(p)[i4] = r;
}
#endif

#ifdef __HIPCC__
// Elements per thread and access: V = 16 bytes worth (4 fp32 / 8 bf16) -- 8-byte accesses reach only 0.54-0.70 of the 16-byte
// rate on MI355X (MI355X_MICROARCH.md), and r02's bf16 norm kernels (4 bf16 = 8 bytes per lane) sat at 3.8-4.3 TB/s.
template <typename T> struct VecOf { static constexpr int V = 4; };
template <> struct VecOf<dwc_bf16> { static constexpr int V = 8; };

__device__ __forceinline__ void ldv(const float* p, size_t iv, float (&o)[4]) {
    const f32x4 v = reinterpret_cast<const f32x4*>(p)[iv];
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2]; o[3] = v[3];
}
__device__ __forceinline__ void ldv(const dwc_bf16* p, size_t iv, float (&o)[8]) {
    const dwc_bf16x8 v = reinterpret_cast<const dwc_bf16x8*>(p)[iv];
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (float)v[k];
}
__device__ __forceinline__ void stv(float* p, size_t iv, const float (&o)[4]) {
    reinterpret_cast<f32x4*>(p)[iv] = f32x4{o[0], o[1], o[2], o[3]};
}
__device__ __forceinline__ void stv(dwc_bf16* p, size_t iv, const float (&o)[8]) {
    dwc_bf16x8 r;
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = (dwc_bf16)o[k];
    reinterpret_cast<dwc_bf16x8*>(p)[iv] = r;
}
// V consecutive fp32 values (per-(n,c) statistics / affine parameters) starting at element e (a multiple of V)
template <int V>
__device__ __forceinline__ void ldf(const float* p, size_t e, float (&o)[V]) {
#pragma unroll
    for (int q = 0; q < V / 4; ++q) {
        const f32x4 v = reinterpret_cast<const f32x4*>(p + e)[q];
        o[4 * q] = v[0]; o[4 * q + 1] = v[1]; o[4 * q + 2] = v[2]; o[4 * q + 3] = v[3];
    }
}
template <int V>
__device__ __forceinline__ void stf(float* p, size_t e, const float (&o)[V]) {
#pragma unroll
    for (int q = 0; q < V / 4; ++q) reinterpret_cast<f32x4*>(p + e)[q] = f32x4{o[4 * q], o[4 * q + 1], o[4 * q + 2], o[4 * q + 3]};
}

#endif

#ifdef __HIPCC__
// ---- largest-magnitude slots of the two-plane f16 split kernels (csrc/conv_halo_x3.hip, "h2") ------------------------------------
// A slot is 64 bits of caller memory: (epoch << 32) | bits of max |a| over a tensor.  Producers raise it with an atomic max -- a
// newer epoch beats any older content, so a slot is never zeroed -- and consumers derive the tensor's power-of-two working scale
// from it (h2_scale).  NaN bits compare above inf bits above every finite magnitude: a non-finite tensor reads as such.
__device__ __forceinline__ void dwc_amax_publish(unsigned long long* slot, unsigned epoch, unsigned abs_bits) {
    atomicMax(slot, ((unsigned long long)epoch << 32) | abs_bits);
}
__device__ __forceinline__ unsigned dwc_abs_bits(float v) { return __float_as_uint(v) & 0x7fffffffu; }
__device__ __forceinline__ unsigned dwc_wave_max_u32(unsigned v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, off, 64));
    return v;
}
// One lane per wave raises the slot, and only if the slot does not seem to hold at least this value already -- a PLAIN (L1-cached)
// load: a stale answer only costs an atomic that changes nothing, while thousands of device-scope loads of ONE address per launch
// queue up at a single L2 channel just like the atomics would (measured: +15 us on a 32 us norm kernel).  Called by ALL lanes of a wave.
__device__ __forceinline__ void dwc_amax_wave_publish(unsigned long long* slot, unsigned epoch, unsigned abs_bits) {
    if (!slot) return;
    abs_bits = dwc_wave_max_u32(abs_bits);
    if ((threadIdx.x & 63) == 0) {
        const unsigned long long v = ((unsigned long long)epoch << 32) | abs_bits;
        if (*reinterpret_cast<volatile unsigned long long*>(slot) < v) atomicMax(slot, v);
    }
}
// The same with the waves of a workgroup folded through LDS first (`sm`: one word per wave): one candidate per workgroup.  Called by
// ALL threads of the workgroup (it contains a barrier).
__device__ __forceinline__ void dwc_amax_block_publish(unsigned long long* slot, unsigned epoch, unsigned abs_bits, unsigned* sm) {
    if (!slot) return;                                      // (workgroup-uniform)
    abs_bits = dwc_wave_max_u32(abs_bits);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = abs_bits;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned m = 0;
        for (unsigned w = 0; w < (blockDim.x + 63) / 64; ++w) m = max(m, sm[w]);
        const unsigned long long v = ((unsigned long long)epoch << 32) | m;
        if (*reinterpret_cast<volatile unsigned long long*>(slot) < v) atomicMax(slot, v);
    }
}
template <int V>
__device__ __forceinline__ unsigned dwc_amax_fold(unsigned am, const float (&o)[V]) {
#pragma unroll
    for (int k = 0; k < V; ++k) am = max(am, dwc_abs_bits(o[k]));
    return am;
}
struct H2Scale { float s, inv; };
__device__ __forceinline__ H2Scale h2_scale(const unsigned long long* slot, unsigned epoch) {
    const unsigned long long v = *slot;
    unsigned e = ((unsigned)v >> 23) & 0xffu;
    H2Scale r;
    if (e == 0xffu) {
        r.s = r.inv = 1.f;
    } else {
        e = min(max(e, 27u), 254u);
        r.s = __uint_as_float((267u - e) << 23);      // 2^(140 - e): max|a| in [2^(e-127), 2^(e-126)) -> [2^13, 2^14)
        r.inv = __uint_as_float((e - 13u) << 23);
    }
    if ((unsigned)(v >> 32) != epoch) r.s = r.inv = __builtin_nanf("");
    return r;
}
#endif

#define DWC_LAUNCH_CHECK()                                   \
    do {                                                     \
        if (hipGetLastError() != hipSuccess) return DWC_ELAUNCH; \
    } while (0)

static inline int dwc_ilog2_exact(int v) {  // -1 if v is not a power of two
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}

__device__ __forceinline__ float dwc_wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// sum over the 256 threads of a block; result valid in every thread. `sm` >= 4 floats.
__device__ __forceinline__ float dwc_block_sum_256(float v, float* sm) {
    v = dwc_wave_sum(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sm[wave] = v;
    __syncthreads();
    return sm[0] + sm[1] + sm[2] + sm[3];
}

__device__ __forceinline__ float dwc_act_apply(float v, int act, int ch) {
    switch (act) {
        case DWC_ACT_RELU: return v < 0.f ? 0.f : v;            // (not fmaxf: a NaN must stay a NaN, as in torch)
        case DWC_ACT_LRELU: return v > 0.f ? v : 0.1f * v;
        case DWC_ACT_TANH: return tanhf(v);
        case DWC_ACT_SIGMOID: return 1.f / (1.f + expf(-v));
        case DWC_ACT_HEADS: return ((ch & 3) == 3) ? 1.f / (1.f + expf(-v)) : tanhf(v);
        case DWC_ACT_HEADS8: return ((ch & 7) == 3) ? 1.f / (1.f + expf(-v)) : ((ch & 7) < 3 ? tanhf(v) : 0.f);
        default: return v;
    }
}

// The activations without a transcendental (none / ReLU / LeakyReLU 0.1) in one branch-free form: v < 0 ? slope*v : v with
// slope 1 / 0 / 0.1 (a compare, a multiply, a select; the slope == 0 select is wave-uniform).  Non-finite values behave as in
// torch: a NaN stays a NaN (NaN < 0 is false -> v), ReLU(-inf) = 0 (not 0 * -inf), LReLU(-inf) = -inf -- the max/min form
// this replaces dropped NaNs (fmaxf/fminf return the other operand).  Epilogues take this form in one loop nest and
// the general dwc_act_apply in a SEPARATE one: with the switch inside the fully unrolled per-element code every GEMM
// kernel carried ~25 000 instructions of inlined tanhf/expf between its hot instructions (r02: a halo kernel with an EMPTY
// main loop still took half the full kernel's time, instruction fetch of that epilogue).
__device__ __forceinline__ bool dwc_act_is_simple(int act) { return act <= DWC_ACT_LRELU; }
static inline bool dwc_act_is_simple_host(int act) { return act <= DWC_ACT_LRELU; }
__device__ __forceinline__ float dwc_act_slope(int act) { return act == DWC_ACT_NONE ? 1.f : (act == DWC_ACT_RELU ? 0.f : 0.1f); }
__device__ __forceinline__ float dwc_act_simple(float v, float slope) { return v < 0.f ? (slope == 0.f ? 0.f : slope * v) : v; }

// derivative expressed through the activation OUTPUT y
__device__ __forceinline__ float dwc_act_grad(float y, int act, int ch) {
    switch (act) {
        case DWC_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case DWC_ACT_LRELU: return y > 0.f ? 1.f : 0.1f;
        case DWC_ACT_TANH: return 1.f - y * y;
        case DWC_ACT_SIGMOID: return y * (1.f - y);
        case DWC_ACT_HEADS: return ((ch & 3) == 3) ? y * (1.f - y) : 1.f - y * y;
        case DWC_ACT_HEADS8: return ((ch & 7) == 3) ? y * (1.f - y) : ((ch & 7) < 3 ? 1.f - y * y : 0.f);
        default: return 1.f;
    }
}

