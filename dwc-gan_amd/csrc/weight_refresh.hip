// Refresh of EVERY prepared weight layout of a network in ONE launch, right behind the optimiser step
// ("fused multi-tensor Adam(+L2 wd)+EMA+KRSC-weight-refresh"; reference solver.py:240,353 are the
// optimiser steps whose results the conv kernels stream in re-laid-out form).
//
// The conv kernels do not read the OIHW master weights: they stream prepared copies -- [N][Kp] im2col rows (fp32 or bf16,
// forward / data-gradient / transposed-filter data-gradient, four parity classes for stride 2), three-way bf16 split planes
// (conv_halo_x3.hip), two-plane f16 split planes with their scales (r05).  Rounds 1-2 rebuilt each of them lazily with its own
// launch the first time a layer ran after the step: ~137 launches per iteration (weight_prepare_dgrad 66, _fwd 35,
// x3_weight_prepare 20, wino_filter 16).  Here a device descriptor table lists every (master weight, prepared tensor, layout)
// pair; workgroup b rebuilds DWC_OPT_CHUNK work items of descriptor chunk_desc[b] from chunk_start[b].  The element formulas
// are those of weight_prepare_*_kernel (conv_igemm.hip, conv_bf16.hip), x3_weight_prepare_kernel / h2_weight_prepare_kernel --
// tests/test_hip_parity.py::test_weight_refresh_multi_matches_single_layout_kernels holds the two bit for bit.
#include "dwc_common.h"

namespace {

typedef __bf16 bf16;

// fwd / dgrad element of the [rows][Kp] im2col layout (conv_igemm.hip: weight_prepare_fwd_kernel / _dgrad_kernel)
__device__ __forceinline__ float im2col_elem(const dwc_refresh_desc& d, size_t idx, bool dgrad) {
    const float* w = d.src;
    if (!dgrad) {
        const int KHW = d.KH * d.KW;
        const int k = idx % d.Kp, co = idx / d.Kp;
        const int ci = k % d.cin_pad, tap = k / d.cin_pad;
        return (co < d.Cout && ci < d.Cin && tap < KHW) ? w[((size_t)co * d.Cin + ci) * KHW + tap] : 0.f;
    }
    const size_t per_class = (size_t)d.cin_pad * d.Kp;
    const int cls = idx / per_class;
    const size_t r = idx % per_class;
    const int k = r % d.Kp, ci = r / d.Kp;
    const int co = k % d.cout_pad, tapo = k / d.cout_pad;
    int kh, kw;
    bool ok = co < d.Cout && ci < d.Cin;
    if (d.stride == 1) {
        ok = ok && tapo < d.KH * d.KW;
        kh = d.KH - 1 - tapo / d.KW;
        kw = d.KW - 1 - tapo % d.KW;
    } else {
        ok = ok && tapo < 4;
        kh = (cls >> 1) + 2 * (tapo >> 1);
        kw = (cls & 1) + 2 * (tapo & 1);
    }
    // transpose_hw: the layout of the filter with its two spatial axes swapped (square filters: "dgrad_t" of hipdwc.ops)
    const int off = d.transpose_hw ? kw * d.KH + kh : kh * d.KW + kw;
    return ok ? w[((size_t)co * d.Cin + ci) * d.KH * d.KW + off] : 0.f;
}

// source element of work item idx of the [tap][slab][plane][row][16] split layouts (conv_halo_x3.hip: x3 / h2 weight prepare) and
// the offset of its plane-0 slot
__device__ __forceinline__ float split_layout_elem(const dwc_refresh_desc& d, size_t idx, bool dgrad, int planes, size_t* base) {
    const int K = d.KH, rows = d.rows, kdim = d.kdim, CS = 16;
    const int ncs = (kdim + CS - 1) / CS;
    const int j = idx % CS;
    size_t r = idx / CS;
    const int row = r % rows;
    r /= rows;
    const int cs = r % ncs, tap = r / ncs;
    const int kh = tap / K, kw = tap - kh * K;
    const int kc = cs * CS + j;
    float v = 0.f;
    if (!dgrad) {
        if (row < d.Cout && kc < d.Cin) v = d.src[(((size_t)row * d.Cin + kc) * K + kh) * K + kw];
    } else {
        if (row < d.Cin && kc < d.Cout) v = d.src[(((size_t)kc * d.Cin + row) * K + (K - 1 - kh)) * K + (K - 1 - kw)];
    }
    *base = ((size_t)(tap * ncs + cs) * planes * rows + row) * CS + (j ^ (((row >> 3) & 1) << 3));
    return v;
}

// AMAX_PASS: the first of the two launches of a refresh -- only the H2 descriptors work: every work item folds |source element| into
// the descriptor's absmax slot (the 8 bytes behind {s_w, 1 / s_w} in the tail of the prepared tensor, epoch `epoch`); the second
// launch derives s_w from it (h2_scale) and writes the planes.
template <bool AMAX_PASS>
__global__ __launch_bounds__(256) void weight_refresh_multi_kernel(const dwc_refresh_desc* __restrict__ descs,
                                                                   const int* __restrict__ chunk_desc,
                                                                   const unsigned* __restrict__ chunk_start, unsigned epoch) {
    const dwc_refresh_desc d = descs[chunk_desc[blockIdx.x]];
    const size_t i0 = chunk_start[blockIdx.x];
    const size_t i1 = min((size_t)d.n_items, i0 + DWC_OPT_CHUNK);
    const bool h2 = d.kind == DWC_REFRESH_H2_FWD || d.kind == DWC_REFRESH_H2_DGRAD;
    unsigned short* h2_planes = reinterpret_cast<unsigned short*>(d.dst);
    float* h2_tail = reinterpret_cast<float*>(h2_planes + 2 * d.n_items);                           // {s_w, 1 / s_w}, then the slot
    unsigned long long* h2_slot = reinterpret_cast<unsigned long long*>(h2_tail + 2);
    if constexpr (AMAX_PASS) {
        if (!h2) return;
        unsigned am = 0;
        for (size_t idx = i0 + threadIdx.x; idx < i1; idx += 256) {
            size_t base;
            am = max(am, dwc_abs_bits(split_layout_elem(d, idx, d.kind == DWC_REFRESH_H2_DGRAD, 2, &base)));
        }
        dwc_amax_wave_publish(h2_slot, epoch, am);
        return;
    }
    H2Scale sw = {1.f, 1.f};
    if (h2) {
        sw = h2_scale(h2_slot, epoch);
        if (i0 == 0 && threadIdx.x == 0) h2_tail[0] = sw.s, h2_tail[1] = sw.inv;
    }
    for (size_t idx = i0 + threadIdx.x; idx < i1; idx += 256) {
        switch (d.kind) {
            case DWC_REFRESH_H2_FWD:
            case DWC_REFRESH_H2_DGRAD: {
                size_t base;
                const float v = split_layout_elem(d, idx, d.kind == DWC_REFRESH_H2_DGRAD, 2, &base) * sw.s;
                const _Float16 h = (_Float16)v;
                const _Float16 l = (_Float16)(v - (float)h);
                h2_planes[base] = __builtin_bit_cast(unsigned short, h);
                h2_planes[base + (size_t)d.rows * 16] = __builtin_bit_cast(unsigned short, l);
                break;
            }
            case DWC_REFRESH_FWD_F32: reinterpret_cast<float*>(d.dst)[idx] = im2col_elem(d, idx, false); break;
            case DWC_REFRESH_DGRAD_F32: reinterpret_cast<float*>(d.dst)[idx] = im2col_elem(d, idx, true); break;
            case DWC_REFRESH_FWD_BF16: reinterpret_cast<bf16*>(d.dst)[idx] = (bf16)im2col_elem(d, idx, false); break;
            case DWC_REFRESH_DGRAD_BF16: reinterpret_cast<bf16*>(d.dst)[idx] = (bf16)im2col_elem(d, idx, true); break;
            case DWC_REFRESH_X3_FWD:
            case DWC_REFRESH_X3_DGRAD: {
                // conv_halo_x3.hip x3_weight_prepare_kernel: out[((tap*ncs + cs)*3 + plane)*rows + row][16], exact bf16 splits
                const int K = d.KH, rows = d.rows, kdim = d.kdim, CS = 16;
                const int ncs = (kdim + CS - 1) / CS;
                const int j = idx % CS;
                size_t r = idx / CS;
                const int row = r % rows;
                r /= rows;
                const int cs = r % ncs, tap = r / ncs;
                const int kh = tap / K, kw = tap - kh * K;
                const int kc = cs * CS + j;
                float v = 0.f;
                if (d.kind == DWC_REFRESH_X3_FWD) {
                    if (row < d.Cout && kc < d.Cin) v = d.src[(((size_t)row * d.Cin + kc) * K + kh) * K + kw];
                } else {
                    if (row < d.Cin && kc < d.Cout) v = d.src[(((size_t)kc * d.Cin + row) * K + (K - 1 - kh)) * K + (K - 1 - kw)];
                }
                const unsigned hb = __float_as_uint(v) & 0xffff0000u;
                const float r1 = v - __uint_as_float(hb);
                const unsigned mb = __float_as_uint(r1) & 0xffff0000u;
                const float r2 = r1 - __uint_as_float(mb);
                unsigned short* o = reinterpret_cast<unsigned short*>(d.dst);
                const size_t base = ((size_t)(tap * ncs + cs) * 3 * rows + row) * CS + (j ^ (((row >> 3) & 1) << 3));
                o[base] = (unsigned short)(hb >> 16);
                o[base + (size_t)rows * CS] = (unsigned short)(mb >> 16);
                o[base + 2 * (size_t)rows * CS] = (unsigned short)(__float_as_uint(r2) >> 16);
                break;
            }
            default: break;
        }
    }
}

}  // namespace

extern "C" {

int dwc_weight_refresh_multi(const dwc_refresh_desc* descs_dev, const int* chunk_desc_dev, const unsigned* chunk_start_dev, int n_chunks,
                             int has_h2, unsigned epoch, void* stream) {
    if (n_chunks <= 0) return DWC_OK;
    if (!descs_dev || !chunk_desc_dev || !chunk_start_dev) return DWC_EINVAL;
    if (has_h2) {      // two-plane f16 layouts: their filters' largest magnitudes first (the planes are scaled by them)
        hipLaunchKernelGGL(weight_refresh_multi_kernel<true>, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, descs_dev, chunk_desc_dev,
                           chunk_start_dev, epoch);
        DWC_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(weight_refresh_multi_kernel<false>, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, descs_dev, chunk_desc_dev,
                       chunk_start_dev, epoch);
    DWC_LAUNCH_CHECK();
    return DWC_OK;
}

}  // extern "C"
