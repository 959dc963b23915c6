/*
 * dwcgan_hip.h — C ABI of libdwcgan_hip.so: the gfx950 (MI355X) kernels under the DWC-GAN
 * training hot path.
 *
 * The reference has no native/FFI layer: its boundary is the Python import contract
 * `solver.Solver`, `networks.networks`, `networks.networks_v2`, and every
 * heavy operation is a torch.nn call.  Each entry point below replaces one of those torch
 * dispatch sites (cited per function as reference file:line); the Python mirror in
 * dwc-gan_amd/ binds them with ctypes.
 *
 * Conventions
 *  - all tensors are dense fp32 in HBM, activations NHWC: x[n][h][w][c]
 *  - conv weights are handed over in the state_dict layout OIHW and re-laid-out on device by
 *    the dwc_weight_prepare_* entry points into the K-contiguous layouts the kernels stream
 *  - channel counts on the data path must be multiples of 4 (16-byte vector accesses); the
 *    3-channel images travel as NHWC4 with a zero 4th plane
 *  - pointers are borrowed device pointers; the library never allocates, frees or synchronises;
 *    scratch is passed in by the caller (`ws`, `ws_bytes`); `stream` is a hipStream_t
 *  - return value: 0 on success, a negative DWC_E* code otherwise; nothing throws
 *  - thread-safe per stream: no device-side state of the library's own (scratch, instance-norm tickets and the LSTM status word
 *    are caller memory, one set per concurrently used stream); host side only write-once caches of switches / occupancy queries
 */
#ifndef DWCGAN_HIP_H
#define DWCGAN_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DWC_OK 0
#define DWC_EINVAL (-1)   /* unsupported shape / argument */
#define DWC_EWORKSPACE (-2) /* scratch buffer too small */
#define DWC_ELAUNCH (-3)  /* hipLaunch failed */

/* activation codes fused into conv epilogues (reference networks.py:556-571) */
#define DWC_ACT_NONE 0
#define DWC_ACT_RELU 1
#define DWC_ACT_LRELU 2    /* slope 0.1, reference networks.py:559 */
#define DWC_ACT_TANH 3
#define DWC_ACT_SIGMOID 4
#define DWC_ACT_HEADS 5    /* channels 0..2 tanh, channel 3 sigmoid: image_content+image_attention
                              fused into one 4-channel conv (reference networks_v2.py:159-160) */
#define DWC_ACT_HEADS8 6   /* the same heads on an 8-plane (NHWC8, bf16) image: planes 0..2 tanh, 3 sigmoid, 4..7 zero */

/* Version of this C ABI: bumped with every change of an entry point's signature or of a structure passed through it (r05: 5 --
 * dwc_weight_refresh_multi gained has_h2 / epoch, the dwc_h2_* / dwc_*_amax entry points).  A binding must refuse a library that
 * reports another number: symbols alone do not tell a changed argument list (hipdwc/_lib.py does). */
#define DWC_ABI_VERSION 8
int dwc_version(void);
/* The fp32 im2col kernels (dwc_conv2d_fwd / _bwd_data* / _bwd_weight*, ring strips) take their inner products as exact three-way
 * bf16 split products on the bf16 matrix cores by default (r04; fp32 operands, results and accumulation -- see
 * csrc/conv_halo_x3.hip for the arithmetic).  dwc_x3_gemm_mode(mode >= 0) sets the process-wide switch (bit 0: forward /
 * data-gradient GEMM, bit 1: weight gradient; 0 = native fp32 MFMA) and returns the previous value; mode < 0 only queries.
 * Environment at first use: DWC_X3_GEMM, DWC_X3_WGRAD (0 / 1). */
int dwc_x3_gemm_mode(int mode);

/* ---- weight re-layout --------------------------------------------------------------------
 * The GEMM kernels stream weights as one row per GEMM column with K contiguous and
 * zero-padded to a multiple of 32:
 *   forward : [co][ (kh*KW+kw)*cin_pad + ci ]               = W[co][ci][kh][kw]
 *   dgrad s1: [ci][ (kh'*KW+kw')*cout_pad + co ]            = W[co][ci][KH-1-kh'][KW-1-kw']
 *   dgrad s2: [ph*2+pw][ci][ (th*2+tw)*cout_pad + co ]      = W[co][ci][ph+2*th][pw+2*tw]   (4x4 only)
 * dwc_weight_prepared_elems gives the float count of the prepared buffer. */
size_t dwc_weight_prepared_elems(int Cout, int Cin, int KH, int KW, int stride, int cout_pad, int cin_pad,
                                 int for_dgrad);
int dwc_weight_prepare_fwd(const float* w_oihw, float* w_prepared, int Cout, int Cin, int KH, int KW,
                           int cout_pad, int cin_pad, void* stream);
int dwc_weight_prepare_dgrad(const float* w_oihw, float* w_prepared, int Cout, int Cin, int KH, int KW,
                             int stride, int cout_pad, int cin_pad, void* stream);

/* ---- convolution (replaces nn.ReflectionPad2d + nn.Conv2d [+ activation];
 *      reference networks.py:579-585, call sites networks.py:432-441,514-515,90-98,
 *      networks_v2.py:106-112,155,159-160; also nn.Linear as a 1x1 conv, networks.py:595) ---- */
/* y = act(conv(reflect_pad(x, pad), w) + bias).  x:[B,H,W,Cin] y:[B,Ho,Wo,Cout], w prepared by dwc_weight_prepare_fwd.
 * Scratch (dwc_conv2d_fwd_ws_bytes) is only used when the product is split along K (few output
 * tiles, long contraction); with less scratch than that the kernel runs un-split. */
size_t dwc_conv2d_fwd_ws_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int dwc_conv2d_fwd(const float* x, const float* w_prepared, const float* bias, float* y,
                   int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                   int act, void* ws, size_t ws_bytes, void* stream);
/* Data gradient, step 1: gradient w.r.t. the reflect-PADDED input image,
 * dxp:[B,H+2*pad,W+2*pad,Cin], from the pre-activation gradient dy:[B,Ho,Wo,Cout]
 * (w in dgrad layout).  With pad == 0 this already is dx.  Scratch as for the forward. */
size_t dwc_conv2d_bwd_data_ws_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int dwc_conv2d_bwd_data(const float* dy, const float* w_dgrad, float* dxp,
                        int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                        void* ws, size_t ws_bytes, void* stream);
/* Data gradient, step 2: adjoint of the reflect padding — fold dxp back onto dx:[B,H,W,C]. */
int dwc_reflect_pad_adjoint(const float* dxp, float* dx, int B, int H, int W, int C, int pad,
                            void* stream);
/* dwc_conv2d_bwd_data followed by dwc_reflect_pad_adjoint as one call: where the GEMM runs unsplit the interior of the padded
 * gradient image goes straight into dx and only its border ring through the scratch image dxp ([B,H+2pad,W+2pad,Cin]), which a
 * band kernel folds onto dx (one pass over the tensor instead of three). */
int dwc_conv2d_bwd_data_fold(const float* dy, const float* w_dgrad, float* dxp, float* dx, int B, int H, int W, int Cin, int Cout, int KH,
                             int KW, int stride, int pad, void* ws, size_t ws_bytes, void* stream);
/* Data gradient of a stride-1 "same" convolution (square filter, 2*pad == K-1: the 3x3 ResBlock, 5x5 upsampling and 7x7
 * head convolutions, reference networks.py:514-515, networks_v2.py:155,159-160) in ONE call, dx:[B,H,W,Cin] final.
 * The interior of the padded gradient image is computed on the H x W grid straight into dx; of the padding ring only
 * four thin strips are formed (each restricted to the filter rows/columns that reach real dY pixels) and folded onto
 * dx by the reflect rule.  w_dgrad: dwc_weight_prepare_dgrad of W; w_dgrad_t: the same of W with its two filter axes
 * swapped.  Cout (channels of dy) a power of two >= 32. */
size_t dwc_conv2d_bwd_data_same_ws_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW, int pad);
int dwc_conv2d_bwd_data_same(const float* dy, const float* w_dgrad, const float* w_dgrad_t, float* dx,
                             int B, int H, int W, int Cin, int Cout, int KH, int KW, int pad,
                             void* ws, size_t ws_bytes, void* stream);
/* Only the ring part of dwc_conv2d_bwd_data_same: dx already holds the interior (dwc_x3_ / dwc_h2_conv2d_same_add_ws with the zero rule). */
int dwc_conv2d_bwd_data_ring(const float* dy, const float* w_dgrad, const float* w_dgrad_t, float* dx,
                             int B, int H, int W, int Cin, int Cout, int KH, int KW, int pad,
                             void* ws, size_t ws_bytes, void* stream);
/* Data gradient w.r.t. an NHWC4 IMAGE dx:[B,H,W,4] of a stride-1 "same" convolution (2*pad == K-1; the 7x7 stems,
 * reference networks.py:432, networks_v2.py:106, reached when generated images are re-encoded, solver.py:176-180).
 * 8 adjacent pixels x 4 channels are produced as 32 GEMM columns.  w_wide: dwc_weight_prepare_fwd layout of the bank
 * [32 = p*4+ci][Cout][KH][KW+7] whose copy p is the flipped filter W[co][ci][KH-1-kh][KW-1-kw] shifted right by p
 * taps.  The padded gradient image is built in ws and folded onto dx (reflect-pad adjoint) by the same call. */
size_t dwc_conv2d_bwd_data_image_ws_bytes(int B, int H, int W, int Cout, int KH, int KW, int pad);
int dwc_conv2d_bwd_data_image(const float* dy, const float* w_wide, float* dx,
                              int B, int H, int W, int Cout, int KH, int KW, int pad,
                              void* ws, size_t ws_bytes, void* stream);
/* dw (written in OIHW, the state_dict layout, [Cout_real][Cin_real][KH][KW]) from x and dy.
 * Cin/Cout are the padded data-path channel counts, cin_real/cout_real the parameter's. */
size_t dwc_conv2d_bwd_weight_ws_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW,
                                      int stride, int pad);
int dwc_conv2d_bwd_weight(const float* x, const float* dy, float* dw_oihw,
                          int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                          int cin_real, int cout_real, void* ws, size_t ws_bytes, void* stream);
/* Forward / weight gradient with per-axis stride and reflect pad.  Used for the image heads
 * (reference networks_v2.py:159-160): 8 horizontally adjacent output pixels x 4 channels are
 * computed as 32 "wide" output channels of a 7x14, stride-(1,8) convolution, so that the
 * 4-channel product fills a 32-wide MFMA tile (2x padding instead of 8x). */
int dwc_conv2d_fwd_ex(const float* x, const float* w_prepared, const float* bias, float* y,
                      int B, int H, int W, int Cin, int Cout, int KH, int KW,
                      int stride_h, int stride_w, int pad_h, int pad_w, int act, void* stream);
size_t dwc_conv2d_bwd_weight_ex_ws_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW,
                                         int stride_h, int stride_w, int pad_h, int pad_w);
int dwc_conv2d_bwd_weight_ex(const float* x, const float* dy, float* dw_oihw,
                             int B, int H, int W, int Cin, int Cout, int KH, int KW,
                             int stride_h, int stride_w, int pad_h, int pad_w,
                             int cin_real, int cout_real, void* ws, size_t ws_bytes, void* stream);
/* g = dy * act'(y) and db[c] = sum over rows of g (rows = B*Ho*Wo).  db may be NULL. */
size_t dwc_act_bwd_bias_ws_bytes(int rows, int C);
int dwc_act_bwd_bias(const float* dy, const float* y, float* g, float* db, int rows, int C, int act,
                     void* ws, size_t ws_bytes, void* stream);

/* ---- instance norm / AdaIN (reference networks.py:545 nn.InstanceNorm2d and
 *      networks.py:706-719 AdaptiveInstanceNorm2d; residual add networks.py:518-522) --------- */
/* y = relu?( (x-mean[n,c])*rstd[n,c]*gamma[n,c] + beta[n,c] ) + residual?
 * gamma/beta: [B*C] or NULL (plain IN).  mean/rstd [B*C] are outputs kept for the backward.
 * (ABI 8 dropped the `tickets` argument of the six instance-norm entry points and the ticket-row size query: rounds 3-4 finalised
 * the statistics inside the partial launch through a caller-owned ticket row; measured at batch 16 that cost 3-5x what a parallel
 * finalise launch costs (csrc/norm.hip), so the statistics are finalised by their own small launch.)  The library keeps no mutable state. */
size_t dwc_instnorm_ws_bytes(int B, int HW, int C);
int dwc_instnorm_fwd(const float* x, const float* gamma, const float* beta, const float* residual,
                     float* y, float* mean, float* rstd, int B, int HW, int C, float eps, int relu,
                     void* ws, size_t ws_bytes, void* stream);
/* dx (and dgamma/dbeta [B*C] when gamma != NULL) given dy w.r.t. the (pre-residual) output. */
int dwc_instnorm_bwd(const float* dy, const float* x, const float* mean, const float* rstd,
                     const float* gamma, const float* beta, float* dx, float* dgamma, float* dbeta,
                     int B, int HW, int C, int relu, void* ws, size_t ws_bytes, void* stream);

/* ---- nn.Linear (+ ReLU) on few rows (r06, ABI 8): the style -> AdaIN-parameter MLP (reference networks.py:491-503, LinearBlock
 *      :587-634) and the style encoder's mapping (networks_v2.py:116-121) ------------------------------------------------------
 * y[M][N] = act(x[M][K] . w[N][K]^T + bias[N]), fp32 row-major, w as nn.Linear stores it (no prepared layout); relu != 0: ReLU.
 * One wave per 16 x 16 output tile on the exact fp32 matrix instruction (v_mfma_f32_16x16x4_f32), operands straight from memory: no
 * LDS, no scratch.  _ok: N and K multiples of 16, M <= 4096 (larger problems belong on dwc_conv2d_fwd as a 1x1 convolution).
 * dwc_linear_small_bwd: dx[M][K] (NULL: skipped), dw[N][K] and db[N] (NULL: skipped) from dy[M][N]; y_relu = the forward's output when
 * it had the ReLU (the mask is applied while dy is read), else NULL. */
int dwc_linear_small_ok(int M, int N, int K);
int dwc_linear_small_fwd(const float* x, const float* w, const float* bias, float* y, int M, int N, int K, int relu, void* stream);
int dwc_linear_small_bwd(const float* dy, const float* y_relu, const float* x, const float* w, float* dx, float* dw, float* db, int M,
                         int N, int K, void* stream);

/* ---- MUNIT LayerNorm (reference networks.py:736-752: per-sample mean, UNBIASED std,
 *      (x-mean)/(std+eps), per-channel gamma/beta) ------------------------------------------ */
size_t dwc_layernorm_ws_bytes(int B, int HW, int C);
int dwc_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y,
                      float* mean, float* inv, int B, int HW, int C, float eps, int relu,
                      void* ws, size_t ws_bytes, void* stream);
int dwc_layernorm_bwd(const float* dy, const float* x, const float* mean, const float* inv,
                      const float* gamma, const float* beta, float* dx, float* dgamma, float* dbeta,
                      int B, int HW, int C, float eps, int relu, void* ws, size_t ws_bytes, void* stream);

/* ---- resampling (reference networks_v2.py:154 nn.Upsample(x2, bilinear);
 *      networks.py:113 F.interpolate(x0.5, bilinear) == 2x2 mean) ---------------------------- */
int dwc_upsample2x_fwd(const float* x, float* y, int B, int H, int W, int C, void* stream);
int dwc_upsample2x_bwd(const float* dy, float* dx, int B, int H, int W, int C, void* stream);
int dwc_avgpool2_fwd(const float* x, float* y, int B, int H, int W, int C, void* stream);
int dwc_avgpool2_bwd(const float* dy, float* dx, int B, int H, int W, int C, void* stream);

/* ---- image boundary: NCHW 3-channel <-> NHWC4 ------------------------------------------- */
int dwc_pack_nchw_to_nhwc4(const float* x_nchw, float* y_nhwc4, int B, int C, int H, int W, void* stream);
int dwc_unpack_nhwc4_to_nchw(const float* x_nhwc4, float* y_nchw, int B, int C, int H, int W, void* stream);

/* ---- VGG16 perceptual loss (reference networks.py:639-688 Vgg16, solver.py:242-247 compute_vgg_loss): frozen 3x3
 *      convolutions with ZERO padding (nn.Conv2d(padding=1)) + ReLU and 2x2 max pooling.  The weights are frozen, so only
 *      the forward and the data gradient exist; instance norm of the features is dwc_instnorm_*. ---- */
int dwc_conv2d_fwd_zeropad(const float* x, const float* w_prepared, const float* bias, float* y,
                           int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                           int act, void* ws, size_t ws_bytes, void* stream);
/* dx:[B,H,W,Cin] from dy:[B,H,W,Cout] (stride 1, 2*pad == K-1); w_dgrad as for dwc_conv2d_bwd_data. */
size_t dwc_conv2d_bwd_data_zeropad_ws_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW, int pad);
int dwc_conv2d_bwd_data_zeropad(const float* dy, const float* w_dgrad, float* dx,
                                int B, int H, int W, int Cin, int Cout, int KH, int KW, int pad,
                                void* ws, size_t ws_bytes, void* stream);
/* F.max_pool2d(x, 2, 2) on NHWC (reference networks.py:666,671,677); ties to the first element in scan order. */
int dwc_maxpool2_fwd(const float* x, float* y, int B, int H, int W, int C, void* stream);
int dwc_maxpool2_bwd(const float* x, const float* dy, float* dx, int B, int H, int W, int C, void* stream);

/* ---- attention blend (reference solver.py:148,161,170,179-180,192,330-331):
 *      out[...,0:3] = img*att + real*(1-att) on NHWC4 images (att = channel 3 of `heads`) ------ */
int dwc_blend_fwd(const float* heads, const float* real, float* out, int npix, void* stream);
int dwc_blend_bwd(const float* dout, const float* heads, const float* real, float* dheads, int npix,
                  void* stream);

/* ---- GMM style-space KL term, all attributes at once (reference gmm.py:13-22, called at solver.py:218-219; r06) ----
 * out[0] = sum_k mean_b sum_d 0.5 (log(sigma / e^lv) + (e^lv + (mu - centre[b][k])^2) / sigma - 1); mu, lv: [B][K][D] contiguous,
 * centre: [B][centre_stride >= K] (the +-1 component centres from the labels).  One launch each way; dmu / dlv may be NULL. */
int dwc_gmm_kl_sp_fwd(const float* mu, const float* lv, const float* centre, int centre_stride, int B, int K, int D, float sigma,
                      float* out_scalar, void* stream);
int dwc_gmm_kl_sp_bwd(const float* mu, const float* lv, const float* centre, int centre_stride, int B, int K, int D, float sigma,
                      const float* dout_scalar, float* dmu, float* dlv, void* stream);

/* ---- mean |a-b| (reference solver.py:113-114) ----------------------------------------------
 * skip4 != 0: the buffers are NHWC4 images; every 4th element (the pad / attention plane) is
 * ignored and the mean is over the 3 image planes only (n*3/4 elements). */
size_t dwc_l1_ws_bytes(size_t n);
int dwc_l1_mean_fwd(const float* a, const float* b, float* out_scalar, size_t n, int skip4, void* ws,
                    size_t ws_bytes, void* stream);
/* da = sign(a-b) * dout[0] / count, db = -da (either may be NULL) */
int dwc_l1_mean_bwd(const float* a, const float* b, const float* dout_scalar, float* da, float* db,
                    size_t n, int skip4, void* stream);

/* ---- text encoder: recurrent part of the packed bi-LSTM (reference networks_v2.py:199-203, 226-233; nn.LSTM over a
 *      pack_padded_sequence).  The caller does the input projections of all steps as one GEMM; these run the T sequential
 *      steps (one launch per step for both directions) with the cell update fused.  Packed semantics by masking: sample b is
 *      active at step t iff t < lens[b]; inactive slots of out / c are written as zeros and a reverse sequence starts from
 *      zeros at its own last token.  Gate order i, f, g, o as in torch.  H % 4 == 0. ---- */
/* xproj:[dirs][T][B][4H] (x W_ih^T + b_ih + b_hh), w_hh:[dirs][4H][H], lens:[B] int32 (device).
 * Outputs out (h_t), c: [dirs][T][B][H]; gates: [dirs][T][B][4H] activated gates kept for the backward. */
int dwc_lstm_fwd(const float* xproj, const float* w_hh, const int* lens, float* out, float* c, float* gates,
                 int T, int B, int H, int dirs, void* stream);
/* d_out / d_c: gradients arriving at h_t / c_t from outside the recurrence, [dirs][T][B][H] or NULL.
 * w_hh_t:[dirs][H][4H] (transposed).  Output dgates:[dirs][T][B][4H] = gradient of the pre-activation gates, from which
 * the caller forms dW_ih, dW_hh, db and dx with three GEMMs.  dc_carry:[dirs][B][H] scratch. */
/* The same recurrence, ALL T steps of both directions in ONE launch: persistent workgroups, W_hh slices and cell state in
 * registers, h_t handed between workgroups inside the launch (write-through stores + agent-scope arrival counter + sc1 loads;
 * bounded polls).  ws >= dwc_lstm_seq_ws_bytes(B, dirs) (a multiple of 16 bytes): [0] timeout word, then the counters; zeroed
 * by the call.  Every workgroup of the launch must be RESIDENT at once: the call queries the device (CU count x occupancy of
 * the kernel at its LDS request; `max_workgroups` > 0 lowers that capacity further, e.g. to leave CUs to collective kernels
 * running beside it) and returns DWC_EINVAL -- nothing launched, use dwc_lstm_fwd / dwc_lstm_bwd -- when the grid does not fit
 * or the shape is not taken (H > 320).  A rendez-vous that is missed all the same (bounded spin, about a second) fails LOUDLY:
 * the workgroup overwrites every result slot it owns with NaN and ORs 1 (forward) / 2 (backward) into *status, a
 * caller-owned device word that the library never clears (may be NULL). */
size_t dwc_lstm_seq_ws_bytes(int B, int dirs);
int dwc_lstm_seq_fwd(const float* xproj, const float* w_hh, const int* lens, float* out, float* c, float* gates, int T, int B, int H,
                     int dirs, void* ws, size_t ws_bytes, unsigned* status, int max_workgroups, void* stream);
int dwc_lstm_seq_bwd(const float* d_out, const float* d_c, const float* w_hh_t, const int* lens, const float* c, const float* gates,
                     float* dgates, int T, int B, int H, int dirs, void* ws, size_t ws_bytes, unsigned* status, int max_workgroups,
                     void* stream);
int dwc_lstm_bwd(const float* d_out, const float* d_c, const float* w_hh_t, const int* lens, const float* c,
                 const float* gates, float* dgates, float* dc_carry, int T, int B, int H, int dirs, void* stream);

/* ---- fused Adam (+coupled L2) + EMA (reference solver.py:62-68,240,353; utils.py:52-54) ----- */
/* One launch over a flat parameter arena.  p,g,m,v,ema: [n].  step is the 1-based Adam step.
 * ema may be NULL.  ema update is the reference's lerp(param, ema, beta) AFTER the step only when
 * do_ema != 0 (the reference runs it once per iteration, after both optimisers). */
int dwc_adam_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int step, void* stream);
int dwc_ema_lerp(const float* p, float* ema, size_t n, float beta, void* stream);

/* Multi-tensor forms: one launch for every parameter tensor of a network.  The descriptor arrays
 * and the chunk maps live in DEVICE memory (the caller uploads them; gradient pointers and the
 * per-tensor bias corrections change every step, the chunk maps never).  Workgroup b updates
 * DWC_OPT_CHUNK consecutive elements of tensor chunk_tensor[b] starting at chunk_start[b].
 * A tensor whose g is NULL is skipped entirely — Adam's behaviour for parameters without a
 * gradient (SURVEY.md section 7 quirk viii).  step_size = lr/(1-beta1^t), bc2_sqrt = sqrt(1-beta2^t)
 * with t that tensor's own step count. */
#define DWC_OPT_CHUNK 8192
typedef struct {
    float* p;
    const float* g;
    float* m;
    float* v;
    unsigned long long n;
    float step_size;
    float bc2_sqrt;
} dwc_adam_tensor;
typedef struct {
    const float* p;
    float* ema;
    unsigned long long n;
} dwc_ema_tensor;
int dwc_adam_multi(const dwc_adam_tensor* tensors_dev, const int* chunk_tensor_dev,
                   const unsigned* chunk_start_dev, int n_chunks, double beta1, double beta2, double eps,
                   double weight_decay, void* stream);
int dwc_ema_multi(const dwc_ema_tensor* tensors_dev, const int* chunk_tensor_dev,
                  const unsigned* chunk_start_dev, int n_chunks, float beta, void* stream);

/* Adversarial loss tail of one discriminator scale in one launch (reference networks.py:116-170: LSGAN on the 1-channel src map,
 * BCE-with-logits on the attribute logits).  The batch is `segs` <= 4 segments of B samples; labels [B][ncls] are shared:
 *   out = sum_s w_src[s] * mean_s((src - target[s])^2) + w_cls[s] * mean_s(bce_with_logits(cls, labels))
 * src [segs*B][src_per_sample], cls [segs*B][ncls], all fp32; _bwd writes d out / d src and d out / d cls scaled by dout[0]. */
typedef struct {
    float target[4], w_src[4], w_cls[4];
} dwc_adv_spec;
int dwc_adv_tail_fwd(const float* src, const float* cls, const float* labels, float* out, int segs, int B, int src_per_sample, int ncls,
                     dwc_adv_spec spec, void* stream);
int dwc_adv_tail_bwd(const float* src, const float* cls, const float* labels, const float* dout, float* dsrc, float* dcls, int segs,
                     int B, int src_per_sample, int ncls, dwc_adv_spec spec, void* stream);

/* Refresh of every prepared weight layout of a network in ONE launch behind the optimiser step (SURVEY.md section 8(f) rank 1;
 * reference solver.py:240,353).  One descriptor per (OIHW master weight, prepared tensor, layout); workgroup b rebuilds
 * DWC_OPT_CHUNK work items of descriptor chunk_desc[b] starting at chunk_start[b].  Layouts and element formulas are those of
 * the single-layout entry points: FWD / DGRAD = dwc_weight_prepare_fwd / _dgrad (fp32) and dwc_bf16_weight_prepare_fwd /
 * _dgrad (work item = output element; n_items = rows * Kp, x 4 parity classes for the stride-2 data gradient; transpose_hw:
 * the filter with its two spatial axes swapped), X3 = dwc_x3_weight_prepare (work item = one of K*K*ceil(kdim/16)*rows*16
 * source slots).  (Kinds 6 / 7 were the Winograd F(2x2,3x3) banks of rounds 1-4; the family was removed in r05.) */
enum { DWC_REFRESH_FWD_F32 = 0, DWC_REFRESH_DGRAD_F32 = 1, DWC_REFRESH_FWD_BF16 = 2, DWC_REFRESH_DGRAD_BF16 = 3,
       DWC_REFRESH_X3_FWD = 4, DWC_REFRESH_X3_DGRAD = 5,
       /* r05: dwc_h2_weight_prepare (two f16 planes, n_items = K*K*ceil(kdim/16)*rows*16 source slots).  The prepared tensor ends
        * with {s_w, 1 / s_w} (fp32) and the filter's absmax slot (8 bytes, zero when the tensor is created): has_h2 != 0 makes
        * dwc_weight_refresh_multi raise those slots (epoch `epoch`, larger at every call) in a launch of its own first. */
       DWC_REFRESH_H2_FWD = 8, DWC_REFRESH_H2_DGRAD = 9 };
typedef struct {
    const float* src;            /* fp32 OIHW master weight [Cout][Cin][KH][KW] */
    void* dst;                   /* prepared tensor */
    unsigned long long n_items;
    int kind;                    /* DWC_REFRESH_* */
    int Cout, Cin, KH, KW, stride, cout_pad, cin_pad, Kp;
    int rows, kdim;              /* X3: rows of the prepared matrix, contraction length */
    int transpose_hw;            /* DGRAD: swap the filter's spatial axes */
    int reserved;
} dwc_refresh_desc;
int dwc_weight_refresh_multi(const dwc_refresh_desc* descs_dev, const int* chunk_desc_dev, const unsigned* chunk_start_dev,
                             int n_chunks, int has_h2, unsigned epoch, void* stream);

/* ---- fp32 convolutions on the bf16 matrix cores by exact three-way operand splits (conv_halo_x3.hip) --------------------
 * An fp32 value is exactly the sum of three bf16 values (truncate / subtract twice); bf16 x bf16 products are exact in the
 * fp32 accumulator; the six leading cross products reproduce a*b to 2^-23 relative, the accuracy of one fp32 rounding.
 * Same layers and halo form as dwc_bf16_conv2d_same_halo (reference networks.py:514-515, networks_v2.py:153-156) but fp32
 * NHWC tensors in and out.  dwc_x3_weight_prepare splits the OIHW filter once per optimiser step: `rows` = row count of the
 * prepared matrix (>= Cout forward, >= Cin for the data gradient); elems = bf16 elements of the prepared tensor. */
int dwc_x3_conv2d_same_ok(int B, int H, int W, int Cin, int Cout, int K);
size_t dwc_x3_weight_prepared_elems(int rows, int kdim, int K);
int dwc_x3_weight_prepare(const float* w_oihw, void* out, int Cout, int Cin, int K, int rows, int dgrad, void* stream);
int dwc_x3_conv2d_same(const float* x, const void* w_prepared, const float* bias, float* y, int B, int H, int W, int Cin, int N,
                       int rows, int K, int act, int reflect, void* stream);
/* dwc_x3_conv2d_same with `add` ([B,H,W,N] fp32 or NULL) added behind bias and activation: the identity-branch gradient of a
 * ResBlock (reference networks.py:521) rides on the data gradient of its first convolution instead of costing a pass. */
int dwc_x3_conv2d_same_add(const float* x, const void* w_prepared, const float* bias, const float* add, float* y, int B, int H, int W,
                           int Cin, int N, int rows, int K, int act, int reflect, void* stream);
/* fp32 7x7 convolutions from 64 channels to the 4 planes of an NHWC4 image as split products (conv_narrow_x3.hip, r04): the fused
 * image heads of the decoder (reference networks.py:218-246) and the image gradient of the 7x7 stems (networks.py:579-585
 * backward).  y[B][OH][OWg][32] (8 pixels x 4 planes per group) = act(sum over the KH x KWW wide taps and 64 channels + bias32), the
 * input window of (oy, gx) starting at (oy + off_h, 8*gx + off_w) of x[B][IH][IW][64].  w_frag: the [32][64][KH][KWW] wide bank
 * (copy p of the filter shifted right by p taps) as three exact bf16 planes in MFMA-fragment order
 * [slab q of 16 channels][tap kh*KWW+u][plane][lane = half*32 + bank row][8 channels 16q + 8*half ..]
 * (dwc_x3_conv2d_narrow_weight_elems bf16 elements).  reflect != 0: reflect rule (forward), else zero rule (image gradient on the
 * padded grid, folded by dwc_reflect_pad_adjoint_pitch with pitch = 8 * OWg).  KH = 7, KWW = 14, Cin = 64 only (_ok). */
int dwc_x3_conv2d_narrow_ok(int B, int IH, int IW, int Cin, int OH, int OWg, int KH, int KWW);
size_t dwc_x3_conv2d_narrow_weight_elems(int KH, int KWW);
int dwc_x3_conv2d_narrow(const float* x, const void* w_frag, const float* bias32, float* y, int B, int IH, int IW, int Cin, int OH,
                         int OWg, int KH, int KWW, int off_h, int off_w, int act, int reflect, void* stream);
int dwc_reflect_pad_adjoint_pitch(const float* dxp, float* dx, int B, int H, int W, int C, int pad, int pitch, void* stream);
/* fp32 7x7 convolutions from the 4 planes of an NHWC4 image to 64 channels as split products (conv_narrow_x3.hip, r04; the fp32 twin
 * of dwc_bf16_conv2d_stem): the stems forward (reference networks.py:163-166 / :60-66; reflect rule, off = -3) and the data
 * gradient of the image heads (zero rule on the padded grid, off = -6; with crop = pad the interior goes straight to `inner` and
 * only the border ring to y, folded by dwc_reflect_pad_adjoint_band).  w_steps: the OIHW filter [64][4][7][7] as three exact bf16
 * planes [plane][13 k-steps][64][16], element (k-step j, channel co, tap 4j + 2h + t, image plane p) at
 * ((h ^ ((co>>3)&1))*8 + 4t + p), taps beyond 48 zero.  act: none / relu / lrelu. */
int dwc_x3_gather_split(const float* src, const int* idx, void* out_3planes, int n, void* stream);   /* out[p][i] = plane p of src[idx[i]] (idx < 0: 0) */
int dwc_x3_conv2d_stem_ok(int B, int IH, int IW, int OH, int OW, int K, int act);
size_t dwc_x3_conv2d_stem_weight_elems(void);
int dwc_x3_conv2d_stem(const float* x, const void* w_steps, const float* bias, float* y, int B, int IH, int IW, int OH, int OW, int K,
                       int off, int act, int reflect, void* stream);
int dwc_x3_conv2d_stem_crop(const float* x, const void* w_steps, const float* bias, float* y, float* inner, int crop, int B, int IH, int IW,
                            int OH, int OW, int K, int off, int act, int reflect, void* stream);
/* dwc_x3_conv2d_stem raising the absmax slot of y from its store pass (r05; see dwc_instnorm_fwd_amax; y_amax NULL: the plain call) */
int dwc_x3_conv2d_stem_amax(const float* x, const void* w_steps, const float* bias, float* y, void* y_amax, unsigned y_epoch, int B, int IH,
                            int IW, int OH, int OW, int K, int off, int act, int reflect, void* stream);
int dwc_reflect_pad_adjoint_band(const float* dxp, float* dx, int B, int H, int W, int C, int pad, void* stream);
/* r05: weight gradient of the two 7x7 layer shapes in fp32 as exact split products -- the fp32 twin of dwc_bf16_conv7_smallk_wgrad
 * (reference networks_v2.py:106,159-160, networks.py:432 through autograd): between an NHWC4 fp32 image img4:[B][H][W][4] and a
 * 64-channel fp32 tensor t64:[B][H][W][64], pad 3, reflect padding in the forward.  heads == 0 (stems, 4 -> 64): img4 = x, t64 = dY,
 * dw:[64][planes][7][7]; heads != 0 (image heads, 64 -> 4): img4 = dY (pre-activation gradient), t64 = x, dw:[planes][64][7][7].
 * planes <= 4 real planes are written.  Scratch: dwc_x3_conv7_smallk_wgrad_ws_bytes. */
size_t dwc_x3_conv7_smallk_wgrad_ws_bytes(int B, int H, int W, int heads);
int dwc_x3_conv7_smallk_wgrad(const float* img4, const float* t64, float* dw, int B, int H, int W, int planes, int heads, void* ws,
                              size_t ws_bytes, void* stream);
/* Small launches (r04): when a shape yields at most 256 tiles of 256 pixels x 64 channels (3x3 256->256 on 32x32 at batch 16) the
 * two-workgroups-per-CU kernel would run one workgroup per CU; the _ws forms cut such launches along the CONTRACTION instead --
 * two workgroups per tile, each half of the channel slabs; the first to finish leaves its half sum in `ws`, the second adds it to
 * its own (a + b = b + a: no dependence on arrival order) and applies bias / activation / add.  dwc_x3_conv2d_ksplit_ws_bytes:
 * bytes of `ws` wanted for a shape (stride 1: K in {3,5}; stride 2: the 4x4 layers, K ignored; 0 = the launch is not split);
 * `tickets`: dwc_x3_conv2d_ksplit_ticket_words() 32-bit words owned by the caller PER STREAM, zero before the first call and
 * left at zero by every call.  The LAST word is a sticky status: bit 0 is set when a tile's hand-off expired (a ticket left dirty by
 * an aborted launch) or its two halves ran on different XCDs -- that tile's result is NaN; the caller should then stop, or zero
 * the whole row before the next call.  ws / tickets NULL or ws_bytes too small: the plain launch.  DWC_X3_KSPLIT=0 disables the
 * split. */
size_t dwc_x3_conv2d_ksplit_ws_bytes(int B, int H, int W, int Cin, int N, int K, int stride);
int dwc_x3_conv2d_ksplit_ticket_words(void);
int dwc_x3_conv2d_same_add_ws(const float* x, const void* w_prepared, const float* bias, const float* add, float* y, int B, int H, int W,
                              int Cin, int N, int rows, int K, int act, int reflect, void* ws, size_t ws_bytes, unsigned* tickets,
                              void* stream);
int dwc_x3_conv2d_s2_ws(const float* x, const void* w_prepared, const float* bias, float* y, int B, int H, int W, int Cin, int N, int rows,
                        int act, void* ws, size_t ws_bytes, unsigned* tickets, void* stream);
/* The 4x4 stride-2 reflect-pad-1 layers (reference networks.py:90,94,437, networks_v2.py:107-111) on the same kernel: 2x2 taps
 * per input-pixel parity over the space-to-depth image, the space-to-depth done by the patch gather.  x:[B,H,W,Cin] ->
 * y:[B,H/2,W/2,N]; H, W multiples of 32, Cin a multiple of 16; w_prepared = dwc_x3_weight_prepare(K = 4, forward). */
int dwc_x3_conv2d_s2_ok(int B, int H, int W, int Cin, int Cout);
int dwc_x3_conv2d_s2(const float* x, const void* w_prepared, const float* bias, float* y, int B, int H, int W, int Cin, int N, int rows,
                     int act, void* stream);
/* Data gradient of the same 4x4 stride-2 reflect-pad-1 layers in halo form (r04; reference networks.py:90,94,437,
 * networks_v2.py:107-111 through autograd).  dwc_x3_conv2d_s2_bwd_data writes the INTERIOR -- all H x W pixels of dx:[B,H,W,Cin]
 * fp32 from dy:[B,H/2,W/2,Cout] -- as four output-parity classes of 2x2-tap zero-padded split-product convolutions over dY
 * (w_prepared = dwc_x3_weight_prepare(K = 4, dgrad = 1), rows >= Cin); dwc_conv2d_bwd_data_s2_ring then forms the border ring of
 * the PADDED gradient image as eight thin fp32 GEMM strips in the scratch image dxp ([B][H+2][W+2][Cin]; only its ring is touched)
 * and folds it onto dx by the reflect rule (w_dgrad = dwc_weight_prepare_dgrad(4x4, stride 2)).  No further scratch. */
int dwc_x3_conv2d_s2_bwd_data_ok(int B, int H, int W, int Cin, int Cout);
int dwc_x3_conv2d_s2_bwd_data(const float* dy, const void* w_prepared, float* dx, int B, int H, int W, int Cin, int Cout, int rows,
                              void* stream);
int dwc_conv2d_bwd_data_s2_ring(const float* dy, const float* w_dgrad, float* dxp, float* dx, int B, int H, int W, int Cin, int Cout,
                                void* stream);
/* ---- the same layers as TWO-plane f16 split products (r05, "h2"; csrc/conv_halo_x3.hip split2h / h2_scale) -------------------------
 * s*a = hi + lo with hi = f16(s*a), lo = f16(s*a - hi), round to nearest: at least 23 significand bits in two planes, so
 * a*b needs THREE f16 MFMAs (hi*hi, hi*lo, lo*hi: one fp32 accumulator, flushed into fp32 vector sums every 16-25 k-steps) where the
 * three-plane bf16 split needs six; error against float64 2-5e-7 of the output scale, below the three-plane kernels' and the native
 * fp32 MFMA path's (profiles/r05_split2_lab.txt, profiles/r05_h2_parity_errors.txt, tests/test_h2_parity.py).  f16 has five exponent bits: each operand tensor is scaled by a power of
 * two s (exact) that puts its largest magnitude at 2^13..2^14.  The largest magnitude travels in an "absmax slot": 8 bytes of
 * caller memory holding (epoch << 32) | bits of max |a|, raised with atomic max by dwc_absmax (or by the kernel that produced the
 * tensor); slots are zero before their first use and the epochs handed to one slot never decrease, so nothing is ever cleared.
 * A consumer that finds another epoch in the slot than the one it was told poisons its result with NaN.  Non-finite operands
 * reach the result as NaN / inf.  Replaces the same reference call sites as the dwc_x3_* entry points beside them. */
int dwc_absmax(const float* x, size_t n, void* slot, unsigned epoch, void* stream);
/* Producers that raise the absmax slot of the tensor they write from their own store pass (no dwc_absmax pass for the consumer):
 * the fp32 norms' apply kernels (y / dx), the activation backward (g) and the two-plane convolutions themselves (y_amax below).
 * Arguments before `out_amax` as the plain entry points; out_amax NULL: exactly the plain entry point. */
int dwc_instnorm_fwd_amax(const float* x, const float* gamma, const float* beta, const float* residual, float* y, float* mean,
                          float* rstd, int B, int HW, int C, float eps, int relu, void* ws, size_t ws_bytes,
                          void* out_amax, unsigned out_epoch, void* stream);
int dwc_instnorm_bwd_amax(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                          const float* beta, float* dx, float* dgamma, float* dbeta, int B, int HW, int C, int relu, void* ws,
                          size_t ws_bytes, void* out_amax, unsigned out_epoch, void* stream);
int dwc_layernorm_fwd_amax(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* inv, int B, int HW,
                           int C, float eps, int relu, void* ws, size_t ws_bytes, void* out_amax, unsigned out_epoch, void* stream);
int dwc_layernorm_bwd_amax(const float* dy, const float* x, const float* mean, const float* inv, const float* gamma,
                           const float* beta, float* dx, float* dgamma, float* dbeta, int B, int HW, int C, float eps, int relu,
                           void* ws, size_t ws_bytes, void* out_amax, unsigned out_epoch, void* stream);
int dwc_act_bwd_bias_amax(const float* dy, const float* y, float* g, float* db, int rows, int C, int act, void* ws, size_t ws_bytes,
                          void* g_amax, unsigned g_epoch, void* stream);
size_t dwc_h2_weight_prepared_elems(int rows, int kdim, int K);      /* 16-bit elements incl. the {s_w, 1/s_w} tail */
int dwc_h2_weight_prepare(const float* w_oihw, void* out, int Cout, int Cin, int K, int rows, int dgrad, const void* w_amax,
                          unsigned w_epoch, void* stream);
/* (y_amax: optional absmax slot of y, raised from the store pass with epoch y_epoch -- the next convolution's x_amax; NULL: not wanted) */
int dwc_h2_conv2d_same_add_ws(const float* x, const void* x_amax, unsigned x_epoch, const void* w_prepared, const float* bias,
                              const float* add, float* y, void* y_amax, unsigned y_epoch, int B, int H, int W, int Cin, int N, int rows,
                              int K, int act, int reflect, void* ws, size_t ws_bytes, unsigned* tickets, void* stream);
/* (r06, ABI 8) The DATA GRADIENT of those layers in ONE launch: dx[B,H,W,N] = interior (zero-rule convolution of dy[B,H,W,Cout] with
 * w_prepared = dwc_h2_weight_prepare(dgrad = 1)) + the border ring of the padded gradient image folded back by the reflect rule
 * (reference networks.py:579-585 through autograd) + `add` (NULL or [B,H,W,N]).  Border tiles read pre-summed patch pixels on the rows /
 * columns the ring folds onto (csrc/conv_halo_x3.hip, RING; no extra MFMA): replaces dwc_h2_conv2d_same_add_ws(reflect = 0) +
 * dwc_conv2d_bwd_data_ring.  H, W multiples of 16 and >= 32, dy below 2 GB; scratch / tickets as dwc_h2_conv2d_same_add_ws. */
int dwc_h2_conv2d_bwd_data_same_fused(const float* dy, const void* dy_amax, unsigned dy_epoch, const void* w_prepared, const float* add,
                                      float* dx, int B, int H, int W, int Cout, int N, int rows, int K, void* ws, size_t ws_bytes,
                                      unsigned* tickets, void* stream);
int dwc_h2_conv2d_s2_ws(const float* x, const void* x_amax, unsigned x_epoch, const void* w_prepared, const float* bias, float* y,
                        void* y_amax, unsigned y_epoch, int B, int H, int W, int Cin, int N, int rows, int act, void* ws, size_t ws_bytes,
                        unsigned* tickets, void* stream);
int dwc_h2_conv2d_s2_bwd_data(const float* dy, const void* dy_amax, unsigned dy_epoch, const void* w_prepared, float* dx, int B, int H, int W,
                              int Cin, int Cout, int rows, void* stream);
/* (r06, ABI 8) the same with the border ring of the padded gradient image folded in by the launch itself (reflect-pad-1 adjoint; no
 * dwc_conv2d_bwd_data_s2_ring behind it): the whole data gradient of a 4x4 stride-2 reflect-pad-1 convolution. */
int dwc_h2_conv2d_s2_bwd_data_fused(const float* dy, const void* dy_amax, unsigned dy_epoch, const void* w_prepared, float* dx, int B, int H,
                                    int W, int Cin, int Cout, int rows, void* stream);
int dwc_h2_conv2d_wgrad(const float* x, const void* x_amax, unsigned x_epoch, const float* dy, const void* dy_amax, unsigned dy_epoch,
                        float* dw_oihw, int B, int H, int W, int Cin, int Cout, int K, int cin_real, int cout_real, void* ws, size_t ws_bytes,
                        void* stream);
/* weight gradient of the same layers as split products: dw (fp32, [cout_real][cin_real][K][K]) from the fp32 NHWC tensors x
 * and dy; both operands are split on the fly (x through LDS, dy in registers); pixel ranges go to fp32 slabs in `ws`, summed in
 * a fixed order.  ws_bytes == 0: shape not handled (K in {3,5}, H % 8 == 0, W % 16 == 0, Cin and Cout multiples of 64). */
size_t dwc_x3_conv2d_wgrad_ws_bytes(int B, int H, int W, int Cin, int Cout, int K);
int dwc_x3_conv2d_wgrad(const float* x, const float* dy, float* dw_oihw, int B, int H, int W, int Cin, int Cout, int K, int cin_real,
                        int cout_real, void* ws, size_t ws_bytes, void* stream);

/* ==== bf16-activation path (BASELINE configs[2]: "bf16 activations + MFMA im2col conv path") ==============================
 * The same call sites as above (reference networks.py:579-585, :514-515, networks_v2.py:153-156) with activations and
 * their gradients stored as bf16 (NHWC, channel counts multiples of 8 = one 16-byte chunk; the 3-channel images travel as
 * NHWC8 with planes 3..7 zero / plane 3 the attention map), v_mfma_f32_32x32x16_bf16 with fp32 accumulation, and fp32
 * everywhere a sum is carried: master weights and their gradients, biases, IN/LN statistics, gamma/beta and their
 * gradients, loss reductions, split-K / ring partial sums.  `void*` tensors are bf16; argument meaning, scratch rules and
 * return codes are those of the fp32 entry point of the same name.  Prepared weights are bf16, K padded to 64. */
size_t dwc_bf16_weight_prepared_elems(int Cout, int Cin, int KH, int KW, int stride, int cout_pad, int cin_pad, int for_dgrad);
int dwc_bf16_weight_prepare_fwd(const float* w_oihw, void* w_prepared, int Cout, int Cin, int KH, int KW, int cout_pad,
                                int cin_pad, void* stream);
int dwc_bf16_weight_prepare_dgrad(const float* w_oihw, void* w_prepared, int Cout, int Cin, int KH, int KW, int stride,
                                  int cout_pad, int cin_pad, void* stream);
size_t dwc_bf16_conv2d_fwd_ws_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int dwc_bf16_conv2d_fwd(const void* x, const void* w_prepared, const float* bias, void* y, int B, int H, int W, int Cin,
                        int Cout, int KH, int KW, int stride, int pad, int act, void* ws, size_t ws_bytes, void* stream);
int dwc_bf16_conv2d_fwd_ex(const void* x, const void* w_prepared, const float* bias, void* y, int B, int H, int W, int Cin,
                           int Cout, int KH, int KW, int stride_h, int stride_w, int pad_h, int pad_w, int act, void* stream);
size_t dwc_bf16_conv2d_bwd_data_ws_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int dwc_bf16_conv2d_bwd_data(const void* dy, const void* w_dgrad, void* dxp, int B, int H, int W, int Cin, int Cout, int KH,
                             int KW, int stride, int pad, void* ws, size_t ws_bytes, void* stream);
int dwc_bf16_reflect_pad_adjoint(const void* dxp, void* dx, int B, int H, int W, int C, int pad, void* stream);
/* dwc_bf16_conv2d_bwd_data followed by dwc_bf16_reflect_pad_adjoint, as one call: where the GEMM runs unsplit the interior of the
 * padded gradient image goes straight into dx and only its border ring into the scratch image dxp ([B,H+2pad,W+2pad,Cin] bf16),
 * which a band kernel folds onto dx -- one pass over the tensor instead of three (reference: autograd of the reflect pad in
 * Conv2dBlock.forward, networks.py:579-585, for the stride-2 layers networks.py:90,94,437 / networks_v2.py:107-111). */
int dwc_bf16_conv2d_bwd_data_fold(const void* dy, const void* w_dgrad, void* dxp, void* dx, int B, int H, int W, int Cin, int Cout,
                                  int KH, int KW, int stride, int pad, void* ws, size_t ws_bytes, void* stream);
/* the band fold alone (dx already holds the interior of the padded gradient image dxp), for producers with their own epilogue */
int dwc_bf16_reflect_pad_adjoint_band(const void* dxp, void* dx, int B, int H, int W, int C, int pad, void* stream);
size_t dwc_bf16_conv2d_bwd_data_same_ws_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW, int pad);
int dwc_bf16_conv2d_bwd_data_same(const void* dy, const void* w_dgrad, const void* w_dgrad_t, void* dx, int B, int H, int W,
                                  int Cin, int Cout, int KH, int KW, int pad, void* ws, size_t ws_bytes, void* stream);
int dwc_bf16_conv2d_bwd_data_ring(const void* dy, const void* w_dgrad, const void* w_dgrad_t, void* dx, int B, int H, int W,
                                  int Cin, int Cout, int KH, int KW, int pad, void* ws, size_t ws_bytes, void* stream);
/* bf16 forms of the frozen VGG16 trunk's operators (reference networks.py:639-688; fp32 forms: dwc_conv2d_fwd_zeropad,
 * dwc_conv2d_bwd_data_zeropad, dwc_maxpool2_*): zero-padded stride-1 convolution forward / data gradient on bf16 NHWC tensors
 * (weights from dwc_bf16_weight_prepare_fwd / _dgrad), 2x2 max pooling. */
int dwc_bf16_conv2d_fwd_zeropad(const void* x, const void* w_prepared, const float* bias, void* y, int B, int H, int W, int Cin,
                                int Cout, int KH, int KW, int stride, int pad, int act, void* ws, size_t ws_bytes, void* stream);
size_t dwc_bf16_conv2d_bwd_data_zeropad_ws_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW, int pad);
int dwc_bf16_conv2d_bwd_data_zeropad(const void* dy, const void* w_dgrad, void* dx, int B, int H, int W, int Cin, int Cout, int KH,
                                     int KW, int pad, void* ws, size_t ws_bytes, void* stream);
int dwc_bf16_maxpool2_fwd(const void* x, void* y, int B, int H, int W, int C, void* stream);
int dwc_bf16_maxpool2_bwd(const void* x, const void* dy, void* dx, int B, int H, int W, int C, void* stream);

/* Halo-tiled form of the stride-1 "same" 3x3 / 5x5 convolutions (the ResBlock and upsampling-block layers, reference
 * networks.py:514-515, networks_v2.py:153-156): a workgroup stages the (16+K-1)^2 input patch of a 16x16 pixel block once
 * per 64-channel slab and walks the K*K taps over it in LDS, instead of re-staging every pixel once per tap as the im2col
 * GEMM does.  reflect != 0: forward (w from dwc_bf16_weight_prepare_fwd); reflect == 0: interior of the data gradient
 * (x := dY, Cin := channels of dY, Cout := channels of dx, w from dwc_bf16_weight_prepare_dgrad, bias NULL, act NONE),
 * to be followed by dwc_bf16_conv2d_bwd_data_ring.  dwc_bf16_conv2d_same_halo_ok says whether a shape is handled
 * (K in {3,5}, H and W multiples of 16, Cin a power of two >= 64, Cout a multiple of 8 >= 64); no scratch. */
int dwc_bf16_conv2d_same_halo_ok(int B, int H, int W, int Cin, int Cout, int K);
int dwc_bf16_conv2d_same_halo(const void* x, const void* w_prepared, const float* bias, void* y, int B, int H, int W, int Cin,
                              int Cout, int K, int act, int reflect, void* stream);
/* ... with `add` ([B,H,W,Cout] bf16 or NULL) added behind bias and activation as one more bf16 addition: the identity-branch
 * gradient of a ResBlock (reference networks.py:521) rides on the data gradient of its first convolution. */
int dwc_bf16_conv2d_same_halo_add(const void* x, const void* w_prepared, const float* bias, const void* add, void* y, int B, int H,
                                  int W, int Cin, int Cout, int K, int act, int reflect, void* stream);
/* (r06, ABI 8) The DATA GRADIENT of those layers in ONE launch: dx[B,H,W,Cin] = interior (zero-rule convolution of dy[B,H,W,Cout]
 * with w_dgrad) + the border ring of the padded gradient image folded back by the reflect rule (reference networks.py:579-585 through
 * autograd) + `add` (NULL or [B,H,W,Cin] bf16, as above).  The border tiles compute the ring pixels that fold onto them as extra
 * MFMAs of the step whose weights they need (csrc/conv_halo16_bf16.inc, RING): replaces dwc_bf16_conv2d_same_halo_add(reflect = 0)
 * + dwc_bf16_conv2d_bwd_data_ring (strip GEMM + fold launch).  _ok: K = 3 with Cin a multiple of 128 above 128 / K = 5 with Cin a
 * multiple of 64 above 64, H, W multiples of 16 and >= 32, Cout a power of two >= 64, at least 512 four-wave workgroups. */
int dwc_bf16_conv2d_bwd_data_same_fused_ok(int B, int H, int W, int Cin, int Cout, int K);
int dwc_bf16_conv2d_bwd_data_same_fused(const void* dy, const void* w_dgrad, const void* add, void* dx, int B, int H, int W, int Cin,
                                        int Cout, int K, void* stream);

/* The stride-2 4x4 reflect-pad-1 convolutions (reference networks.py:90,94,437 -- content encoder / discriminator --,
 * networks_v2.py:107-111 -- style encoder), forward, bf16 NHWC: x [B,H,W,Cin] -> y [B,H/2,W/2,Cout], halo form over the
 * space-to-depth image (the loader does the space-to-depth).  w_prepared: dwc_bf16_weight_prepare_fwd with KH = KW = 4,
 * cout_pad = Cout, cin_pad = Cin.  dwc_bf16_conv2d_s2_halo_ok says whether a shape is handled (H, W multiples of 32, Cin a
 * power of two >= 64, Cout a multiple of 64); otherwise use dwc_bf16_conv2d_fwd. */
int dwc_bf16_conv2d_s2_halo_ok(int B, int H, int W, int Cin, int Cout);
int dwc_bf16_conv2d_s2_halo(const void* x, const void* w_prepared, const float* bias, void* y, int B, int H, int W, int Cin, int Cout,
                            int act, void* stream);
/* bf16 twins (conv_halo16_kernel, S2 == 2; w_dgrad = dwc_bf16_weight_prepare_dgrad(4x4, stride 2, cout_pad = Cout, cin_pad = Cin) for
 * both calls; dxp: bf16 scratch image [B][H+2][W+2][Cin]). */
int dwc_bf16_conv2d_s2_halo_bwd_data_ok(int B, int H, int W, int Cin, int Cout);
int dwc_bf16_conv2d_s2_halo_bwd_data(const void* dy, const void* w_dgrad, void* dx, int B, int H, int W, int Cin, int Cout, void* stream);
/* (r06, ABI 8) the same with the border ring of the padded gradient image folded in by the launch itself (reflect-pad-1 adjoint; no
 * dwc_bf16_conv2d_bwd_data_s2_ring behind it): the whole data gradient of a 4x4 stride-2 reflect-pad-1 convolution. */
int dwc_bf16_conv2d_s2_halo_bwd_data_fused(const void* dy, const void* w_dgrad, void* dx, int B, int H, int W, int Cin, int Cout, void* stream);
int dwc_bf16_conv2d_bwd_data_s2_ring(const void* dy, const void* w_dgrad, void* dxp, void* dx, int B, int H, int W, int Cin, int Cout,
                                     void* stream);
/* Halo form of the weight gradient of the same layers (reflect padding): a workgroup stages the (8+K-1)x(16+K-1) patch of x
 * and the 8x16 block of dY once and forms all 9 taps of a 3x3 / one filter row of a 5x5 from it; pixel ranges are split into
 * fp32 slabs in `ws` that a second kernel sums in a fixed order into dw ([cout_real][cin_real][K][K], fp32).  ws_bytes == 0:
 * shape not handled (K in {3,5}, H % 8 == 0, W % 16 == 0, Cin a power of two >= 64, Cout a multiple of 64) - use
 * dwc_bf16_conv2d_bwd_weight. */
size_t dwc_bf16_conv2d_wgrad_halo_ws_bytes(int B, int H, int W, int Cin, int Cout, int K);
int dwc_bf16_conv2d_wgrad_halo(const void* x, const void* dy, float* dw_oihw, int B, int H, int W, int Cin, int Cout, int K,
                               int cin_real, int cout_real, void* ws, size_t ws_bytes, void* stream);
/* 7x7 convolutions from 64 channels to the 8 planes of an NHWC8 image in the "wide" form (4 pixels x 8 planes = 32 columns,
 * bank [32][7][10][64] from dwc_bf16_weight_prepare_fwd): the fused image heads (reference networks.py:218-246; off = -3,
 * reflect rule, DWC_ACT_HEADS8) and the gradient of a 7x7 stem w.r.t. its input image on the padded grid (off = -6, zero
 * rule; dwc_bf16_conv2d_bwd_data_image_narrow = that + the reflect fold).  Patch staged once per 16x32-pixel block, taps dealt
 * to the waves, weight fragments straight from L2 in fragment order w_frag[tap][q][hi][row][8] (the [32][Kp] layout of
 * dwc_bf16_weight_prepare_fwd permuted by the caller; r04: the tap count rounded up to a multiple of 8 with ZERO taps, 72 for
 * KH x KWW = 7 x 10, so that every wave walks the same number of taps).  y: [B][OH][OWg][32] bf16.
 * r06: where the block columns of the images fill the rounds of one workgroup per CU (B = 64, 128, 192 ... at 128 x 128) the
 * launch is ONE persistent workgroup per CU (filter taps in registers, the next block's patch by LDS-DMA behind the tap loop,
 * XCD-aware walk down block columns); same results bit for bit.  DWC_ACT_HEADS8 rounds tanh / sigmoid for the bf16 result from
 * the hardware exp2 / rcp (absolute 2e-7).  DWC_NARROW_PERSIST=0 (environment, read per call) keeps the block-per-workgroup form. */
int dwc_bf16_conv2d_narrow_ok(int B, int IH, int IW, int Cin, int OH, int OWg, int KH, int KWW);
int dwc_bf16_conv2d_narrow(const void* x, const void* w_frag, const float* bias32, void* y, int B, int IH, int IW, int Cin, int OH,
                           int OWg, int KH, int KWW, int off_h, int off_w, int act, int reflect, void* stream);
int dwc_bf16_conv2d_bwd_data_image_narrow(const void* dy, const void* w_frag, void* dx, int B, int H, int W, int Cout, int KH, int KW,
                                          int pad, void* ws, size_t ws_bytes, void* stream);
/* 7x7 convolutions from the 8 planes of an NHWC8 image to 64 channels: the stems (forward; off = -3, reflect rule) and the
 * data gradient of the image heads on the padded grid (off = -6, zero rule, OH = H + 6; fold with
 * dwc_bf16_reflect_pad_adjoint).  The whole filter stays in LDS (w_steps: [25][64][16] bf16, k-step j = taps 2j, 2j+1 x 8
 * planes; element (j, co, h, p) at ((h ^ ((co>>3)&1))*8 + p)), persistent workgroups walk 16x16-pixel blocks.  act: none / relu
 * / lrelu.  y: [B][OH][OW][64] bf16. */
int dwc_bf16_conv2d_stem_ok(int B, int IH, int IW, int OH, int OW, int K, int act);
int dwc_bf16_conv2d_stem(const void* x, const void* w_steps, const float* bias, void* y, int B, int IH, int IW, int OH, int OW, int K,
                         int off, int act, int reflect, void* stream);
/* ... with the interior of the output grid diverted to `inner` ([B][OH-2crop][OW-2crop][64]): the data gradient of the image heads
 * on the padded grid leaves only its border ring in y (to be folded by dwc_bf16_reflect_pad_adjoint_band). */
int dwc_bf16_conv2d_stem_crop(const void* x, const void* w_steps, const float* bias, void* y, void* inner, int crop, int B, int IH, int IW,
                              int OH, int OW, int K, int off, int act, int reflect, void* stream);
/* weight gradient of the two 7x7 shapes between an NHWC8 image and a 64-channel tensor (pad 3, reflect): heads == 0: stems
 * (img8 = x, t64 = dY, dw [64][planes][7][7]); heads != 0: image heads (img8 = gradient of the pre-activation planes, t64 = x,
 * dw [planes][64][7][7]).  An MFMA row tile is 4 adjacent taps x 8 planes read from the pixel-major patch with the
 * transposing LDS read; pixel splits -> fp32 slabs in ws -> fixed-order sum. */
size_t dwc_bf16_conv7_smallk_wgrad_ws_bytes(int B, int H, int W, int heads);
int dwc_bf16_conv7_smallk_wgrad(const void* img8, const void* t64, float* dw, int B, int H, int W, int planes, int heads, void* ws,
                                size_t ws_bytes, void* stream);
/* gradient w.r.t. an NHWC8 image through a stem: 4 pixels x 8 planes per GEMM row, bank [p*8 + plane][co][KH][KW+3] */
size_t dwc_bf16_conv2d_bwd_data_image_ws_bytes(int B, int H, int W, int Cout, int KH, int KW, int pad);
int dwc_bf16_conv2d_bwd_data_image(const void* dy, const void* w_wide, void* dx, int B, int H, int W, int Cout, int KH, int KW,
                                   int pad, void* ws, size_t ws_bytes, void* stream);
size_t dwc_bf16_conv2d_bwd_weight_ws_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad);
int dwc_bf16_conv2d_bwd_weight(const void* x, const void* dy, float* dw_oihw, int B, int H, int W, int Cin, int Cout, int KH,
                               int KW, int stride, int pad, int cin_real, int cout_real, void* ws, size_t ws_bytes, void* stream);
size_t dwc_bf16_conv2d_bwd_weight_ex_ws_bytes(int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride_h,
                                              int stride_w, int pad_h, int pad_w);
int dwc_bf16_conv2d_bwd_weight_ex(const void* x, const void* dy, float* dw_oihw, int B, int H, int W, int Cin, int Cout, int KH,
                                  int KW, int stride_h, int stride_w, int pad_h, int pad_w, int cin_real, int cout_real,
                                  void* ws, size_t ws_bytes, void* stream);
int dwc_bf16_act_bwd_bias(const void* dy, const void* y, void* g, float* db, int rows, int C, int act, void* ws,
                          size_t ws_bytes, void* stream);
int dwc_bf16_instnorm_fwd(const void* x, const float* gamma, const float* beta, const void* residual, void* y, float* mean,
                          float* rstd, int B, int HW, int C, float eps, int relu, void* ws, size_t ws_bytes, void* stream);
int dwc_bf16_instnorm_bwd(const void* dy, const void* x, const float* mean, const float* rstd, const float* gamma,
                          const float* beta, void* dx, float* dgamma, float* dbeta, int B, int HW, int C, int relu, void* ws,
                          size_t ws_bytes, void* stream);
int dwc_bf16_layernorm_fwd(const void* x, const float* gamma, const float* beta, void* y, float* mean, float* inv, int B,
                           int HW, int C, float eps, int relu, void* ws, size_t ws_bytes, void* stream);
int dwc_bf16_layernorm_bwd(const void* dy, const void* x, const float* mean, const float* inv, const float* gamma,
                           const float* beta, void* dx, float* dgamma, float* dbeta, int B, int HW, int C, float eps, int relu,
                           void* ws, size_t ws_bytes, void* stream);
int dwc_bf16_upsample2x_fwd(const void* x, void* y, int B, int H, int W, int C, void* stream);
int dwc_bf16_upsample2x_bwd(const void* dy, void* dx, int B, int H, int W, int C, void* stream);
int dwc_bf16_avgpool2_fwd(const void* x, void* y, int B, int H, int W, int C, void* stream);
int dwc_bf16_avgpool2_bwd(const void* dy, void* dx, int B, int H, int W, int C, void* stream);
/* image boundary: NCHW fp32 (C <= 8) <-> NHWC8 bf16 */
int dwc_pack_nchw_to_nhwc8_bf16(const float* x, void* y, int B, int C, int H, int W, void* stream);
int dwc_unpack_nhwc8_bf16_to_nchw(const void* x, float* y, int B, int C, int H, int W, void* stream);
/* attention blend on NHWC8 images (plane 3 of `heads` = attention) */
int dwc_bf16_blend_fwd(const void* heads, const void* real, void* out, int npix, void* stream);
int dwc_bf16_blend_bwd(const void* dout, const void* heads, const void* real, void* dheads, int npix, void* stream);
/* mean |a-b| of bf16 tensors, fp32 result; skip4 = 8: NHWC8 images, planes 0..2 only (0: every element).
 * (the fp32 dwc_l1_mean_* accept skip4 = 0, 1 or 4 (NHWC4 images), 8 likewise) */
int dwc_bf16_l1_mean_fwd(const void* a, const void* b, float* out, size_t n, int skip4, void* ws, size_t ws_bytes, void* stream);
int dwc_bf16_l1_mean_bwd(const void* a, const void* b, const float* dout, void* da, void* db, size_t n, int skip4, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DWCGAN_HIP_H */
