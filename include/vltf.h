/* vltf.h -- C-ABI of libvltf_hip.so: the MI355X (gfx950) hot path of npit/video-learning-tf.
 *
 * The reference has no FFI: every FLOP of its hot path runs inside two TensorFlow executor calls,
 *   train:   sess.run([summaries, loss, lr, global_step, optimizer], feed_dict)   run_task.py:29,44
 *   forward: sess.run(model.logits, feed_dict)                                     run_task.py:95
 * over a graph assembled from stock TF ops (models/alexnet/alexnet.py, models/lstm/lstm.py,
 * tf_util.py, train.py).  Each entry point below replaces one of those TF ops (cited per
 * function); the Python host (video-learning-tf_amd/) composes them exactly where the
 * reference composes the TF ops, so the two sess.run calls become two host functions.
 *
 * Conventions
 *   - extern "C"; every function returns 0 on success, non-zero on failure;
 *     vl_last_error() returns a thread-local human readable message for the last failure.
 *   - Every data pointer is a DEVICE pointer owned by the caller (torch-ROCm tensors are used
 *     only as allocations).  Nothing is allocated after vl_conv_create(); workspaces are
 *     caller provided.  All work is enqueued asynchronously on `stream` (a hipStream_t).
 *   - Activations are NCHW fp32.  Parameters keep the reference's layouts: conv kernels HWIO
 *     [kh][kw][cin/group][cout] (alexnet.py:73,113), fc weights [in][out] (alexnet.py:225),
 *     LSTM kernel [D+H][4H] with gate order i, j, f, o (TF BasicLSTMCell; lstm.py:17).
 *   - Thread-compatible: one stream/descriptor set per host thread (exceptions, both process-wide words: vl_set_conv_math
 *     and the test hooks vl_lstm_seq_test_hooks, vl_pool_lrn_bwd_test_ranges).
 */
#ifndef VLTF_H
#define VLTF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vl_stream_t;                 /* hipStream_t */
typedef struct vl_conv_desc vl_conv_desc;  /* opaque convolution descriptor */

const char* vl_last_error(void);
int vl_version(void);
/* Number of HIP devices visible; <0 on error.  Does not create a context. */
int vl_device_count(void);

/* ---- input preparation: Dataset.process_image (dataset_.py:481-501) on device ------------------
 * src: uint8 [n][raw_h][raw_w][3] HWC, BGR (the TFRecord 'image_raw' bytes, dataset_.py:125-126).
 * dst: fp32  [n][3][out_h + 2*dst_halo][out_w + 2*dst_halo] NCHW; the interior ROWS are written whole -- their halo (and, in the
 * phase-split layout, padding) columns with the 0.0 a halo holds -- the halo rows above and below are not touched (they must have
 * been zeroed once: the halo is conv1's SAME padding, see vl_conv_set_halo).  n <= 65535.  crop_y/crop_x: int32[n] top-left crop offsets
 * (center: floor((raw-want)/2), dataset_.py:572-573); mirror: uint8[n] flips the W axis
 * (dataset_.py:497-499); mean_bgr: float[3] subtracted per channel (dataset_.py:521-530), may be NULL.
 * crop_y, crop_x, mirror may be NULL (= 0).  dst_phase = 1, or the consuming conv's vl_conv_x_phase(): dst is then the
 * column-phase-split layout [n][3][phase][out_h + 2*dst_halo][ceil((out_w + 2*dst_halo) / phase)]. */
int vl_input_prep_u8(const uint8_t* src, float* dst, int n, int raw_h, int raw_w, int out_h, int out_w,
                     const int32_t* crop_y, const int32_t* crop_x, const uint8_t* mirror,
                     const float* mean_bgr, int dst_halo, int dst_phase, vl_stream_t stream);
/* The reference's own feed format: fp32 NHWC placeholder (models/model.py:54) -> NCHW (+halo, dst_phase as above). */
int vl_nhwc_to_nchw(const float* src, float* dst, int n, int h, int w, int c, int dst_halo, int dst_phase, vl_stream_t stream);
int vl_nchw_to_nhwc(const float* src, float* dst, int n, int c, int h, int w, vl_stream_t stream);

/* ---- convolution: dcnn.conv (alexnet.py:15-31) = tf.nn.conv2d 'SAME' per group + bias_add ------
 * The descriptor fixes geometry and owns small device index tables (allocated at create). */
int vl_conv_create(vl_conv_desc** out, int cin, int h, int w, int cout, int kh, int kw, int stride, int groups);
void vl_conv_destroy(vl_conv_desc* d);
int vl_conv_out_hw(const vl_conv_desc* d, int* oh, int* ow);
/* Padded ("halo") activation layout.  A tensor with halo p is stored [n][c][H + 2p][W + 2p] with p zero
 * pixels on every side of each plane; only interiors are ever written.  When the gathered operand's halo
 * covers the SAME padding, the im2col gather needs no bounds test at all (every tap is in-bounds and the
 * padding taps read the zeros), which removes all vector-ALU address work from the kernel's steady state.
 *   x_halo : layout of x (vl_conv_fwd / vl_conv_wgrad input; also of vl_conv_dgrad's relu_mask)
 *   y_halo : layout of y written by vl_conv_fwd
 *   dy_halo: layout of dy read by vl_conv_dgrad / vl_conv_wgrad
 *   dx_halo: layout of dx written by vl_conv_dgrad
 * Default 0 everywhere (dense NCHW; bounds-tested gather).  Rebuilds the index tables (setup time only). */
int vl_conv_set_halo(vl_conv_desc* d, int x_halo, int y_halo, int dy_halo, int dx_halo);
/* Column-phase-split x for a strided conv in the padded layout (conv1, stride 4): x is stored
 * [n][c][phase = stride][H + 2p][ceil((W + 2p) / stride)], physical column iw at [iw % stride][iw / stride].  The taps of
 * consecutive output columns are then CONSECUTIVE addresses (a 64-lane gather touches 2 cache lines instead of 8), for
 * vl_conv_fwd and vl_conv_wgrad alike.  on = 0 restores plain [n][c][H + 2p][W + 2p].  Call after vl_conv_set_halo; no-op
 * for stride 1.  vl_conv_x_phase returns the phase count in force (1 = plain). */
int vl_conv_set_x_phase_split(vl_conv_desc* d, int on);
int vl_conv_x_phase(const vl_conv_desc* d);
/* Arithmetic of the contraction in vl_conv_fwd / vl_conv_dgrad / vl_conv_wgrad and, given the workspace of
 * vl_gemm_split_ws_bytes, of the large vl_gemm products (process-wide):
 *   0  fp32 MFMA (default; the reference's arithmetic, alexnet.py:21 tf.nn.conv2d on float32)
 *   3  "bf16x3": each fp32 operand is split into a bf16 head and tail (x = hi + lo + O(2^-17 |x|)) and a product is
 *      hi*hi + hi*lo + lo*hi on the bf16 matrix pipe with fp32 accumulation -- results agree with mode 0 to ~5e-6
 *      relative L2 per layer, inside every parity tolerance of tests/, but it is NOT fp32 arithmetic and is opt-in.
 *      Applies to layers in the padded layout with >= 96 output channels per group and unit column stride in memory
 *      (stride 1, or the phase-split x of a strided conv); other layers keep mode 0.
 *   6  "bf16x6": three bf16 pieces per operand and the six products above 2^-23 -- agrees with mode 0 to ~1e-7 relative L2 per
 *      layer, i.e. to fp32 rounding (as close as two fp32 summation orders are to each other); opt-in like mode 3.
 *   1  plain bf16 products (heads only), fp32 accumulation: ~2.3e-3 relative L2 per layer -- the reduced-precision conv path
 *      of BASELINE config 5; outside the fp32 parity tolerances by design (tests hold it to 3e-2 on logits).
 * The environment variable VL_CONV_MATH=bf16x3 | bf16x6 | bf16 presets mode 3 | 6 | 1.
 * NOT thread-safe: the mode is ONE process-wide word read by every later launch call, the only exception to this header's
 * "thread-compatible" convention.  A process that mixes arithmetics sets it before each group of launches from the one thread
 * that issues them (LRCNEngine does, at the top of every forward and backward pass); two host threads launching with
 * different modes need an external lock around (set, launches). */
int vl_set_conv_math(int math);
int vl_conv_math(void);
/* y[n][cout][oh][ow] = conv(x[n][cin][h][w], w_hwio) + bias, optional fused ReLU (alexnet.py:77). */
int vl_conv_fwd(const vl_conv_desc* d, const float* x, const float* w_hwio, const float* bias, float* y,
                int n, int relu, vl_stream_t stream);
/* wt = flipped/transposed copy of w used by vl_conv_dgrad: wt[kh'][kw'][co][g*cin_g+ci] =
 * w[KH-1-kh'][KW-1-kw'][ci][g*cout_g+co].  Same element count as w. */
int vl_conv_wt_transpose(const vl_conv_desc* d, const float* w_hwio, float* wt, vl_stream_t stream);
/* dx = d(loss)/dx given dy (stride-1 layers only; conv1 never needs it).  If relu_mask != NULL
 * (same shape as dx) the ReluGrad of the producing layer is fused: dx = relu_mask > 0 ? dx : 0. */
int vl_conv_dgrad(const vl_conv_desc* d, const float* dy, const float* wt, float* dx, const float* relu_mask,
                  int n, vl_stream_t stream);
/* dw (HWIO) = d(loss)/dw.  Deterministic split reduction through `ws` (>= vl_conv_wgrad_ws_bytes).
 * db (optional, [cout]): the bias gradient sum_{n,h,w} dy, accumulated from the dy tiles the kernel streams
 * anyway (no second pass over dy); available when vl_conv_wgrad_fuses_bias() != 0 (padded layout), else pass
 * NULL and use vl_bias_grad_nchw. */
size_t vl_conv_wgrad_ws_bytes(const vl_conv_desc* d, int n);
int vl_conv_wgrad_fuses_bias(const vl_conv_desc* d);
int vl_conv_wgrad(const vl_conv_desc* d, const float* x, const float* dy, float* dw, float* db, void* ws,
                  size_t ws_bytes, int n, vl_stream_t stream);
/* ---- the bf16 conv PATH (BASELINE config 5; csrc/conv_c8.hip) --------------------------------------
 * Same arithmetic as vl_set_conv_math(1) -- operands rounded to bf16 (nearest even), fp32 accumulation, fp32 bias / ReLU -- but the
 * operands ARE bf16 in memory, in the "c8" layout: a tensor [n][c][h][w] with halo p is stored
 * [n][ceil(c/8)][h + 2p][w + 2p][8] bf16 (8 consecutive channels of a pixel = one 16-byte chunk = one MFMA operand; zero halo, zero
 * beyond c).  Kernels fetch 16 bytes per lane straight into LDS and run v_mfma_f32_32x32x16_bf16 with no conversion in the loop.
 * Layers: channels per group a multiple of 8, padded layout (vl_conv_set_halo), not phase split; dgrad / wgrad: stride 1, wgrad
 * additionally x_halo == dy_halo (every SAME layer with an odd kernel).  Halos are those of the descriptor, as for the fp32 calls.
 * PRECONDITION on every c8 tensor handed to these calls: halo chunks and channels beyond c hold ZEROS.  No kernel here writes them
 * (only interiors are ever written), so a buffer zeroed once at allocation stays valid; a caller that recycles memory must zero it
 * first.  vl_conv_c8_wgrad (it sweeps each image's plane linearly, halo columns included: their dy must be 0) and vl_bias_grad_c8
 * (it sums whole rows) return WRONG dw / db on a dirty halo, without any check; forward / dgrad read halo taps as the padding zeros.
 *   vl_c8_bytes            bytes of a c8 tensor (allocate zeroed once: only interiors are ever written)
 *   vl_pack_c8             fp32 NCHW (x_halo) -> c8 (xb_halo): the stand-alone producer
 *   vl_conv_c8_pack_w      HWIO fp32 weights -> the packed operand of vl_conv_c8_fwd (bwd = 0) / vl_conv_c8_dgrad (bwd = 1);
 *                          wb: vl_conv_c8_w_bytes(d, bwd) bytes
 *   vl_conv_c8_fwd         y (fp32 NCHW, y_halo; may be NULL) and / or yb (c8, y_halo; may be NULL) = conv(xb) + bias (ReLU)
 *   vl_conv_c8_dgrad       dx (fp32 NCHW, dx_halo) and / or dxb (c8, dx_halo) from dyb (c8, dy_halo); ReluGrad of the producing layer
 *                          from relu_mask (fp32, as vl_conv_dgrad) or relu_mask_c8 (the layer's own packed input xb, x_halo)
 *   vl_bias_grad_c8        db[c] = sum_{n,h,w} dy from the packed gradient (c8, halo); ws: float[64 * 8 * ceil(c/8)]
 *   vl_conv_c8_wgrad       dw (HWIO fp32) from xb and dyb; deterministic slab reduction through ws (vl_conv_c8_wgrad_ws_bytes) */
size_t vl_c8_bytes(int n, int c, int h, int w, int halo);
int vl_pack_c8(const float* x, void* xb, int n, int c, int h, int w, int x_halo, int xb_halo, vl_stream_t stream);
size_t vl_conv_c8_w_bytes(const vl_conv_desc* d, int bwd);
int vl_conv_c8_pack_w(const vl_conv_desc* d, const float* w_hwio, void* wb, int bwd, vl_stream_t stream);
int vl_conv_c8_fwd(vl_conv_desc* d, const void* xb, const void* wb, const float* bias, float* y, void* yb, int n, int relu,
                   vl_stream_t stream);
int vl_conv_c8_dgrad(vl_conv_desc* d, const void* dyb, const void* wbt, float* dx, void* dxb, const float* relu_mask,
                     const void* relu_mask_c8, int n, vl_stream_t stream);
size_t vl_conv_c8_wgrad_ws_bytes(const vl_conv_desc* d, int n);
int vl_conv_c8_wgrad(vl_conv_desc* d, const void* xb, const void* dyb, float* dw, void* ws, size_t ws_bytes, int n,
                     vl_stream_t stream);
int vl_bias_grad_c8(const void* dyb, float* db, float* ws, int n, int c, int h, int w, int halo, vl_stream_t stream);
/* Dense products on the same pipeline (fc6 of the bf16 path): c[m][n] = sum_k a[k][m] b[k][n] (+ bias[n]) (ReLU) with both operands
 * reduction-major in 8-channel blocks, "kc8" = [ceil(channels / 8)][k][8] bf16 -- what the wgrad kernel consumes.
 *   vl_pack_kc8   dst = src[position * pos_stride + channel * ch_stride] (fp32, any strides: a matrix or its transpose) in kc8
 *   vl_gemm_kc8   the product (fp32 out, row-major; m, n multiples of 8); ws: vl_gemm_kc8_ws_bytes (split-k slabs, fixed order) */
int vl_pack_kc8(const float* src, void* dst, int64_t positions, int channels, int64_t pos_stride, int64_t ch_stride, vl_stream_t stream);
size_t vl_gemm_kc8_ws_bytes(int m, int n, int k);
int vl_gemm_kc8(const void* a_kc8, const void* b_kc8, float* c, int m, int n, int k, const float* bias, int relu, void* ws,
                size_t ws_bytes, vl_stream_t stream);
/* A strided first layer (conv1: 11 x 11 / 4 over 3 channels) on the same kernels: with kh = s a + py, kw = s b + px it is a ka x ka
 * (ka = ceil(k / s), odd) STRIDE-1 layer over the cin s^2 channels (c, py, px) of the space-to-depth input -- identical products and
 * sums, plus multiplies by zero where s a + py >= k.  The caller creates that layer's descriptor (cin s^2, oh, ow, cout, ka, ka, 1, 1;
 * x_halo = dy_halo = (ka - 1) / 2) and runs vl_conv_c8_fwd / vl_conv_c8_wgrad on it with
 *   vl_s2d_c8_from_x0   x0 (fp32, as the strided layer d's vl_conv_fwd takes it) -> packed input [n][cin s^2 / 8][oh + ka - 1][ow + ka - 1][8]
 *   vl_s2d_weights      grad = 0: w [k][k][cin][cout] -> [ka][ka][cin s^2][cout];  grad = 1: the stride-1 layer's dw -> dw
 *   vl_input_prep_u8_s2d  the uint8 frames straight into that packed input (arguments as vl_input_prep_u8; the crop is d's h x w):
 *                       the same values as vl_input_prep_u8 followed by vl_s2d_c8_from_x0 */
int vl_s2d_c8_from_x0(const vl_conv_desc* d, const float* x0, void* xb, int n, vl_stream_t stream);
int vl_input_prep_u8_s2d(const vl_conv_desc* d, const uint8_t* src, void* xb, int n, int raw_h, int raw_w, const int32_t* crop_y,
                         const int32_t* crop_x, const uint8_t* mirror, const float* mean_bgr, vl_stream_t stream);
int vl_s2d_weights(const vl_conv_desc* d, const float* src, float* dst, int grad, vl_stream_t stream);
/* db[c] = sum_{n,h,w} dy[n][c][h][w]  (gradient of tf.nn.bias_add, alexnet.py:31).
 * ws: float[64*c] scratch. */
int vl_bias_grad_nchw(const float* dy, float* db, float* ws, int n, int c, int hw, vl_stream_t stream);

/* ---- tf.nn.local_response_normalization (alexnet.py:79-89,120-130), across channels ------------
 * y = x / (bias + alpha * sum_{|c'-c|<=radius} x^2)^beta   (alpha not divided by the window). */
int vl_lrn_fwd(const float* x, float* y, int n, int c, int hw, int radius, float alpha, float beta, float bias,
               vl_stream_t stream);
/* dx for the above; relu_fused != 0 additionally applies the ReluGrad of the layer that produced
 * x (x is a ReLU output, alexnet.py:77): dx = x > 0 ? dx : 0.  x and dy are dense; dx may carry a halo
 * (plane width w, hw = h*w): it is the dy of the conv that produced x. */
int vl_lrn_bwd(const float* x, const float* dy, float* dx, int n, int c, int hw, int radius, float alpha,
               float beta, float bias, int relu_fused, int w, int dx_halo, vl_stream_t stream);

/* Fused backward of [LRN -> max_pool 3x3/2 VALID] (alexnet.py:79-98,120-139): dx = LRN'(x) applied to the
 * pooled gradient routed through argmax, + optional ReluGrad of x; the gradient wrt the LRN output is never
 * written.  x dense NCHW [n][c][h][w] (the LRN input); dp / argmax: pool-output layout NCHW with p_halo;
 * dx: NCHW with dx_halo; its interior ROWS are written whole (the halo columns of those rows receive the 0.0 a zero halo holds). */
int vl_pool_lrn_bwd(const float* x, const float* dp, const uint8_t* argmax, float* dx, int n, int c, int h, int w,
                    int p_halo, int radius, float alpha, float beta, float bias, int relu_fused, int dx_halo,
                    vl_stream_t stream);

/* vl_pool_lrn_bwd with the gradient written as packed bf16 (dxb: "c8" layout of the bf16 conv path, dxb_halo; rounded to nearest
 * even) instead of fp32 dx -- the only form that path's wgrad / dgrad / bias gradient read.  x_packed != 0: x (the LRN input) is packed
 * as well, c8 without a halo, as the producing conv's epilogue writes it. */
int vl_pool_lrn_bwd_c8(const void* x, int x_packed, const float* dp, const uint8_t* argmax, void* dxb, int n, int c, int h, int w,
                       int p_halo, int radius, float alpha, float beta, float bias, int relu_fused, int dxb_halo, vl_stream_t stream);

/* Test hook of vl_pool_lrn_bwd / vl_pool_lrn_bwd_c8, process-wide: force the number of channel ranges a (band, image) is split into
 * (1..4; fewer where the channel count does not allow it); 0 = the launcher's cost model (the default). */
int vl_pool_lrn_bwd_test_ranges(int ranges);

/* Fused forward of [LRN -> max_pool 3x3/2 VALID] (alexnet.py:79-98,120-139): p = max_pool(lrn(x)), argmax = window-local
 * index (0..8) of the first maximum in scan order; the LRN output is never written (vl_pool_lrn_bwd needs only x).
 * x dense NCHW [n][c][h][w]; p / argmax: NCHW with p_halo.  The interior ROWS are written whole: the halo columns of those rows
 * receive 0.0 in p (what a zero halo holds anyway) and unspecified bytes in argmax (its halo is never read); halo rows are not touched. */
int vl_lrn_pool_fwd(const float* x, float* p, uint8_t* argmax, int n, int c, int h, int w, int p_halo, int radius,
                    float alpha, float beta, float bias, vl_stream_t stream);

/* vl_lrn_pool_fwd with the pooled output written as packed bf16 (pb: "c8" layout of the bf16 conv path, p_halo) instead of fp32 p:
 * the next conv's operand.  argmax as vl_lrn_pool_fwd.  x_packed != 0: x is packed as well (c8 without a halo). */
int vl_lrn_pool_fwd_c8(const void* x, int x_packed, void* pb, uint8_t* argmax, int n, int c, int h, int w, int p_halo, int radius,
                       float alpha, float beta, float bias, vl_stream_t stream);

/* ---- tf.nn.max_pool k x k, stride s, VALID (alexnet.py:91-98,132-139,204-211) ------------------
 * x NCHW [n][c][h][w]; y element (n,c,oh,ow) is stored at y[n*ys_n + c*ys_c + oh*ys_h + ow*ys_w]
 * (NCHW: ys = {c*oh*ow, oh*ow, ow, 1}; (h,w,c)-flat for fc6, alexnet.py:228: {oh*ow*c, 1, ow*c, c}).
 * argmax: uint8 per output element, stored with the same strides: window-local index of the
 * first maximum in scan order (TF-CPU MaxPoolGrad target). */
int vl_maxpool_fwd(const float* x, float* y, uint8_t* argmax, int n, int c, int h, int w, int k, int s,
                   int64_t ys_n, int64_t ys_c, int64_t ys_h, int64_t ys_w, vl_stream_t stream);
/* dx NCHW with dx_halo (interior fully written).  relu_mask (the pool input, dense NCHW) optional: fused ReluGrad. */
int vl_maxpool_bwd(const float* dy, const uint8_t* argmax, float* dx, const float* relu_mask, int n, int c,
                   int h, int w, int k, int s, int64_t ys_n, int64_t ys_c, int64_t ys_h, int64_t ys_w,
                   int dx_halo, vl_stream_t stream);

/* ---- dense GEMM on fp32 MFMA: tf.nn.relu_layer / xw_plus_b / matmul gradients ------------------
 * C[m][n] = sum_k opA(m,k) * opB(k,n) (+ bias[n]) (ReLU) ; then C = relu_mask>0 ? C : 0 if given.
 * transa == 0: A is [m][k] row-major (lda);  transa != 0: A is stored [k][m] (lda).
 * transb == 0: B is [k][n] row-major (ldb);  transb != 0: B is stored [n][k] (ldb).
 * relu_mask has C's layout (ldc).  ws/ws_bytes: scratch for split-K (may be NULL/0: no split). */
int vl_gemm(int transa, int transb, int m, int n, int k, const float* a, int64_t lda, const float* b, int64_t ldb,
            float* c, int64_t ldc, const float* bias, int relu, const float* relu_mask, void* ws, size_t ws_bytes,
            vl_stream_t stream);
/* Workspace (bytes) with which vl_gemm runs an m x n x k product in the split-bf16 arithmetic selected by vl_set_conv_math
 * (operand images + split-K slabs); with a smaller workspace, or in mode 0, vl_gemm is the fp32 MFMA kernel. */
size_t vl_gemm_split_ws_bytes(int m, int n, int k);
/* out[n] = sum_m a[m][n] (bias gradients of fc layers); ws: float[64*n]. */
int vl_colsum(const float* a, int64_t lda, float* out, float* ws, int m, int n, vl_stream_t stream);

/* ---- LSTM: tf.contrib.rnn.BasicLSTMCell under tf.nn.dynamic_rnn (lstm.py:9-20,102-143) ---------
 * Rows are clip-major: row r = b*T + t.  gx = X @ kernel[:D] + bias for all rows is hoisted into one
 * vl_gemm; per step the host calls vl_gemm for gh = h_{t-1} @ kernel[D:] and then this kernel:
 *   z = gx[r] + gh[b];  i,j,f,o = split(z);  c = c_prev*sigmoid(f+forget_bias) + sigmoid(i)*tanh(j);
 *   h = tanh(c)*sigmoid(o).
 * act[r][4H] receives the activated gates (i, j, f, o), cseq[r][H] the cell state, hseq[r][H] the
 * output, hprev[r][H] a copy of h_{t-1} (zeros at t == 0).  gh may be NULL at t == 0. */
int vl_lstm_step_fwd(const float* gx, const float* gh, float* act, float* cseq, float* hseq, float* hprev,
                     int batch, int T, int t, int H, float forget_bias, vl_stream_t stream);
/* BPTT step t: dh = dout[r] + dh_next[b] (dh_next may be NULL at t == T-1); writes dz[r][4H] and
 * dc (in/out, [batch][H], zero before t == T-1). */
int vl_lstm_step_bwd(const float* dout, const float* dh_next, const float* act, const float* cseq, float* dc,
                     float* dz, int batch, int T, int t, int H, vl_stream_t stream);

/* The whole recurrence in ONE launch per direction (csrc/lstm_cluster.hip).  kh = kernel[D:] ([H][4H], row stride 4H).
 * Same outputs as T x {vl_gemm + vl_lstm_step_fwd}.  H <= 1024.
 *   h0, c0 (nullable, [batch][H]): initial output and cell state -- the reference's get_state_tuple (lstm.py:34-42) passes ONE
 *   vector as both (c = h = init), from input_state_fc (lstm.py:74-77); NULL = dynamic_rnn's zero state.  hprev[r] at t == 0 is h0.
 * H <= 512: weight-stationary cluster form -- ceil(H/16) workgroups per group of <= 8 clips keep their 64 gate columns of kh in
 * LDS for the whole sequence and exchange h_t (forward) / partial dh_{t-1} (backward) through tagged 8-byte words in `ws`
 * (agent-scope atomics, bounded spins); larger H: one workgroup per clip streaming kh every step.
 * ws: device scratch of vl_lstm_seq_ws_bytes(batch, T, H) bytes, ZEROED ONCE by the caller when it is allocated: its first word is
 * the sticky time-out flag of vl_lstm_seq_status, which no launch clears; the rest holds the exchange words, whose tags are unique
 * per launch within the process (no launch zeroes them), so it must not be handed anything else to scribble on. */
size_t vl_lstm_seq_ws_bytes(int batch, int T, int H);
int vl_lstm_seq_fwd(const float* gx, const float* kh, const float* h0, const float* c0, float* act, float* cseq, float* hseq,
                    float* hprev, int batch, int T, int H, float forget_bias, void* ws, size_t ws_bytes, vl_stream_t stream);
/* BPTT over all steps: dout may be NULL; writes dz[r][4H]; c0 as given to the forward call (NULL = zero state);
 * dh0 / dc0 (nullable, [batch][H]) receive the gradients w.r.t. the initial output / cell state. */
int vl_lstm_seq_bwd(const float* dout, const float* kh, const float* act, const float* cseq, const float* c0, float* dz,
                    float* dh0, float* dc0, int batch, int T, int H, void* ws, size_t ws_bytes, vl_stream_t stream);
/* Synchronous read-and-reset of the time-out flag in `ws`: *timed_out = 1 if a workgroup of ANY cluster-form launch on `ws` since
 * the last call of this function gave up waiting for its peers (the cluster form needs all its workgroups resident at once: one
 * per CU; a launch that shares the device with another kernel holding CUs can starve).  Results are then invalid -- it never
 * hangs -- and the caller must discard the step (LRCNEngine / GraphEngine raise). */
int vl_lstm_seq_status(void* ws, int* timed_out);
/* Test hooks of the cluster form, process-wide: spin_limit polls before a gather gives up (0 = the default 2^18);
 * mute_workgroup >= 0: that workgroup of every launch publishes nothing, so its peers time out (-1 = off). */
int vl_lstm_seq_test_hooks(unsigned spin_limit, int mute_workgroup);
/* dst[cols][rows] = src[rows][cols]^T (src row stride ld). */
int vl_transpose(const float* src, int64_t ld, float* dst, int rows, int cols, vl_stream_t stream);

/* ---- apply_temporal_fusion (tf_util.py:4-30) over x[batch][T][H] ---------------------------------
 * method 0 = avg, 1 = last. */
int vl_temporal_fusion_fwd(const float* x, float* y, int batch, int T, int H, int method, vl_stream_t stream);
int vl_temporal_fusion_bwd(const float* dy, float* dx, int batch, int T, int H, int method, vl_stream_t stream);

/* ---- imresize: scipy.misc.imresize(image, shape) of Dataset.process_image (dataset_.py:481-495: imgproc `raw_resize` to the raw
 * shape, `resize` to the network input size; also serialize.py:424-425) = PIL Image.resize(BILINEAR) on uint8, bit-exact:
 * Pillow's two-pass fixed-point resample (22-bit coefficients, uint8 intermediate, horizontal pass first; csrc/resize.hip).
 * The descriptor holds the coefficient tables of one (h, w) -> (oh, ow) pair on the device.  Images are uint8 [n][h][w][3] (HWC). */
typedef struct vl_resize_desc vl_resize_desc;
int vl_resize_create(vl_resize_desc** out, int h, int w, int oh, int ow, int channels);
void vl_resize_destroy(vl_resize_desc* d);
/* bytes of the uint8 intermediate vl_resize_u8 needs for n images (0 when at most one axis changes size) */
size_t vl_resize_tmp_bytes(const vl_resize_desc* d, int n);
int vl_resize_u8(const vl_resize_desc* d, const uint8_t* src, uint8_t* tmp, uint8_t* dst, int n, vl_stream_t stream);

/* ---- tensor-list plumbing of multi-input pipelines (tf_util.py:99-192) ------------------------------------------------
 * vl_copy2d: dst[r][c] = src[r][c] for r < rows, c < cols with row strides src_ld / dst_ld (src_ld 0 repeats one row).
 * tf.concat / vec_seq_concat (tf_util.py:99-124), the ibias insertion (tf_util.py:154-176) and replicate_auxilliary_tensor
 * (tf_util.py:182-192) are such block copies, forward and backward. */
int vl_copy2d(const float* src, int64_t src_ld, float* dst, int64_t dst_ld, int rows, int cols, vl_stream_t stream);
/* op 0: a + b; 1: mean of the two (tf.reduce_mean, fusion avg, tf_util.py:142-143); 2: max (fusion maximum, :144-145). */
int vl_eltwise2(const float* a, const float* b, float* out, int64_t count, int op, vl_stream_t stream);
/* gradient of max(a, b) w.r.t. both: the larger input takes d, equal inputs share it evenly (tf.reduce_max's _MinOrMaxGrad). */
int vl_max2_grad(const float* a, const float* b, const float* d, float* da, float* db, int64_t count, vl_stream_t stream);
/* apply_tensor_list_fusion avg | maximum over a LIST of n <= 8 equally shaped tensors (tf.reduce_mean / tf.reduce_max over the
 * stacked list, tf_util.py:142-145).  ins / dins: HOST arrays of n device pointers.  op 0: mean, 1: maximum.
 * vl_fuse_n_grad writes dins[i] (null entries are skipped): d / n for the mean; for the maximum the inputs equal to it share d
 * evenly (ins may be null for op 0). */
int vl_fuse_n(const float* const* ins, int n, float* out, int64_t count, int op, vl_stream_t stream);
int vl_fuse_n_grad(const float* const* ins, int n, const float* d, float* const* dins, int64_t count, int op, vl_stream_t stream);

/* ---- tf.nn.dropout (lstm.py:50-56): y = x * mask / keep, mask ~ Bernoulli(keep) ------------------
 * Counter-based RNG keyed by (seed, element index); mask (uint8) is written for the backward. */
int vl_dropout_fwd(const float* x, float* y, uint8_t* mask, int64_t count, float keep, uint64_t seed, vl_stream_t stream);
int vl_dropout_bwd(const float* dy, const uint8_t* mask, float* dx, int64_t count, float keep, vl_stream_t stream);

/* ---- loss: mean_b softmax_cross_entropy_with_logits (train.py:120-123) + accuracy (142-149) ----
 * labels: int32 one/multi-hot [batch][classes] (the reference's labels placeholder, train.py:117).
 * dlogits = (softmax - labels) * grad_scale   (grad_scale = 1/global_batch); may be NULL.
 * stats[0] += sum_b loss_b, stats[1] += number of rows with argmax(logits) == argmax(labels);
 * zero `stats` first (vl_fill).  rows: float[2*batch] workspace receiving the per-row losses and hits (one wave per row over
 * the whole chip, summed in a fixed order); NULL walks every row in one workgroup (same result, for small batches only). */
int vl_softmax_xent(const float* logits, const int32_t* labels, float* dlogits, float* stats, float* rows,
                    int batch, int classes, float grad_scale, vl_stream_t stream);

/* ---- optimizer: clip_by_global_norm + GradientDescentOptimizer (train.py:199-222) ---------------
 * vl_sumsq: out[0] (+)= sum g^2 over count elements (ws: float[1024]); accumulate != 0 adds to out. */
int vl_sumsq(const float* g, int64_t count, float* out, float* ws, int accumulate, vl_stream_t stream);
/* w -= lr * gscale * clip_scale * g with clip_scale = clip_norm / max(gscale*sqrt(*sumsq), clip_norm)
 * (1 if clip_norm <= 0 or sumsq == NULL).  gscale folds the 1/world averaging of DP all-reduce.
 * skip (device word, may be NULL): when *skip != 0 at execution time the launch changes nothing -- the step's gradients are invalid
 * (a cluster-form LSTM launch of the step timed out, vl_status_or) and must not reach the weights; the host raises when it next reads
 * the status (vl_lstm_seq_status). */
int vl_sgd_apply(float* w, const float* g, int64_t count, float lr, float clip_norm, const float* sumsq,
                 float gscale, const uint32_t* skip, vl_stream_t stream);
/* tf.train.AdamOptimizer (train.py:205-206) with TF defaults beta1=.9 beta2=.999 eps=1e-8; step >= 1.  skip: as vl_sgd_apply (m, v
 * stay untouched too). */
int vl_adam_apply(float* w, const float* g, float* m, float* v, int64_t count, float lr, float clip_norm,
                  const float* sumsq, float gscale, int step, const uint32_t* skip, vl_stream_t stream);
/* *dst = (init ? 0 : *dst) | (the sticky time-out word of an LSTM cluster workspace != 0), on the stream: collects the `skip` word of a
 * step without a host round trip (init != 0 for the step's first workspace; the workspace's own word stays set until
 * vl_lstm_seq_status reads it). */
int vl_status_or(uint32_t* dst, const void* lstm_ws, int init, vl_stream_t stream);

/* ---- utilities ------------------------------------------------------------------------------- */
int vl_fill(float* p, int64_t count, float value, vl_stream_t stream);
/* ReluGrad in place: d[i] = y[i] > 0 ? d[i] : 0 (y = the ReLU's forward output, alexnet.py:228,248). */
int vl_relu_grad(float* d, const float* y, int64_t count, vl_stream_t stream);
/* truncated-normal / uniform parameter initialisers are host side; nothing here. */

#ifdef __cplusplus
}
#endif
#endif /* VLTF_H */
