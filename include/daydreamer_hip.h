/* C ABI of libdaydreamer_hip.so: the MI355X (gfx950) kernels behind the
 * DreamerV2+ learner step.
 *
 * The reference (danijar/daydreamer) has no FFI on this path: its learner is
 * Python (embodied/agents/dreamerv2plus/{agent,nets,tfutils}.py) whose
 * arithmetic is TensorFlow/XLA ops.  Each entry point below replaces the TF
 * op(s) cited next to it; a maintainer binds them with ctypes (see
 * INTEGRATION.md).  Paths are relative to embodied/agents/dreamerv2plus/.
 *
 * Conventions: every function returns 0 on success or a non-zero HIP / -1
 * argument error (text via dd_last_error()); all pointers are DEVICE pointers
 * to fp32 unless noted; matrices are row-major with an explicit leading
 * dimension `ld*` (in elements) so column slices of wider buffers can be
 * passed without copies; the caller owns all memory including the `ws`
 * scratch workspace; nothing synchronises; `stream` is a hipStream_t (may be
 * a capturing stream).
 */
#ifndef DAYDREAMER_HIP_H_
#define DAYDREAMER_HIP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI version.  1: rounds 1-2.  2 (round 3): dd_grad_norm gained `int mixed` in front of
 * `stream` and its opt_state grew from 3 to 5 doubles - a caller built against version 1 must
 * not call it.  3 (round 4): the version was bumped for that change.  4: dd_ln_act_bwd gained
 * `beta_ln` after `gamma` (out may then be NULL); dd_gemm_set_ws added.  5: dd_symexp added.
 * 6: dd_adam_step gained `float warmup` in front of `stream`; the `sync2` buffer of
 * dd_observe_scan_fwd / _bwd grew from 2 to 1088 words (the launches clear 512 row-block counters
 * at word 576: a caller with the old 2-word buffer gets an out-of-bounds device write).
 * 7 (round 5): dd_video_grid added.  8: dd_reduce_stats_multi added.
 * 9: dd_imagine_rollout_oh_fwd added; dd_imagine_rollout_supported answers discrete = 1 shapes.
 * 10: dd_ln_act_fwd_head / dd_ln_act_bwd_head added.
 * 11 (round 6): dd_gemm_f32_x, dd_imag_set_rows, dd_conv2d_s2_wgrad_ln,
 * dd_conv2d_s2_down_ln, dd_conv2d_s2_down_lnbwd, dd_stamp added. */
#define DD_ABI_VERSION 11
int dd_version(void);
const char* dd_last_error(void);

/* ---- dense / conv contractions (MFMA fp32) -------------------------------- */

/* C[M,N] = alpha * op(A) @ op(B) + beta * C + bias[N]   (bias may be NULL)
 * op(A) is A[M,K] (transA=0) or A stored [K,M] (transA=1); likewise B [K,N] /
 * stored [N,K].  Replaces `x @ kernel (+ bias)` nets.py:573-579 and its
 * gradients taken by GradientTape, tfutils.py:214.  ws: split-K scratch. */
int dd_gemm_f32(const float* A, const float* B, float* C, int M, int N, int K,
                long lda, long ldb, long ldc, int transA, int transB,
                float alpha, float beta, const float* bias,
                float* ws, size_t ws_bytes, int* deferred, void* stream);
/* dd_gemm_f32 with a hint about one operand: columns [xa0, xa1) of the STORED A matrix (or
 * [xb0, xb1) of the stored B matrix; pass an empty range for the other) hold values that are
 * exact in bfloat16 - the one-hot `stoch` columns of the feature matrix, nets.py:88-97,
 * tfutils.py:368-382.  Those elements have all-zero middle / low planes in the three-way
 * split, so the three plane products that involve them are left out: the result is
 * bit-identical to dd_gemm_f32 at half the matrix instructions over that range.  A hint the
 * variant does not cover (shape, alignment, arithmetic mode) is ignored. */
int dd_gemm_f32_x(const float* A, const float* B, float* C, int M, int N, int K,
                  long lda, long ldb, long ldc, int transA, int transB,
                  float alpha, float beta, const float* bias,
                  float* ws, size_t ws_bytes, int* deferred,
                  int xa0, int xa1, int xb0, int xb1, void* stream);
/* Deferred split-K sum.  With `deferred` != NULL (host pointer) and alpha == 1, a GEMM that
 * splits K leaves its n = *deferred partial results [n][M][N] at the start of ws instead of
 * running the reduce pass, and does not touch C; *deferred = 0 means C is complete.  The
 * consumer kernels below that take (slabs, n_slabs, beta_pre[, bias_pre]) add the slabs in
 * the reduce pass's order (slabs ascending, + bias, + beta * old input), write the total
 * back to their input buffer and continue: one launch less per layer of the latency-bound
 * scans.  Nothing else may use that workspace in between.  dd_splitk_finish is the
 * stand-alone pass (used when a consumer cannot take slabs). */
int dd_splitk_finish(const float* slabs, int n_slabs, float* C, long ldc, int M, int N,
                     float beta, const float* bias, void* stream);

/* Arithmetic of every contraction below: 6 (default) = fp32 operands split exactly into
 * three bf16 terms, six cross products on the bf16 matrix pipe with fp32 accumulation
 * (fp32-level accuracy: measured max error 3.3e-6 of the output scale at K = 4096 vs
 * 3.1e-6 for mode 0); 0 = native fp32 MFMA; 1 = REDUCED precision, opt-in only
 * (`hip.precision: bfloat16`): operands rounded to bf16 (nearest even), one product per pair,
 * fp32 accumulation and fp32 storage - the counterpart of the reference's tf.precision
 * float16 compute dtype (tfagent.py:161-168, tfutils.py:164-167; bf16 keeps the fp32 exponent
 * range, so no loss scaling is needed), with its own looser parity tolerance.  Also settable
 * with the environment variable DD_GEMM_MODE before the first call.  Returns the previous
 * mode, -1 for an invalid one. */
int dd_gemm_set_mode(int mode);
/* Selector of the role-separated form of the 128x128 contraction loop (k_mfma_gemm_ws: four MFMA
 * waves + four staging waves per workgroup, bit-identical results): used for launches whose
 * contraction length per split-K slab is >= kmin (banded transposed-convolution launches: kmin_tc);
 * on = 0 keeps the product loop everywhere.  kmin / kmin_tc <= 0 leave the threshold unchanged.
 * Off by default: measured equal to the product loop in steady state (both are bound by the
 * power-limited clock).  Environment defaults: DD_WS (0), DD_WS_KMIN (1024),
 * DD_WS_KMIN_TC (1024).  Returns the previous
 * `on`.  Process-wide, not thread-safe against concurrent launches (a measurement switch). */
int dd_gemm_set_ws(int on, int kmin, int kmin_tc);

/* Stride-2 VALID convolution family over NHWC tensors.  "big" is the
 * full-resolution side [n,hb,wb,Cb], "small" the downsampled side
 * [n,hs,ws,Cs], filter w[k,k,Cb,Cs] (= tf.nn.conv2d's [kh,kw,in,out] for the
 * encoder, tf.nn.conv2d_transpose's [kh,kw,out,in] for the decoder).
 *   down : small = conv(big) (+bias[Cs])     tf.nn.conv2d nets.py:547 (encoder
 *          forward) and the input-gradient of conv2d_transpose (decoder bwd)
 *   up   : big = conv^T(small) (+bias[Cb])   tf.nn.conv2d_transpose nets.py:539
 *          (decoder forward) and the input-gradient of conv2d (encoder bwd)
 *   wgrad: dw = beta*dw + sum big (x) small  filter gradient of both
 * big may be uint8 (big_is_u8=1, values scaled by in_scale: the `/255`
 * of Agent.preprocess agent.py:129-130 fused into the first conv). */
int dd_conv2d_s2_down(const void* big, int big_is_u8, const float* w, const float* bias,
                      float* small, int n_img, int hb, int wb, int Cb,
                      int hs, int ws_, int Cs, int k, float in_scale,
                      float* ws, size_t ws_bytes, void* stream);
int dd_conv2d_s2_up(const float* small, const float* w, const float* bias, float* big,
                    int n_img, int hs, int ws_, int Cs, int hb, int wb, int Cb, int k,
                    float* ws, size_t ws_bytes, void* stream);
int dd_conv2d_s2_wgrad(const void* big, int big_is_u8, const float* small, float* dw,
                       int n_img, int hb, int wb, int Cb, int hs, int ws_, int Cs, int k,
                       float in_scale, float beta, float* ws, size_t ws_bytes, void* stream);

/* dd_conv2d_s2_down of an image-side layer (big has <= 4 channels, Cs = 64) followed by its
 * LayerNorm + ELU (the encoder's first layer, nets.py:291-305, Norm :585-602) in one pass: a
 * pixel's 64 channels are in registers when the contraction ends, so `small` (pre-norm), `out`
 * (activations) and stats [pixels, 2] = mean / rstd are written without a second pass over
 * `small` - the results of dd_conv2d_s2_down + dd_ln_act_fwd.  Returns 1 with nothing computed
 * when the geometry is not covered. */
int dd_conv2d_s2_down_ln(const void* big, int big_is_u8, const float* w, const float* bias,
                         const float* gamma, const float* beta_ln, float* small, float* out, float* stats,
                         int n_img, int hb, int wb, int Cb, int hs, int ws_, int Cs, int k, float in_scale,
                         float* ws, size_t ws_bytes, void* stream);
/* dd_conv2d_s2_down as the data gradient of the decoder's image layer (big = d loss / d image,
 * float, <= 4 channels; Cs = 64) followed by the LayerNorm + ELU backward of the layer in front of
 * it (nets.py:308-327, Norm :585-602): the gradient at that layer's OUTPUT exists only in the
 * accumulators; dz = d loss / d (its pre-norm rows) and dgamma / dbeta / dbias are what
 * dd_conv2d_s2_down + dd_ln_act_bwd (activation recomputed from z) produce.  Returns 1 with nothing
 * computed when the geometry is not covered. */
int dd_conv2d_s2_down_lnbwd(const float* big, const float* w, const float* z, const float* stats,
                            const float* gamma, const float* beta_ln, float* dz, float* dgamma,
                            float* dbeta, float* dbias, int accumulate, int n_img, int hb, int wb, int Cb,
                            int hs, int ws_, int Cs, int k, float* ws, size_t ws_bytes, void* stream);
/* dd_conv2d_s2_wgrad for an image-side layer (big has 3 channels) whose small side is a Conv2D +
 * LayerNorm + ELU (the encoder's first layer, nets.py:291-305, Norm :585-602): `dout` is the
 * gradient at the layer OUTPUT [n, hs, ws, Cs]; dd_ln_act_bwd's arithmetic (activation recomputed
 * from z, stats = [pixels, 2] mean / rstd) runs on the rows as they are staged, so dz is never
 * written to or read from HBM.  Produces dw (beta * dw + ...), and what dd_ln_act_bwd would:
 * dgamma, dbeta and dbias = column sum of dz (accumulated when `accumulate`).  Returns 1 with
 * nothing launched when the geometry is not covered (then: dd_ln_act_bwd + dd_conv2d_s2_wgrad). */
int dd_conv2d_s2_wgrad_ln(const void* big, int big_is_u8, float in_scale, const float* dout,
                          const float* z, const float* stats, const float* gamma, const float* beta_ln,
                          float* dw, float beta, float* dgamma, float* dbeta, float* dbias, int accumulate,
                          int n_img, int hb, int wb, int Cb, int hs, int ws_, int Cs, int k,
                          float* ws, size_t ws_bytes, void* stream);

/* Stride-1 SAME convolutions (odd k; NHWC, filter [k,k,Cin,Cout]) of the residual encoder /
 * decoder: ImageEncoderResnet nets.py:330-358, ImageDecoderResnet nets.py:361-391, through
 * Conv2D's defaults stride 1 / pad 'same' (nets.py:497-499, tf.nn.conv2d nets.py:547).
 *   dd_conv2d_same          y  = alpha*conv(x, w) + bias[Cout] + beta*y   (x may be uint8, scaled
 *                                by in_scale: the image layer 'in' with the `/255` fused)
 *   dd_conv2d_same_bwd_data dx = alpha*conv(dy, rot180(w)^T) + beta*dx    (input gradient; the
 *                                rotated filter is written to the head of ws: ws_bytes must
 *                                cover k*k*Cin*Cout floats rounded up to 256 bytes)
 *   dd_conv2d_same_wgrad    dw = alpha*sum_pixels x (x) dy + beta*dw      (filter gradient)
 * A tap outside the image contributes an exact zero. */
int dd_conv2d_same(const void* x, int x_is_u8, const float* w, const float* bias, float* y,
                   int n_img, int h, int wd, int Cin, int Cout, int k, float in_scale,
                   float alpha, float beta, float* ws, size_t ws_bytes, void* stream);
int dd_conv2d_same_bwd_data(const float* dy, const float* w, float* dx, int n_img, int h,
                            int wd, int Cin, int Cout, int k, float alpha, float beta,
                            float* ws, size_t ws_bytes, void* stream);
int dd_conv2d_same_wgrad(const void* x, int x_is_u8, const float* dy, float* dw, int n_img,
                         int h, int wd, int Cin, int Cout, int k, float in_scale,
                         float alpha, float beta, float* ws, size_t ws_bytes, void* stream);
/* y[n,i,j,c] = scale * (sum of the 2x2 block of x) with y of ho x wo pixels, x of 2ho x 2wo:
 * tf.nn.avg_pool(x, 2, 2, 'SAME') nets.py:343 (scale 0.25) and the gradient of dd_repeat2 (1). */
int dd_pool2(const float* x, float* y, long n_img, int ho, int wo, int C, float scale,
             void* stream);
/* y[n,2i+a,2j+b,c] = scale * x[n,i,j,c] + beta*y with x of hi x wi pixels:
 * tf.repeat(tf.repeat(x, 2, 1), 2, 2) nets.py:378 (scale 1) and the gradient of avg_pool (0.25). */
int dd_repeat2(const float* x, float* y, long n_img, int hi, int wi, int C, float scale,
               float beta, void* stream);

/* ---- LayerNorm(+ELU), GRU cell -------------------------------------------- */

/* out = act(LN(z)*gamma+beta), eps 1e-3, population variance; act 0 none,
 * 1 elu.  stats[rows,2] = (mean, rstd), row stride lds.  Norm nets.py:585-602 + get_act. */
int dd_ln_act_fwd(float* z, long ldz, const float* gamma, const float* beta,
                  float* out, long ldo, float* stats, long lds, int rows, int C, int act,
                  const float* slabs, int n_slabs, float beta_pre, const float* bias_pre,
                  void* stream);
/* dz from dout; if dgamma != NULL also dgamma/dbeta and, if dbias_pre != NULL,
 * the column sum of dz (gradient of a bias added before the norm, as in Conv2D
 * nets.py:548-553); accumulate: += .  The activation's derivative is taken from the stored
 * activation `out`, or - out == NULL, beta_ln = the LayerNorm offset - from z, stats, gamma and
 * beta_ln recomputed with dd_ln_act_fwd's own expression (bit-identical result, one tensor less to
 * read: the pass is HBM-bound).  beta_ln may be NULL when out is given.  (ABI version 4.) */
int dd_ln_act_bwd(float* dout, long ldd, const float* z, long ldz,
                  const float* out, long ldo, const float* stats, long lds, const float* gamma,
                  const float* beta_ln,
                  float* dz, long lddz, float* dgamma, float* dbeta, float* dbias_pre,
                  int accumulate,
                  int rows, int C, int act, float* ws, size_t ws_bytes,
                  const float* slabs, int n_slabs, float beta_pre, void* stream);
/* A Linear + LayerNorm + ELU layer that feeds a ONE-UNIT output layer (the reward / cont / critic
 * heads: MLP -> DistLayer with shape (), nets.py:428-492): the output layer is folded into the
 * LayerNorm kernels instead of running as a 1-column contraction.  Forward: head_out[row] =
 * out[row, :] . head_w + head_b[0] next to everything dd_ln_act_fwd writes.  Backward: the layer's
 * output gradient is head_dy[row] * head_w[:] (formed in registers, never stored: no `dout`
 * tensor), everything else as dd_ln_act_bwd without a deferred sum.  Vector path only
 * (C % 4 == 0, C <= 1024, 16-byte aligned rows).  (ABI 10.) */
int dd_ln_act_fwd_head(float* z, long ldz, const float* gamma, const float* beta,
                       float* out, long ldo, float* stats, long lds, int rows, int C, int act,
                       const float* slabs, int n_slabs, float beta_pre, const float* bias_pre,
                       const float* head_w, const float* head_b, float* head_out, void* stream);
int dd_ln_act_bwd_head(const float* head_dy, const float* head_w, const float* z, long ldz,
                       const float* out, long ldo, const float* stats, long lds, const float* gamma,
                       const float* beta_ln,
                       float* dz, long lddz, float* dgamma, float* dbeta, float* dbias_pre,
                       int accumulate, int rows, int C, int act, float* ws, size_t ws_bytes,
                       void* stream);
int dd_ln_bwd_parts(int rows, int C);
/* dgamma/dbeta only, from stored activations of all scan steps. */
int dd_ln_param_grad(const float* dout, long ldd, const float* z, long ldz,
                     const float* out, long ldo, const float* stats, long lds,
                     float* dgamma, float* dbeta, int accumulate, int rows, int C,
                     int act, float* ws, size_t ws_bytes, void* stream);

/* out[c] = beta*out[c] + sum_rows x[row][c]: bias gradients. */
int dd_col_sum(const float* x, long ldx, float* out, float beta, long rows, int C,
               float* ws, size_t ws_bytes, void* stream);

/* RSSM._gru nets.py:149-160 after the [D+U,3D] matmul: LayerNorm over all 3D,
 * reset/cand/update gates, new deter.  z3 [rows,3D]; h, hn [rows,D]. */
int dd_gru_cell_fwd(float* z3, long ldz, const float* gamma, const float* beta,
                    const float* h, long ldh, float* hn, long ldn, float* stats, long lds,
                    int rows, int D, const float* slabs, int n_slabs, float beta_pre,
                    void* stream);
/* dz3 (through LN), dh = (1-update)*dhn, dy3 = gradient at the LN output.
 * zx (NULL ok): a [rows,U] region to zero-fill, so that one beta=1 GEMM can
 * then accumulate dz3 @ W^T into the adjacent [dh | dx] buffer. */
int dd_gru_cell_bwd(const float* dhn, long lddn, const float* z3, long ldz,
                    const float* stats, long lds, const float* gamma, const float* beta,
                    const float* h, long ldh, float* dz3, long lddz,
                    float* dh, long lddh, float* dy3, long lddy,
                    float* zx, long ldzx, int U, int rows, int D, void* stream);

/* ---- fused observe scan ------------------------------------------------------- */

/* RSSM.observe forward (nets.py:66-76, obs_step :99-117, _gru :149-160) for all T steps in ONE
 * persistent launch (csrc/scan.hip): per step four phases separated by grid barriers -
 * img_in, GRU, obs_out (+ the hoisted embed part already in zo), obs_stats + the latent draw -
 * each a small-M MFMA contraction whose operand is built from the RAW output rows of the
 * previous phase (the consumer applies LayerNorm / ELU / GRU gates).  Writes every buffer the
 * unfused launch sequence writes (same layouts, rows b*T + t): xin[:, :S] (masked stoch),
 * z1 / st1 / gin (= [hprev | x1]), z3 / gst, post (= [deter | stoch]), zo / xo / st3, xq,
 * post_logit.  The draw is the shared sampler (latent_core.h): same indices as
 * dd_stats_sample_fwd / dd_onehot_sample_host given the same statistics.
 * wt2..wt4: weight caches from dd_scan_wprep for gru_out (N = 3D columns, Kp = D+U), obs_out[:D]
 * (U, D), obs_stats (S, U); wt1 is not read (img_in's one-hot part is a gather of w_in's rows).
 * The cache holds three bf16 planes (exact 3-way split of the fp32 weight) in FRAGMENT-MAJOR order:
 * [column tile n/16][k-step k/128][plane][thread 0..255] x 8 values, thread (wave w, lane l) holding
 * k = k-step*128 + w*32 + (l>>4)*8 .. +7 of column tile*16 + (l&15) - a wave's 16-byte load is 1 KB
 * contiguous.  N % 16 == 0, Kp % 128 == 0, 3*N*Kp values.  sync2: 1088 zero-initialised device words - [0] grid-wide
 * barrier counter (use_carry bit 7 selects it; bit 8: release fence at every arrival instead of
 * write-through stores - measurement aids), [1] error word, [2, 576) time
 * stamps of the measurement flag, [576 + 128 m] the barrier counter of row block m (a phase only
 * consumes what the 16 workgroups of its own 16-row block produced, so the blocks synchronise
 * separately; use_carry bit 9: the older workgroup -> (row block, stride) map, a measurement aid).
 * The launch resets the counters only (by a kernel - never a memset node in a captured graph); the error word is sticky (bit 0: a
 * bounded grid-barrier spin timed out, bit 1: a carried / initial stoch group was not one-hot):
 * the host reads it after the step, raises on non-zero and clears it.  w_in: the
 * img_in kernel itself [S+A, U] (the one-hot stoch part of that layer is a gather of its rows);
 * idx_ws: (B*T + B + 1) * G ints of scratch (drawn classes per row, of the carry, of the
 * initial state).
 * dd_observe_scan_supported: the compiled shapes (B <= 64; deter = units = 256 or 512, 32 x 32
 * latents, the action widths of the BASELINE configs) on a device with at least 64 CUs (the grid
 * barrier needs every workgroup resident: a smaller compute partition keeps the launch sequence). */
int dd_observe_scan_supported(int B, int D, int U, int G, int C, int A);
int dd_scan_wprep(const float* W, long ld, int K, int N, int Kp, void* planes, void* stream);
int dd_observe_scan_fwd(
    int B, int T, int D, int U, int G, int C, int A, int use_carry, float unimix,
    const float* first, const float* carry, const float* init_deter, const float* init_stoch,
    const float* u_post,
    const void* wt1, const void* wt2, const void* wt3, const void* wt4,
    const float* g1, const float* b1, const float* gg, const float* bg, const float* g3,
    const float* b3, const float* bias4,
    float* xin, float* z1, float* st1, float* gin, float* z3, float* gst, float* post, float* zo,
    float* xo, float* st3, float* xq, float* post_logit, const float* w_in, int* idx_ws,
    unsigned* sync2, void* stream);

/* Weight cache of the reverse scan: W [N, K] fp32 (row stride ld) -> three bf16 planes in the
 * fragment-major order of dd_scan_wprep with column n = W's row n (the backward contractions
 * multiply by W^T); N % 16 == 0, K % 128 == 0. */
int dd_scan_wprep_rows(const float* W, long ld, int N, int K, void* planes, void* stream);

/* Fused reverse observe scan: the data gradient of the T obs_steps (what tf.GradientTape derives
 * for RSSM.observe, nets.py:66-76,99-160) in one persistent launch; replaces the per-step launch
 * sequence stats bwd / obs_stats dgrad / LN-ELU bwd / obs_out dgrad / GRU bwd / GRU dgrad /
 * LN-ELU bwd / img_in dgrad / reset mask.  Rows are b*T + t.  On entry dfeat [N, D+S] holds the
 * heads' gradient w.r.t. every posterior, dlogit the KL gradient w.r.t. the posterior logits and
 * dxq the stats gradient of the LAST step's rows (dd_stats_sample_bwd); on return dxq, dxo, dzo,
 * dz3, dy3, dgin = [dh | dx1], dz1, dxs hold what the bulk weight-gradient contractions read.
 * w1..w4: dd_scan_wprep_rows planes of obs_stats [U,S], obs_out (deter rows) [D,U], gru [D+U,3D],
 * img_in (stoch rows) [S,U].  flags bit 6: time stamps (measurement aid).  sync2: as in the
 * forward scan.  Shapes: dd_observe_scan_bwd_supported (deter = units = 256 or 512, 32 x 32 latents;
 * at 512 the planes of three of the four phases are streamed instead of held in registers). */
int dd_observe_scan_bwd_supported(int B, int D, int U, int G, int C);
int dd_observe_scan_bwd(
    int B, int T, int D, int U, int G, int C, int flags, float unimix, const float* first,
    const float* xq, const float* zo, const float* xo, const float* st3, const float* z3,
    const float* gst, const float* gin, const float* z1, const float* st1, const float* dlogit,
    const void* w1, const void* w2, const void* w3, const void* w4,
    const float* g3, const float* gg, const float* bg, const float* g1,
    float* dfeat, float* dxq, float* dxo, float* dzo, float* dz3, float* dy3, float* dgin,
    float* dz1, float* dxs, unsigned* sync2, void* stream);

/* ---- categorical latent ------------------------------------------------------ */

/* logit = log((1-unimix)*softmax(x)+unimix/C); stoch = one_hot(draw).
 * mode 0: inverse-CDF draw with u[rows,G]; mode 1: argmax.
 * RSSM._stats_layer nets.py:165-170 + OneHotDist.sample tfutils.py:368-378. */
int dd_stats_sample_fwd(float* x, long ldx, const float* u, long ldu,
                        float* logit, long ldl, float* stoch, long lds,
                        int rows, int G, int C, float unimix, int mode,
                        const float* slabs, int n_slabs, float beta_pre, const float* bias_pre,
                        void* stream);
/* Host twin of dd_stats_sample_fwd (no GPU, no stream): the categorical draw on the host
 * cores from the same source as the kernel (csrc/sampler_core.h: deterministic exp, the
 * sub-wave's butterfly sum and Kogge-Stone scan orders), so the class index of every
 * (row, group) equals the device's bit for bit given the same x and u.  Optional outputs:
 * logit / stoch as the kernel writes them, index int[rows, G] (row stride ldi).
 * OneHotDist.sample tfutils.py:368-378 (tf.random.categorical :374). */
int dd_onehot_sample_host(const float* x, long ldx, const float* u, long ldu,
                          float* logit, long ldl, float* stoch, long lds,
                          int* index, long ldi, int rows, int G, int C,
                          float unimix, int mode);
/* dx from dlogit (NULL ok) and the straight-through dstoch (NULL ok),
 * tfutils.py:380-381. */
int dd_stats_sample_bwd(const float* x, long ldx, const float* dlogit, long ldl,
                        const float* dstoch, long lds, float* dx, long lddx,
                        int rows, int G, int C, float unimix, void* stream);
/* kl[row] = sum_g KL(post||prior) and both entropies.  RSSM.kl_loss
 * nets.py:178-183 (value; lhs == rhs numerically). */
int dd_cat_kl_fwd(const float* post, long ldp, const float* prior, long ldq,
                  float* kl, float* ent_post, float* ent_prior,
                  int rows, int G, int C, void* stream);
/* Balanced gradient: dpost = c*(1-balance)*dKL/dpost, dprior = c*balance*
 * dKL/dprior with c = coef_host * (*coef_dev). */
int dd_cat_kl_bwd(const float* post, long ldp, const float* prior, long ldq,
                  const float* coef_dev, float coef_host, float balance,
                  float* dpost, long lddp, float* dprior, long lddq,
                  int rows, int G, int C, void* stream);

/* ---- losses and imagination scalars --------------------------------------------- */

/* sigmoid + MSEDist(sum) against u8 image/255: nets.py:325, tfutils.py:320-329.
 * Channels [c0,c1) of ctot: one call per image key of the channel-concatenated
 * decoder output (nets.py:274-277). */
int dd_image_loss(const float* z, const unsigned char* img, float* loss, float* dz,
                  int rows, long P, int ctot, int c0, int c1, float coef, void* stream);
/* Agent.report's videos assembled on the device (tfutils.video_grid, tfutils.py:390-392, of
 * WorldModel.report agent.py:266-282 and Greedy.report behaviors.py:32-46).  z: pre-sigmoid decoder
 * output, images of H x W x ctot floats; image (b, t) is image number b * zsb + t * zst; channels
 * [c0, c1) are one image key.  img = the uint8 truth in the same layout: out [nt, 3H, nb*W, c1-c0]
 * = truth / 255 | sigmoid(z) | (model - truth + 1) / 2 stacked on the height axis; img = NULL:
 * out [nt, H, nb*W, c1-c0] = sigmoid(z).  (ABI 7.) */
int dd_video_grid(const float* z, const unsigned char* img, float* out, int nb, int nt,
                  int H, int W, int ctot, int c0, int c1, long zsb, long zst, void* stream);
int dd_mse_loss(const float* pred, long ldp, const float* tgt, long ldt, float* loss,
                float* dpred, long lddp, int rows, int D, float coef, void* stream);
/* kind 0 SymlogDist tfutils.py:347-356; kind 1 Bernoulli(logits) nets.py:469-471 */
int dd_scalar_loss(const float* pred, const float* tgt, float* loss, float* dpred,
                   long n, float coef, int kind, void* stream);
/* tfutils.action_noise tfutils.py:85-93 (Agent.policy's expl_noise / eval_noise, agent.py:62-63),
 * in place on act[rows, A]: continuous (discrete = 0) a <- clip(a + amount*noise, -1, 1) with
 * noise[rows, A] standard normals; discrete a <- one_hot(draw from amount/A + (1-amount)*a) with
 * one uniform per row in noise[:, 0].  amount == 0: no-op. */
int dd_action_noise(float* act, long lda, const float* noise, long ldn, int rows, int A,
                    float amount, int discrete, void* stream);
/* Normal head nets.py:461-468 with reparameterised sample (eps NULL: mode). */
int dd_normal_head_fwd(const float* om, long ldm, const float* os, long ldsd,
                       const float* eps, long lde, float* act, long lda,
                       int rows, int A, float lo, float hi, void* stream);
int dd_normal_head_bwd(const float* om, long ldm, const float* os, long ldsd,
                       const float* eps, long lde, const float* dact, long ldda,
                       const float* w, const float* scale,
                       float* dom, long lddm, float* dos, long lddsd, float* ent_row,
                       int rows, int rows_ent, int A, float lo, float hi,
                       float ent_coef, float ent_lo, float ent_div, void* stream);
int dd_actent_stats(const float* os, long ldsd, int rows, int A, float lo, float hi,
                    float ent_lo, float ent_div, double* out, double* ws,
                    size_t ws_bytes, void* stream);
/* symexp / sigmoid of the head outputs, discount weights agent.py:256-259 and
 * VFunction.target: the 'gve' lambda-return agent.py:434-440 (gae = 0) or the same return
 * summed as generalised advantages, 'gae' agent.py:428-433 (gae = 1); one thread per
 * trajectory.  dd_imag_returns_bwd serves both (identical derivative). */
int dd_imag_returns_fwd(const float* rew_raw, const float* val_raw, const float* cont_raw,
                        const float* first_cont, float* reward, float* value,
                        float* cont, float* weight, float* ret, int H, long N,
                        float gamma, float lam, int gae, void* stream);
int dd_imag_returns_bwd(const float* dret, const float* dbase, const float* rew_raw,
                        const float* val_raw, const float* cont_raw, const float* value,
                        const float* ret, float* d_rew_raw, float* d_val_raw,
                        float* d_cont_raw, int H, long N, float gamma, float lam, void* stream);
int dd_critic_loss(const float* out, const float* ret, const float* w, float* loss,
                   float* dout, long n, float coef, void* stream);
int dd_actor_seed(const float* ret, const float* base, const float* w, const float* ent_row,
                  const float* sc, float* loss, float* dret, float* dbase, long n, float coef,
                  void* stream);
int dd_sub(const float* a, const float* b, float* o, long n, void* stream);
/* o = symexp(x) = sign(x) * (exp(|x|) - 1): SymlogDist.mean() (tfutils.py:345-349) of the critic's
 * own prediction on the imagined states, metrics imag_critic_mean / _std (agent.py:411-412). */
int dd_symexp(const float* x, float* o, long n, void* stream);
/* One-hot policy head (nets.py:480-491) trained by REINFORCE (agent.py:357-358):
 * normalised entropy per row, and d loss / d logit of -logp(a)*sg(score) plus the
 * entropy regulariser, with score = ((ret-base)*sc[0]-sc[1])*sc[2]. */
int dd_onehot_entropy(const float* logit, long ldl, float* ent_out, int rows, int A,
                      float ent_div, void* stream);
int dd_onehot_policy_grad(const float* logit, long ldl, const float* action, long lda,
                          const float* ret, const float* base, const float* w,
                          const float* sc, const float* scale, float* dlogit, long lddl,
                          float* loss_pg, float* loss_ent, int rows, int rows_grad,
                          int A, float coef, float ent_div, void* stream);

/* ---- learner state ------------------------------------------------------------- */

/* Philox4x32-10 noise keyed by (seed, *step_dev, site, global row).
 * kind 0 uniform [0,1), kind 1 standard normal. */
int dd_philox(float* out, long outer, long inner, int cols, long inner_global,
              long inner_offset, unsigned long long seed,
              const unsigned long long* step_dev, unsigned site, int kind, void* stream);
int dd_counter_add(unsigned long long* counter, unsigned long long v, void* stream);
/* measurement aid: slot[0] counts the stamps, slot[1 + i % 8] = the device's 100 MHz wall clock when
 * the stream reached this point the i-th time (nine words per slot) */
int dd_stamp(unsigned long long* slot, void* stream);
/* sums = {sum, sum sq, sum abs} (fp64), maxs = {max, max(-x), max|x|}. */
int dd_reduce_stats(const float* x, long n, long stride, double* sums, float* maxs, void* stream);
/* dd_reduce_stats of `count` <= 16 vectors in one launch (HOST arrays of device pointers / sizes;
 * the same sums / maxs per vector, bit for bit).  (ABI 8.) */
int dd_reduce_stats_multi(int count, const float* const* x, const long* n, const long* stride,
                          double* const* sums, float* const* maxs, void* stream);
int dd_autoadapt_update(float* scale, const double* sums, int n, double count,
                        float target, float thres, float vel, float lo, float hi,
                        int inverse, int impl, void* stream);
int dd_normalize_update(double* state, const double* sums, double count,
                        const float* in_scale_dev, double decay, double maxv, int impl,
                        int do_update, float* out_off_scale, void* stream);
int dd_scalar_mul(float* dst, const float* a, const float* b, float c, int n, void* stream);
/* opt_state = {step, grad_norm, finite, grad_scale, good_steps}: FIVE doubles, all written by
 * every call (ABI version >= 2; version 1 had three and no `mixed` argument):
 * tf.linalg.global_norm tfutils.py:243; the step advances only on a finite norm (:255-260).
 * mixed != 0: also the loss-scale controller of the reduced-precision mode (tfutils.py:225-240:
 * overflow halves the scale, 1000 good steps double it, clip [1e-4, 1e4]). */
int dd_grad_norm(const float* g, long n, double* opt_state, double* ws, size_t ws_bytes,
                 int mixed, void* stream);
/* clip + weight decay (first n_decay elements) + Adam, tfutils.py:244-283.  warmup > 0
 * (ABI 6, tfutils.py:160-162): the learning rate is lr * clip(step / warmup, 0, 1) with the step
 * count at the time of use - the decay sees the count before this step's increment, Adam the one
 * after it (opt_state[0], advanced by dd_grad_norm); 0 = off. */
int dd_adam_step(float* p, const float* g, float* m, float* v, long n, long n_decay,
                 const double* opt_state, float lr, float wd, float eps, float b1,
                 float b2, float clip, float warmup, void* stream);
int dd_fill(float* p, long n, float v, void* stream);
/* y = (accumulate ? y : 0) + alpha * (alpha_dev ? alpha_dev[0] : 1) * x: the summed, scaled
 * world-model loss map of agent.py:186-190 (its mean / std are metrics). */
int dd_axpy(const float* x, float alpha, const float* alpha_dev, float* y, long n,
            int accumulate, void* stream);
/* tfutils.balance_stats tfutils.py:395-411 of a scalar head (kind 0: symlog-MSE head, mean =
 * symexp(out); kind 1: Bernoulli head, mean = sigmoid(out)) as seven float64 sums out7 =
 * {loss*pos, loss*neg, pred*pos, (1-pred)*neg, pos, target, mean}. */
int dd_balance_stats(const float* out, const float* target, const float* loss, long n,
                     float thres, int kind, double* out7, double* ws, size_t ws_bytes,
                     void* stream);
int dd_copy2d(const float* src, long lds, float* dst, long ldd, long rows, int cols, void* stream);
/* is_first reset of RSSM.obs_step nets.py:100-107 and its gradient. */
int dd_reset_mask(const float* prev, long ldp, const float* first, long fstride,
                  const float* init, float* out, long ldo, long rows, int cols, void* stream);
int dd_reset_mask_bwd(const float* dout, long ldo, const float* first, long fstride,
                      float* dprev, long ldp, long rows, int cols, void* stream);
/* The same for two column segments at once (the [deter | stoch] halves of the carried
 * state go to different buffers): one launch per scan step instead of two. */
int dd_reset_mask2(const float* prev_a, long ldpa, const float* init_a, float* out_a, long ldoa, int cols_a,
                   const float* prev_b, long ldpb, const float* init_b, float* out_b, long ldob, int cols_b,
                   const float* first, long fstride, long rows, void* stream);
int dd_reset_mask_bwd2(const float* dout_a, long ldda, float* dprev_a, long ldpa, int cols_a,
                       const float* dout_b, long lddb, float* dprev_b, long ldpb, int cols_b,
                       const float* first, long fstride, long rows, void* stream);
int dd_batch_prep(const unsigned char* is_first, const unsigned char* is_terminal,
                  const float* action, float* first_f, float* cont_f,
                  float* act_masked, long ldm, long n, int A, void* stream);
/* Minibatch assembly from an HBM-resident replay ring: out[b, t, :] = ring[starts[b] + t, :]
 * for row_bytes-byte rows (any dtype), b < B, t < T.  first_flag != 0 writes the
 * chunk's is_first column instead (1 at t = 0, else 0).  Replaces the host-side
 * slicing + np.stack + upload of FixedLength._sample / Prefetch.__next__ / TFAgent._convert_inps
 * (replay/fixed_length.py:64-81, core/prefetch.py:55, tfagent.py:105-116). */
int dd_replay_gather(const void* ring, long row_bytes, const long long* starts, int B, int T,
                     void* out, int first_flag, void* stream);
int dd_tanh_fwd(const float* x, float* y, int n, void* stream);
int dd_tanh_bwd(const float* x, const float* dy, float* dx, int n, float beta, void* stream);

/* ---- fused imagination rollout ------------------------------------------------------------
 * WorldModel.imagine (agent.py:234-254) as one persistent launch: per 16-row block of the N
 * start states, all H steps of {actor MLP + normal head + action sample, RSSM.img_step
 * (nets.py:119-138), categorical draw}.  Rows never interact, so there is no grid-wide
 * synchronisation.  Writes every buffer the per-layer launch sequence writes (pre-norm z,
 * LayerNorm statistics, outputs per layer, z3, raw statistics, traj[1..H], actions).
 * dd_imag_wprep: W [K, n] fp32 -> columns col0.. of the fragment-major bf16 plane cache
 * [Npad/16][K/32][3][64 lanes][8] of a [K, Npad] operand (exact 3-way split).
 * dd_imagine_rollout_fwd: `ptrs` is a HOST array of n_ptrs = 65 device pointers, order documented
 * at the definition (csrc/imag.hip).  A launch runs the policy of steps t0 .. t1 - 1 and the
 * img_steps of those below H from the state in traj[t0] (0 <= t0 < t1 <= H + 1: the whole rollout
 * is t0 = 0, t1 = H + 1; a split lets the caller work on finished time rows meanwhile).
 * Shapes: dd_imagine_rollout_supported. */
int dd_imag_wprep(const float* W, long ld, int K, int n, int col0, void* planes, void* stream);
/* Reverse pass (actor_grad 'backprop', agent.py:355-356: the data gradient of the score through
 * the imagined world model): from d score / d state_t in dtraj[t][:, :F] (heads, bulk launches)
 * steps t = H .. 1 of {draw straight-through + img_stats, img_out 2..0, GRU, img_in} backward in
 * one persistent launch; leaves dtraj[t][:, :D] = total gradient of deter_t and adds the step's
 * contribution to dtraj[t-1][:, D:] (stoch | action), as the per-layer launch sequence does.
 * dd_imag_wprep_t: the transposed operand, W stored [n, K] -> planes of B[k][col] = W[col][k].
 * `ptrs`: HOST array of 29 device pointers, order at the definition (csrc/imag.hip). */
int dd_imag_wprep_t(const float* W, long ld, int K, int n, void* planes, void* stream);
int dd_imagine_rollout_bwd(int N, int H, int D, int U, int G, int C, int A, float unimix,
                           const void* const* ptrs, int n_ptrs, void* stream);
int dd_imagine_rollout_supported(int D, int U, int G, int C, int A, int actor_units,
                                 int actor_layers, int prior_layers, int discrete);
int dd_imagine_rollout_fwd(int N, int H, int t0, int t1, int D, int U, int G, int C, int A,
                           int actor_units, float unimix, float lo, float hi,
                           const void* const* ptrs, int n_ptrs, void* stream);
/* Rows of the imagination batch per workgroup of dd_imagine_rollout_fwd: 32 (default: a workgroup
 * multiplies two 16-row operand tiles against every streamed weight fragment, the launch holds
 * half the CUs) or 16 (the round 3-5 kernel; also DD_IMAG_ROWS=16).  Same outputs, same arithmetic
 * per row.  Returns the previous value. */
int dd_imag_set_rows(int rows);
/* The same for ONE-HOT action spaces (actor_grad 'reinforce', agent.py:357-358: no gradient through
 * the dynamics, so forward only) at deter = units = 512 (xarm / ur5 blocks, configs.yaml:245-295):
 * policy head = Linear(A) + unimix softmax + inverse-CDF draw (the class dd_stats_sample_fwd draws
 * from u_act, bit for bit); writes the actor's activations, its raw logits, normalised
 * log-probabilities, the one-hot actions and states of traj[t0 + 1 .. t1 - 1] (rows of `row_width`
 * floats, a multiple of four >= deter + stoch + A), and the img_step buffers the launch sequence
 * writes.  `ptrs`: HOST array of 64 device pointers, order at the definition (csrc/imag_oh.hip).
 * Shapes: dd_imagine_rollout_supported(..., discrete = 1).  (ABI 9.) */
int dd_imagine_rollout_oh_fwd(int N, int H, int t0, int t1, int D, int U, int G, int C, int A,
                              int actor_units, int row_width, float unimix, float actor_unimix,
                              const void* const* ptrs, int n_ptrs, void* stream);

/* ---- launch runtime: process-owned streams and HIP-graph segments ------------------------
 * Role of the reference's concrete-function cache (tfagent.py:56-70, tf.function :60-64): the
 * step is captured once and replayed.  Streams are created per role by the library (never a
 * handle of a shared pool); a graph executable lives until dd_graph_destroy, which is only
 * safe after a device synchronisation.  dd_graph_capture_end returns *exec_out = NULL for a
 * segment without nodes; dd_graph_launch(NULL, ..) is a no-op.  A segment may hold kernel launches
 * (and the empty nodes of stream joins) only: dd_graph_capture_end fails on a memset / copy node
 * unless DD_GRAPH_ALLOW_NONKERNEL=1 (then it reports them on stderr). */
int dd_stream_create(void** stream_out);
int dd_stream_destroy(void* stream);
int dd_graph_capture_begin(void* stream);
int dd_graph_capture_end(void* stream, void** exec_out, int* nodes_out);
int dd_graph_launch(void* exec, void* stream);
int dd_graph_destroy(void* exec);
/* Native (C) backtrace to stderr on SIGSEGV / SIGBUS / SIGABRT / SIGFPE / SIGILL, then the
 * previously installed handler runs (Python's faulthandler, or the default action). */
int dd_install_crash_handler(void);

#ifdef __cplusplus
}
#endif
#endif  /* DAYDREAMER_HIP_H_ */
