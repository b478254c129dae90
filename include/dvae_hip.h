/*
 * C ABI of libdvae_hip.so: the gfx950 (MI355X) kernels behind the video-VAE training and inference
 * hot path.  Plain pointers and sizes only; no torch types.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp32 data unless said otherwise; the library borrows it for
 *    the duration of the call, enqueues work on `stream` and returns without synchronising;
 *  - nothing here allocates, frees or synchronises: scratch is passed in (`ws` arguments) and its
 *    size in floats is given by the matching *_ws_floats() helper;
 *  - return value 0 on success, -EINVAL for a shape/argument the kernels do not support (nothing
 *    launched), -EIO if a launch failed; dvae_last_error() returns the message of the last failure
 *    on the calling thread;
 *  - activations are NHWC with frames folded into the leading dimension (rows = frame*H*W, channels
 *    contiguous); conv weights arrive in packed layouts produced by the dvae_pack_* calls.
 *
 * The reference has no native interface: its boundary is the Python surface cited per entry point
 * (paths relative to the reference tree).
 */
#ifndef DVAE_HIP_H
#define DVAE_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dvae_stream_t; /* hipStream_t */

#define DVAE_ACT_NONE 0
#define DVAE_ACT_LRELU 1   /* LeakyReLU(0.2) */
#define DVAE_ACT_SIGMOID 2

int dvae_version(void);
const char* dvae_last_error(void);

/* ---- operand precision of the MFMA kernels (dvae_gemm_*, dvae_conv4s2_fwd, dvae_deconv4s2_fwd,
 *      dvae_conv4s2_wgrad with multi-channel operands).  Process-wide, read when a kernel is launched.
 *      F32: v_mfma_f32_32x32x2_f32, every product exact (the default; what the 1e-4 parity contract
 *      is stated for).  BF16: tensors stay fp32 in memory, operands are rounded to bf16 (nearest
 *      even) as they are staged into LDS and multiplied by v_mfma_f32_32x32x16_bf16 with fp32
 *      accumulation.  F32_SPLIT (the Linear-layer GEMMs only; every other kernel runs as in F32):
 *      each fp32 operand is split into three bf16 terms and the six largest of the nine cross
 *      products are accumulated in fp32 by v_mfma_f32_32x32x16_bf16 -- the dropped terms are below
 *      2^-24 of the product, fp32's own rounding -- at 6/16 of the fp32 matrix time.  The reference
 *      has one precision (fp32, torch defaults); this switch has no counterpart there. */
#define DVAE_PRECISION_F32 0
#define DVAE_PRECISION_BF16 1
#define DVAE_PRECISION_F32_SPLIT 2
int dvae_set_mfma_precision(int precision);
int dvae_get_mfma_precision(void);

/* ---- content / motion split: video.mean(dim=1) (models/encoders.py:251) and
 *      video[:,1:]-video[:,:-1] (models/encoders.py:302-303) in one pass.
 *      video [B,T,P], mean [B,P], diff [B,T-1,P]; P % 4 == 0. */
int dvae_split_mean_diff_f32(const float* video, float* mean, float* diff, int64_t B, int64_t T,
                             int64_t P, dvae_stream_t stream);

/* ---- Moving-MNIST frame rendering (data/moving_mnist.py:130-153, the `_render_frame` max-composite):
 *      bank [bank_size][digit_size][digit_size] digit images; digit_idx [B][num_digits] int32 indices into
 *      the bank (the caller guarantees 0 <= idx < bank_size); positions [B][T][num_digits][2] int32 (x, y)
 *      of each digit's top-left corner per frame, already truncated to integers by the host-side
 *      trajectory simulation; video [B][T][frame_size][frame_size] out.  Pixels outside the canvas are
 *      clipped as in the reference. */
int dvae_render_clips_f32(const float* bank, int64_t bank_size, const int32_t* digit_idx,
                          const int32_t* positions, float* video, int64_t B, int64_t T, int64_t num_digits,
                          int64_t frame_size, int64_t digit_size, dvae_stream_t stream);

/* ---- reconstruction loss (models/disentangled_vae.py:246-253,263-268; models/lstm_vae.py:182-189;
 *      models/baseline_vae.py:222-229).  kind 0 = MSE, 1 = BCE with log terms clamped at -100.
 *      out[0] = mean loss per pixel, out[1] = out[0]*T*P, out[2] = temporal-difference MSE * (T-1)*P
 *      (0 when with_temporal == 0).  ws: dvae_recon_loss_ws_floats(B,T,P) floats. */
int64_t dvae_recon_loss_ws_floats(int64_t B, int64_t T, int64_t P);
int dvae_recon_loss_fwd_f32(const float* recon, const float* target, float* out, float* ws, int64_t B,
                            int64_t T, int64_t P, int kind, int with_temporal, dvae_stream_t stream);
/* d(out[1] + temporal_weight*out[2]) / d recon, times the device scalar *gscale. */
int dvae_recon_loss_bwd_f32(const float* recon, const float* target, const float* gscale, float* drecon,
                            int64_t B, int64_t T, int64_t P, int kind, float temporal_weight,
                            dvae_stream_t stream);
/* The same gradient taken through the sigmoid that produced recon (models/decoders.py:67-68: the decoder's
 * output activation): dlogit = [d.../d recon] * recon * (1 - recon), written in one pass instead of a gradient pass
 * followed by a sigmoid-backward pass over it, and *dbias = sum(dlogit), the gradient of the final transposed
 * convolution's bias (dbias may be NULL).  ws: dvae_recon_loss_bwd_logit_ws_floats(B,P) floats, 8-byte aligned. */
int64_t dvae_recon_loss_bwd_logit_ws_floats(int64_t B, int64_t P);
int dvae_recon_loss_bwd_logit_f32(const float* recon, const float* target, const float* gscale, float* dlogit,
                                  float* dbias, float* ws, int64_t B, int64_t T, int64_t P, int kind,
                                  float temporal_weight, dvae_stream_t stream);

/* ---- reconstruction metrics of the evaluation loop in one pass (evaluation/metrics.py:14-22 mse and
 *      psnr, :48-50 temporal_consistency; called per batch from reconstruction_metrics, :80-83).
 *      out[0] = mean squared error, out[1] = 10*log10(max_val^2 / out[0]) (+inf when out[0] == 0),
 *      out[2] = mean |recon[:,1:] - recon[:,:-1]| (NaN when T == 1, like torch's mean of no elements).  recon, target: [B][T][P] with
 *      P % 4 == 0; ws: dvae_recon_loss_ws_floats(B,T,P) floats, 8-byte aligned. */
int dvae_recon_metrics_f32(const float* recon, const float* target, float* out, float* ws, int64_t B,
                           int64_t T, int64_t P, float max_val, dvae_stream_t stream);

/* ---- dvae_recon_metrics_f32 for callers that cannot allocate device memory (code under HIP graph
 *      capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the reference.
 *      Same arguments minus the workspace, same kernels, same results bit for bit: the per-workgroup
 *      partial sums go to a 32 KiB array that is part of the library's device image (each GPU the
 *      library runs on holds its own copy until the process ends), located once per device at the first
 *      call on that device.  The call allocates and frees nothing.  It must be made with the device of
 *      the pointers current.  Limit: B*T*P <= DVAE_NOALLOC_MAX_ELEMENTS elements per clip tensor
 *      (-EINVAL beyond; dvae_recon_metrics_f32 has no such limit).  All calls on one GPU share that
 *      GPU's array: calls that overlap in time on one GPU (different streams, different threads) race;
 *      calls ordered on one stream do not.  Nothing needs clearing between calls. */
#define DVAE_NOALLOC_MAX_ELEMENTS ((int64_t)1 << 31)
int64_t dvae_recon_metrics_noalloc_max_elements(void);
/* Locates the array on the current device now instead of at the first call there.  A caller that will issue
 * dvae_recon_metrics_noalloc_f32 under HIP graph capture calls this once per device beforehand: the look-up is a
 * runtime call (it may load the library's code object) that a capturing stream does not allow.  No counterpart in
 * the reference. */
int dvae_recon_metrics_noalloc_init(void);
int dvae_recon_metrics_noalloc_f32(const float* recon, const float* target, float* out, int64_t B, int64_t T,
                                   int64_t P, float max_val, dvae_stream_t stream);

/* ---- reparameterisation (models/encoders.py:207-214): z = mu + eps*exp(0.5*logvar), n elements. */
int dvae_reparam_fwd_f32(const float* mu, const float* logvar, const float* eps, float* z, int64_t n,
                         dvae_stream_t stream);
int dvae_reparam_bwd_f32(const float* dz, const float* logvar, const float* eps, float* dmu,
                         float* dlogvar, int64_t n, dvae_stream_t stream);

/* ---- KL term (models/disentangled_vae.py:255-259; free-bits floor of models/lstm_vae.py:191-200).
 *      mu, logvar [B,D].  out[0] = mean_b sum_d max(kl_bd, free_bits) (no floor when free_bits <= 0),
 *      out[1] = the same without the floor, out[2] = mean over all B*D of the floored values. */
int dvae_kl_fwd_f32(const float* mu, const float* logvar, float* out, int64_t B, int64_t D,
                    float free_bits, dvae_stream_t stream);
/* dmu, dlogvar of coef * (*gscale) * out[0].  coef_dev (optional): a device scalar read by the kernel in place of
 *      coef -- the annealed beta of training/trainer.py:115-149 changes per epoch, and a step captured into a
 *      hipGraph must not hold it as a launch constant. */
int dvae_kl_bwd_f32(const float* mu, const float* logvar, const float* gscale, float coef, const float* coef_dev,
                    float* dmu, float* dlogvar, int64_t B, int64_t D, float free_bits, dvae_stream_t stream);

/* ---- the whole objective of one compute_loss call (models/disentangled_vae.py:222-292, models/lstm_vae.py:167-220,
 *      models/baseline_vae.py:208-252) in two launches: the pass over the clips, then one workgroup that sums its
 *      partials, computes the KL terms and writes every entry of the loss dictionary.
 *      One latent (mu1 == NULL), out[7]: total = recon + beta0*kl + temporal_weight*temporal, recon per pixel,
 *      recon, temporal, kl (free-bits floor applied), kl raw, kl per dim.
 *      Two latents, out[12]: total = recon + kl + temporal_weight*temporal + indep_weight*indep, kl = beta0*kl0 +
 *      beta1*kl1, recon per pixel, recon, temporal, kl0 (mean, raw, per dim), kl1 (same), indep (*indep, or 0).
 *      total_out (optional): the total once more.  ws: dvae_recon_loss_ws_floats() floats, 8-byte aligned.
 *      beta_dev / beta_dev_mask: bit p set = latent p's weight is the device scalar *beta_dev instead of beta_p
 *      (the per-epoch annealed beta of training/trainer.py:185 that `beta_c = beta_content if beta_content is not
 *      None else beta`, models/disentangled_vae.py:243-244, falls back to); mask 0 = both are launch constants. */
int dvae_vae_loss_fwd_f32(const float* recon, const float* target, int64_t B, int64_t T, int64_t P, int kind,
                          int with_temporal, const float* mu0, const float* logvar0, int64_t D0, float beta0,
                          const float* mu1, const float* logvar1, int64_t D1, float beta1, const float* beta_dev,
                          int beta_dev_mask, float free_bits, float temporal_weight, const float* indep,
                          float indep_weight, float* out, float* total_out, float* ws, dvae_stream_t stream);
/* gradients of coef_i * KL_i (batch mean) wrt mu_i, logvar_i for one or two latents in one launch, times gscale[0];
 *      coef_dev / coef_dev_mask as beta_dev / beta_dev_mask above */
int dvae_kl_bwd2_f32(const float* mu0, const float* logvar0, float coef0, float* dmu0, float* dlogvar0, int64_t D0,
                     const float* mu1, const float* logvar1, float coef1, float* dmu1, float* dlogvar1, int64_t D1,
                     const float* coef_dev, int coef_dev_mask, const float* gscale, int64_t B, float free_bits,
                     dvae_stream_t stream);

/* ---- independence term (models/disentangled_vae.py:270-278): mean of the squared cross-correlation
 *      of the standardised latents (unbiased std + 1e-8, correlation / B).
 *      zc [B,Dc], zm [B,Dm]; out[0] = loss.  ws: Dc*Dm + 2*Dc + 2*Dm + Dc floats, reused by bwd. */
int64_t dvae_indep_ws_floats(int64_t Dc, int64_t Dm);
int dvae_indep_fwd_f32(const float* zc, const float* zm, float* out, float* ws, int64_t B, int64_t Dc,
                       int64_t Dm, dvae_stream_t stream);
int dvae_indep_bwd_f32(const float* zc, const float* zm, const float* ws, const float* gscale, float coef,
                       float* dzc, float* dzm, int64_t B, int64_t Dc, int64_t Dm, dvae_stream_t stream);

/* ---- BatchNorm2d pieces (models/encoders.py:35,79; models/decoders.py:44), channels-last [M,C].
 *      Forward statistics come out of the conv kernels as per-workgroup partial sums
 *      partials[nparts][2][C] (sum, sum of squares).  Finalize turns them into the folded affine
 *      scale = gamma*invstd, shift = beta - mean*scale that the NEXT kernel applies on load, keeps
 *      mean/invstd for backward and updates the running buffers (momentum 0.1, unbiased variance,
 *      num_batches_tracked += 1; pass NULL to skip). */
int dvae_bn_finalize_f32(const float* partials, int64_t nparts, int64_t M, int64_t C, const float* gamma,
                         const float* beta, float* running_mean, float* running_var,
                         int64_t* num_batches_tracked, float momentum, float eps, float* scale,
                         float* shift, float* mean, float* invstd, dvae_stream_t stream);
/* The same when the layer's batch is `replicas` identical copies of the M rows behind the partial sums (the
 * baseline model decodes one latent per clip and repeats the frame T times, models/baseline_vae.py:120-131: the
 * copies are computed once).  Mean and biased variance do not change; the running variance takes its unbiased
 * correction from the full count M*replicas, as nn.BatchNorm2d does on the repeated batch. */
int dvae_bn_finalize_rep_f32(const float* partials, int64_t nparts, int64_t M, int64_t replicas, int64_t C,
                             const float* gamma, const float* beta, float* running_mean, float* running_var,
                             int64_t* num_batches_tracked, float momentum, float eps, float* scale,
                             float* shift, float* mean, float* invstd, dvae_stream_t stream);
/* Eval mode: scale/shift from the running statistics. */
int dvae_bn_eval_affine_f32(const float* gamma, const float* beta, const float* running_mean,
                            const float* running_var, float eps, float* scale, float* shift, int64_t C,
                            dvae_stream_t stream);
/* out = act(x*scale[c]+shift[c]) materialised, [M,C]. */
int dvae_affine_act_f32(const float* x, const float* scale, const float* shift, int act, float* out,
                        int64_t M, int64_t C, dvae_stream_t stream);
/* Backward of BN + LeakyReLU given du = dL/d(post-activation), y = raw conv output.
 * Step 1 accumulates sum(dz) and sum(dz*xhat) per channel (dz = du*lrelu'(y*scale+shift)) and turns
 * them into dgamma, dbeta; step 2 overwrites du with dL/dy.  ws: dvae_colsum_ws_floats(M,C) floats.
 * tickets (here and in the calls below that take it; NULL allowed): dvae_colsum_tickets() 32-bit words of device
 * memory that are ZERO when the call is made and are zero again when its kernels have finished.  With them the
 * column sums take one launch instead of two (the workgroup of a column group that finishes last adds the group's
 * partial rows, in row order, so results do not depend on which one it is).  Calls that share the words must be
 * ordered on one stream. */
int64_t dvae_colsum_ws_floats(int64_t M, int64_t C);
int64_t dvae_colsum_tickets(void);
int dvae_bn_bwd_f32(float* du, const float* y, const float* scale, const float* shift, const float* mean,
                    const float* invstd, const float* gamma, float* dgamma, float* dbeta, float* ws,
                    unsigned* tickets, int64_t M, int64_t C, dvae_stream_t stream);
/* The same with step 1 done elsewhere: partials[nparts][2][C] rows of (sum dz, sum dz*xhat) as written by
 * dvae_conv4s2_dgrad_bnstat_f32 / dvae_deconv4s2_dgrad_bnstat_f32 while they produced du. */
/* Step 1's finish alone, for consumers that apply step 2 themselves as they load du (see
 * dvae_conv4s2_dgrad_bnstat_bnin_f32): dgamma, dbeta from the partial rows, and the vectors A, B with
 * dL/dy = scale*dz + A*y + B  (A = -scale*invstd*dgamma/M, B = -scale*dbeta/M - A*mean). */
int dvae_bn_bwd_coeffs_f32(const float* partials, int64_t nparts, const float* scale, const float* mean,
                           const float* invstd, float* dgamma, float* dbeta, float* coef_a, float* coef_b,
                           int64_t M, int64_t C, dvae_stream_t stream);
/* Step 1 alone: partials [dvae_bn_bwd_reduce_rows(M,C)][2][C], the rows dvae_bn_bwd_coeffs_f32 takes. */
int64_t dvae_bn_bwd_reduce_rows(int64_t M, int64_t C);
int dvae_bn_bwd_reduce_f32(const float* du, const float* y, const float* scale, const float* shift,
                           const float* mean, const float* invstd, float* partials, int64_t M, int64_t C,
                           dvae_stream_t stream);
/* dvae_bn_bwd_reduce_f32 followed by dvae_bn_bwd_coeffs_f32, in one launch (tickets required). */
int dvae_bn_bwd_reduce_coeffs_f32(const float* du, const float* y, const float* scale, const float* shift,
                                  const float* mean, const float* invstd, float* dgamma, float* dbeta, float* coef_a,
                                  float* coef_b, float* ws, unsigned* tickets, int64_t M, int64_t C,
                                  dvae_stream_t stream);
int dvae_bn_bwd_from_partials_f32(float* du, const float* y, const float* scale, const float* shift,
                                  const float* mean, const float* invstd, const float* partials, int64_t nparts,
                                  float* dgamma, float* dbeta, int64_t M, int64_t C, dvae_stream_t stream);
/* Backward of (bias +) activation of a Linear layer: dpre = dout * act'(out) written in place over
 * dout (act NONE leaves it untouched), dbias[n] = sum_m dpre[m][n] (NULL to skip). [M,N], N%4==0. */
int dvae_bias_act_bwd_f32(float* dout, const float* out, int act, float* dbias, float* ws, unsigned* tickets,
                          int64_t M, int64_t N, dvae_stream_t stream);
/* The same with dpre written to its own [M,N] buffer and dout left untouched (autograd hands a node gradients it
 * must not modify: torch's LeakyReLU/Linear backward, models/encoders.py:40-43, allocate their results too);
 * dpre == dout is the in-place form above. */
int dvae_bias_act_bwd_to_f32(const float* dout, float* dpre, const float* out, int act, float* dbias, float* ws,
                             unsigned* tickets, int64_t M, int64_t N, dvae_stream_t stream);
/* dlogit = drecon * r * (1-r) (sigmoid backward), dbias[0] = sum(dlogit); n elements. */
int dvae_sigmoid_bwd_f32(const float* drecon, const float* recon, float* dlogit, float* dbias, float* ws,
                         int64_t n, dvae_stream_t stream);

/* ---- layout helpers.  permute3: out = in.permute(p0,p1,p2) of a contiguous [D0,D1,D2] tensor (used to
 *      move Linear weights between the reference's NCHW flatten order, models/encoders.py:51 and
 *      models/decoders.py:63, and the kernels' NHWC order, and sequences between batch-major and
 *      time-major).  weighted_sum: out[0] = sum_i coefs[i] * terms[i][0] over n <= 8 device scalars
 *      (terms: host array of device pointers, coefs: host array), the total loss of
 *      models/disentangled_vae.py:280. */
int dvae_permute3_f32(const float* in, float* out, int64_t D0, int64_t D1, int64_t D2, int p0, int p1,
                      int p2, dvae_stream_t stream);
int dvae_weighted_sum_f32(int n, const float* const* terms, const float* coefs, float* out,
                          dvae_stream_t stream);

/* ---- Linear layers (nn.Linear at models/encoders.py:43-46,87-90,197-198; models/decoders.py:30-33,
 *      91-92,102-105,218-219,228-231) as f32-MFMA GEMMs, row-major, K%4==0 / N%4==0 as noted.
 *      ws: dvae_gemm_ws_floats(M,N,K) floats (split-K slabs). */
int64_t dvae_gemm_ws_floats(int64_t M, int64_t N, int64_t K);
/* C[M,N] = act(A[M,K] * B[N,K]^T + bias[N]); lda/ldb/ldc row strides in floats; K%4==0. */
int dvae_gemm_nt_f32(const float* A, int64_t lda, const float* B, int64_t ldb, const float* bias, int act,
                     float* C, int64_t ldc, int64_t M, int64_t N, int64_t K, float* ws,
                     dvae_stream_t stream);
/* C[M,N] (+)= A[M,K] * B[K,N]; K%4==0, N%4==0; accumulate != 0 adds into C. */
int dvae_gemm_nn_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                     int64_t M, int64_t N, int64_t K, int accumulate, float* ws, dvae_stream_t stream);
/* C[M,N] (+)= A[K,M]^T * B[K,N]; M%4==0, N%4==0 (weight gradients: K is the batch dimension). */
int dvae_gemm_tn_f32(const float* A, int64_t lda, const float* B, int64_t ldb, float* C, int64_t ldc,
                     int64_t M, int64_t N, int64_t K, int accumulate, float* ws, dvae_stream_t stream);

/* ---- 4x4 stride-2 pad-1 convolution family (nn.Conv2d at models/encoders.py:32,76 and
 *      nn.ConvTranspose2d at models/decoders.py:40,49) as implicit GEMMs on f32 MFMA.
 *      x [N,H,W,Cin] NHWC; in_scale/in_shift/in_act: affine + activation applied to x as it is
 *      loaded (the previous layer's folded BatchNorm; NULL scale = identity);  zero padding is applied
 *      AFTER that transform.  stat_partials (NULL to skip) receives per-workgroup column sums and
 *      sums of squares of the raw output, dvae_conv_stat_parts(...) rows of [2][Cout]. */
/* weight packs: torch Conv2d weight [A][Bc][4][4] -> [A][kh][kw][Bc]  */
int dvae_pack_conv_w_f32(const float* w, float* wp, int64_t A, int64_t Bc, dvae_stream_t stream);
/* torch weight [Kc][Nc][4][4] -> [phase=ph*2+pw][Nc][th][tw][Kc], the four 2x2 sub-pixel filters of
 * the transposed convolution (kh = 1,3 for even output rows, 0,2 for odd ones). */
int dvae_pack_deconv_w_f32(const float* w, float* wp, int64_t Kc, int64_t Nc, dvae_stream_t stream);

/* n weights packed in ONE launch (per 16): item i is dvae_pack_conv_w_f32(w[i], wp[i], d0[i], d1[i]) when
 * transposed[i] == 0, dvae_pack_deconv_w_f32(w[i], wp[i], d0[i], d1[i]) otherwise.  The arrays are host arrays,
 * read during the call.  The reference packs nothing (torch keeps OIHW weights, models/encoders.py:33-38,
 * models/decoders.py:36-42); one call per encoder/decoder forward replaces its per-layer calls. */
int dvae_pack_w_many_f32(int n, const float* const* w, float* const* wp, const int64_t* d0, const int64_t* d1,
                         const int* transposed, dvae_stream_t stream);
int64_t dvae_conv_stat_parts(int64_t M_rows);
/* exact number of partial rows a forward launch with that many GEMM rows per phase writes
 * (phases = 1 for the convolution, 4 for the transposed convolution) */
int64_t dvae_conv_stat_parts_written(int64_t M_rows_per_phase, int phases);
/* y[N,H/2,W/2,Cout] = conv(act(x)) with wp = packed [Cout][4][4][Cin]; Cin == 1 or Cin%4 == 0. */
int dvae_conv4s2_fwd_f32(const float* x, const float* in_scale, const float* in_shift, int in_act,
                         const float* wp, float* y, float* stat_partials, int64_t N, int64_t H, int64_t W,
                         int64_t Cin, int64_t Cout, dvae_stream_t stream);
/* y[N,2H,2W,Cout] = conv_transpose(act(x)) with wp = dvae_pack_deconv_w_f32 output
 * (Kc = Cin, Nc = Cout); Cin%4 == 0.  Also the data gradient of dvae_conv4s2_fwd_f32. */
int dvae_deconv4s2_fwd_f32(const float* x, const float* in_scale, const float* in_shift, int in_act,
                           const float* wp, float* y, float* stat_partials, int64_t N, int64_t H,
                           int64_t W, int64_t Cin, int64_t Cout, dvae_stream_t stream);
/* The same two with scratch memory from the caller: layers with few row tiles (small batches, the late encoder
 * layers) then share their K loop out over several workgroups per tile, which write partial outputs to the scratch
 * and are summed in a fixed order by a second launch (which also takes the BatchNorm partial sums).  Results are
 * those of the plain entry points up to the order of fp32 additions.  ws: dvae_conv_fwd_ws_floats(...) floats,
 * 16-byte aligned; a smaller (or NULL) scratch turns the split off. */
int64_t dvae_conv_fwd_ws_floats(int transposed, int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t Cout);
int dvae_conv4s2_fwd_ws_f32(const float* x, const float* in_scale, const float* in_shift, int in_act,
                            const float* wp, float* y, float* stat_partials, int64_t N, int64_t H, int64_t W,
                            int64_t Cin, int64_t Cout, float* ws, int64_t ws_floats, int storage,
                            dvae_stream_t stream);
int dvae_deconv4s2_fwd_ws_f32(const float* x, const float* in_scale, const float* in_shift, int in_act,
                              const float* wp, float* y, float* stat_partials, int64_t N, int64_t H, int64_t W,
                              int64_t Cin, int64_t Cout, float* ws, int64_t ws_floats, int storage,
                              dvae_stream_t stream);
/* `storage` (here and in the calls below that take it; 0 = every tensor holds floats): in the bf16-MFMA mode the
 * large few-channel activations and their gradients may be kept in HBM as bfloat16, which halves the bytes of the
 * HBM-bound kernels around them.  A set bit says that the tensor it names holds bfloat16 elements (same shape and
 * element order; its pointer keeps the float type of the signature): DVAE_ST_X the input x (weight gradients: P),
 * DVAE_ST_Y the output y (weight gradients: Q), DVAE_ST_BNY bn_y (p_y), DVAE_ST_INY in_y, DVAE_ST_DX dx_out.
 * Only in bf16 mode, and only the combinations that a training step produces have kernels, -EINVAL otherwise.
 * Decoder's last block (64 -> 32 channels on 16 x 16, then 32 -> 1; its raw output, du and dL/dy are bf16):
 *   dvae_deconv4s2_fwd_ws_f32           64 -> 32 row kernel, no fused statistics of a layer below:  DVAE_ST_Y
 *   dvae_deconv4s2_final_fwd_f32        Cin = 32:                                                   DVAE_ST_X
 *   dvae_conv4s2_wgrad_f32              single-channel Q, C1 = 32:                                  DVAE_ST_X
 *                                       64 x 32 row kernel with a transformed P:                    DVAE_ST_Y
 *   dvae_conv4s2_dgrad_bnstat_f32       Cin = 1, Cout = 32:                               DVAE_ST_Y | DVAE_ST_BNY
 *   dvae_conv4s2_dgrad_bnstat_bnin_f32  32 -> 64 row kernel:                   DVAE_ST_X | DVAE_ST_INY | DVAE_ST_DX
 * ... together with the block before it (128 -> 64 channels on 8 x 8; its raw output and du are bf16):
 *   dvae_deconv4s2_fwd_ws_f32           128 -> 64 row kernel, load transform, no fused statistics:  DVAE_ST_Y
 *                                       64 -> 32 row kernel, load transform:                   DVAE_ST_X | DVAE_ST_Y
 *   dvae_conv4s2_wgrad_f32              64 x 32 row kernel with a transformed P:                DVAE_ST_X | DVAE_ST_Y
 *   dvae_conv4s2_dgrad_bnstat_bnin_f32  32 -> 64 row kernel: DVAE_ST_X | DVAE_ST_INY | DVAE_ST_DX | DVAE_ST_BNY | DVAE_ST_Y
 *   dvae_conv4s2_wgrad_bnq_f32          128 x 64 row kernel (Q = du: DVAE_ST_Y, q_y: DVAE_ST_INY):  DVAE_ST_Y | DVAE_ST_INY
 * Few-channel encoder's first block (1 -> 16 channels on 64 x 64, then 16 -> 32; its raw output and du are bf16):
 *   dvae_conv4s2_fwd_ws_f32             Cin = 1, Cout <= 16, no fused statistics:                   DVAE_ST_Y
 *                                       16 -> 32 row kernel with a load transform:                  DVAE_ST_X
 *   dvae_conv4s2_wgrad_f32              32 x 16 row kernel with a transformed Q:                    DVAE_ST_Y
 *   dvae_deconv4s2_dgrad_bnstat_bnin_f32  32 -> 16 row kernel:                            DVAE_ST_Y | DVAE_ST_BNY
 *   dvae_conv4s2_wgrad_bnp_f32          C1 = 16:                                          DVAE_ST_X | DVAE_ST_BNY
 * What a kernel writes as bfloat16 is the round-to-nearest-even of the fp32 value it would have written; statistics
 * are always reduced from the fp32 accumulators, before rounding.  No counterpart in the reference. */
#define DVAE_ST_X 1
#define DVAE_ST_Y 2
#define DVAE_ST_BNY 4
#define DVAE_ST_INY 8
#define DVAE_ST_DX 16
/* The two kernels above used as data gradients inside a conv -> BatchNorm -> LeakyReLU stack (autograd of
 * models/encoders.py:32-36 and models/decoders.py:40-45): x is dL/d(raw output of the next layer), y
 * receives du = dL/d(activation of this layer), unchanged; in the same pass the kernel also reduces what
 * this layer's BatchNorm backward needs, reading the layer's raw output bn_y (shaped like y) and its
 * folded forward statistics: stat_partials rows hold sum(dz) and sum(dz*xhat) per channel, with
 * dz = du*lrelu'(bn_y*bn_scale + bn_shift), xhat = (bn_y - bn_mean)*bn_invstd.  Feed them to
 * dvae_bn_bwd_from_partials_f32.  No load-time transform on x. */
int dvae_conv4s2_dgrad_bnstat_f32(const float* x, const float* wp, float* y, float* stat_partials, int64_t N,
                                  int64_t H, int64_t W, int64_t Cin, int64_t Cout, const float* bn_y,
                                  const float* bn_scale, const float* bn_shift, const float* bn_mean,
                                  const float* bn_invstd, int storage, dvae_stream_t stream);
/* dvae_conv4s2_dgrad_bnstat_f32 whose input has not been through its own BatchNorm backward yet (the
 * decoder's conv-transpose -> BatchNorm -> LeakyReLU blocks, models/decoders.py:40-45, walked backwards):
 * du_in is dL/d(activation) of the layer above, in_y that layer's raw output, in_scale/in_shift its
 * folded forward statistics and in_a/in_b the vectors of dvae_bn_bwd_coeffs_f32.  The kernel forms
 * dL/d(in_y) row by row as it loads, uses it as the convolution input and also writes it to dx_out
 * (shaped like du_in) for the weight-gradient call; the separate elementwise pass of
 * dvae_bn_bwd_from_partials_f32 is not needed.  Only where dvae_conv4s2_bnin_supported says so
 * (few-channel layers in fp32 mode); otherwise -EINVAL. */
int dvae_conv4s2_bnin_supported(int64_t H, int64_t W, int64_t Cin, int64_t Cout);
int dvae_conv4s2_dgrad_bnstat_bnin_f32(const float* du_in, const float* in_y, const float* in_scale,
                                       const float* in_shift, const float* in_a, const float* in_b,
                                       float* dx_out, const float* wp, float* y, float* stat_partials,
                                       int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                                       const float* bn_y, const float* bn_scale, const float* bn_shift,
                                       const float* bn_mean, const float* bn_invstd, int storage,
                                       dvae_stream_t stream);
/* The transposed-conv counterpart (the encoder's conv -> BatchNorm -> LeakyReLU blocks,
 * models/encoders.py:32-36, walked backwards); same arguments and rules. */
int dvae_deconv4s2_bnin_supported(int64_t H, int64_t W, int64_t Cin, int64_t Cout);
int dvae_deconv4s2_dgrad_bnstat_bnin_f32(const float* du_in, const float* in_y, const float* in_scale,
                                         const float* in_shift, const float* in_a, const float* in_b,
                                         float* dx_out, const float* wp, float* y, float* stat_partials,
                                         int64_t N, int64_t H, int64_t W, int64_t Cin, int64_t Cout,
                                         const float* bn_y, const float* bn_scale, const float* bn_shift,
                                         const float* bn_mean, const float* bn_invstd, int storage,
                                         dvae_stream_t stream);
int dvae_deconv4s2_dgrad_bnstat_f32(const float* x, const float* wp, float* y, float* stat_partials, int64_t N,
                                    int64_t H, int64_t W, int64_t Cin, int64_t Cout, const float* bn_y,
                                    const float* bn_scale, const float* bn_shift, const float* bn_mean,
                                    const float* bn_invstd, dvae_stream_t stream);
/* Weight gradient shared by both layer kinds:
 *   dw[c1][c2][kh][kw] = sum_{n,a,b} P[n,a,b,c1] * Q[n,2a-1+kh,2b-1+kw,c2]
 * P [N,Hs,Ws,C1] is the small-resolution tensor, Q [N,2Hs,2Ws,C2] the large one (Conv2d: P = dY,
 * Q = layer input; ConvTranspose2d: P = layer input, Q = dY), each with its own load-time transform.
 * dw is in the torch layout [C1][C2][4][4].  C1%4==0; C2 == 1 or C2%4 == 0.
 * ws: dvae_conv_wgrad_ws_floats(N,Hs,Ws,C1,C2). */
int64_t dvae_conv_wgrad_ws_floats(int64_t N, int64_t Hs, int64_t Ws, int64_t C1, int64_t C2);
int dvae_conv4s2_wgrad_f32(const float* P, const float* p_scale, const float* p_shift, int p_act,
                           const float* Q, const float* q_scale, const float* q_shift, int q_act,
                           float* dw, float* ws, int64_t N, int64_t Hs, int64_t Ws, int64_t C1, int64_t C2,
                           int storage, dvae_stream_t stream);
/* Weight gradient of the first encoder convolution (single-channel Q = the frames, models/encoders.py:32)
 * whose P = dL/d(activation) has not been through that layer's BatchNorm backward: du, the layer's raw
 * output p_y, its forward affine and the vectors of dvae_bn_bwd_coeffs_f32; P becomes scale*dz + A*y + B
 * as it is loaded, so the first layer needs no elementwise BatchNorm-backward pass at all.  ws as for
 * dvae_conv4s2_wgrad_f32 with C2 = 1. */
int dvae_conv4s2_wgrad_bnp_f32(const float* du, const float* p_y, const float* p_scale, const float* p_shift,
                               const float* p_a, const float* p_b, const float* Q, float* dw, float* ws,
                               int64_t N, int64_t Hs, int64_t Ws, int64_t C1, int storage, dvae_stream_t stream);
/* dvae_conv4s2_wgrad_f32 whose Q = dL/d(activation) has not been through its layer's BatchNorm backward
 * (weight gradient of a decoder block, models/decoders.py:40-45): du, the layer's raw output q_y, its
 * forward affine and the vectors of dvae_bn_bwd_coeffs_f32.  Q becomes scale*dz + A*y + B as it is
 * loaded, and the finished dL/dy is also written to dx_out (shaped like du) for the data-gradient call
 * that follows.  Only where dvae_conv4s2_wgrad_bnq_supported says so; otherwise -EINVAL. */
int dvae_conv4s2_wgrad_bnq_supported(int64_t N, int64_t Hs, int64_t Ws, int64_t C1, int64_t C2);
int dvae_conv4s2_wgrad_bnq_f32(const float* P, const float* p_scale, const float* p_shift, int p_act,
                               const float* du, const float* q_y, const float* q_scale, const float* q_shift,
                               const float* q_a, const float* q_b, float* dx_out, float* dw, float* ws,
                               int64_t N, int64_t Hs, int64_t Ws, int64_t C1, int64_t C2, int storage,
                               dvae_stream_t stream);
/* Final decoder layer (models/decoders.py:49-53,67-68): ConvTranspose2d(Cin -> 1, bias) + sigmoid.
 * x [N,H,W,Cin] (+ load-time transform), w torch layout [Cin][1][4][4], y [N,2H,2W]. Cin%4==0. */
int dvae_deconv4s2_final_fwd_f32(const float* x, const float* in_scale, const float* in_shift, int in_act,
                                 const float* w, const float* bias, int out_act, float* y, int64_t N,
                                 int64_t H, int64_t W, int64_t Cin, int storage, dvae_stream_t stream);

/* ---- LSTM (nn.LSTM at models/encoders.py:145-152, models/decoders.py:94-100,221-226), gate order
 *      i,f,g,o.  One call advances one layer/direction set by one time step for all B rows:
 *        pre = [xproj] + [x * w_ih^T] + h_prev * w_hh^T + [b_ih + b_hh]
 *      gates_out [B,4H] receives the activated gates (kept for backward), h_out/c_out the new state.
 *      Every row-pointer has its own row stride (floats) so batch-first sequences can be addressed in
 *      place.  H%8==0, I%4==0. */
int dvae_lstm_step_fwd_f32(const float* xproj, int64_t ld_xproj, const float* x, int64_t ld_x,
                           const float* w_ih, int64_t I, const float* h_prev, int64_t ld_h,
                           const float* c_prev, int64_t ld_c, const float* w_hh, const float* b_ih,
                           const float* b_hh, float* gates_out, float* h_out, int64_t ld_hout,
                           float* c_out, int64_t ld_cout, int64_t B, int64_t H, dvae_stream_t stream);
/* Pointwise part of one backward step: from dh (total gradient reaching h_t), dc (carried), the saved
 * activated gates, c_prev and c: dgates [B,4H] (pre-activation gradients) and dc_prev. */
int dvae_lstm_step_bwd_pointwise_f32(const float* dh, int64_t ld_dh, const float* dc, int64_t ld_dc,
                                     const float* gates, const float* c_prev, int64_t ld_cprev,
                                     const float* c, int64_t ld_c, float* dgates, float* dc_prev,
                                     int64_t ld_dcprev, int64_t B, int64_t H, dvae_stream_t stream);
/* One whole backward step of the recurrence in a launch: dh_prev [B,H] += dg_t [B,K] * wt [H,K]^T
 * (K = 4H; wt is W_hh transposed, or (W_ih + W_hh)^T for a single-layer decoder), and every finished
 * element goes straight through the pointwise backward of step t-1 as in
 * dvae_lstm_step_bwd_pointwise_f32 (dc, gates, c_prev, c of step t-1 -> dgates, dc_prev). */
int dvae_lstm_bwd_step_f32(const float* dg_t, int64_t ld_dg, const float* wt, int64_t ld_wt, float* dh_prev,
                           int64_t ld_dh, const float* dc, int64_t ld_dc, const float* gates,
                           const float* c_prev, int64_t ld_cprev, const float* c, int64_t ld_c,
                           float* dgates, float* dc_prev, int64_t ld_dcprev, int64_t B, int64_t H,
                           int64_t K, dvae_stream_t stream);

/* the same step for a stacked decoder, where two gradients reach one hidden state (recurrent path of the layer's
 * next step and input path of the layer above, or of layer 0's next step for the top layer):
 * dh_prev += dg_a * wt_a^T + dg_b * wt_b^T (dg_b / wt_b may be NULL), then the pointwise backward as above. */
int dvae_lstm_bwd_step2_f32(const float* dg_a, int64_t ld_dga, const float* wt_a, int64_t ld_wta, int64_t Ka,
                            const float* dg_b, int64_t ld_dgb, const float* wt_b, int64_t ld_wtb, int64_t Kb,
                            float* dh_prev, int64_t ld_dh, const float* dc, int64_t ld_dc, const float* gates,
                            const float* c_prev, int64_t ld_cprev, const float* c, int64_t ld_c, float* dgates,
                            float* dc_prev, int64_t ld_dcprev, int64_t B, int64_t H, dvae_stream_t stream);
/* Skinny GEMM of the recurrence's backward step, dh_{t-1} (+)= dG_t * W with W handed over transposed:
 * out[B,N] (+)= a[B,K] * wt[N,K]^T, every row with its own stride (floats).  Latency-oriented: 16x16
 * output tiles, K split over the waves of a workgroup, operands loaded straight into MFMA registers.
 * K%4==0, ld_a%4==0, ld_wt%4==0, a and wt 16-byte aligned. */
int dvae_skinny_gemm_nt_f32(const float* a, int64_t ld_a, const float* wt, int64_t ld_wt, float* out,
                            int64_t ld_out, int64_t B, int64_t N, int64_t K, int accumulate,
                            dvae_stream_t stream);

/* ---- the whole recurrence of one LSTM layer in ONE launch (nn.LSTM over a sequence, models/encoders.py:145-161;
 *      the autoregressive decoder loop, models/decoders.py:129-132,264-267, from its second step on, where a
 *      single layer's input is its own previous output and x W_ih^T + h W_hh^T = h (W_ih + W_hh)^T).
 *      Up to two independent chains per launch (the two directions of a bidirectional layer); host arrays
 *      of nchains device pointers.  Per chain and step t (walked T-1..0 when reverse[c] != 0):
 *        pre = [xproj[t]] + h_prev * w_rec^T + [b_a + b_b];   gates/c/h as in dvae_lstm_step_fwd_f32
 *      xproj [T][B][4H] (NULL array or entry: none), w_rec [4H][H], h_init / c_init [B] rows with strides
 *      ld_hinit / ld_cinit (NULL: zero state), outputs gates [T][B][4H], cseq [T][B][H] and h of step t at
 *      hseq + t*step_h + b*ld_h.  The recurrent weights stay in registers for all T steps: a group of 32
 *      batch rows is served by H/8 workgroups that exchange the hidden state through ws inside the launch.
 *      All workgroups must be resident together, so dvae_lstm_seq_supported() admits H in {128, 256, 512}
 *      and nchains*ceil(B/32)*(H/8) <= dvae_lstm_resident_workgroups(0, H, 2) only (512 on an MI355X with all
 *      256 compute units; 0 without a GPU); otherwise -EINVAL (use the per-step calls).
 *      ws: dvae_lstm_seq_ws_floats(nchains, B, H) floats, 256-byte aligned; ws[0] (read as unsigned) is
 *      non-zero after the launch if a wait inside it ran into its 200 ms bound.  The outputs are then invalid,
 *      and say so: a workgroup that gives up fills the part of hseq / cseq (backward: dgates, dh_init, dc_init)
 *      it has not produced yet with NaN.
 *      Not to be run concurrently with another launch of itself unless both fit the chip together. */
int dvae_lstm_seq_supported(int64_t nchains, int64_t T, int64_t B, int64_t H);
/* How many workgroups of the one-launch recurrence kernels (kind 0: dvae_lstm_seq_fwd/bwd_f32, kind 1:
 * dvae_lstm_dec2_fwd/bwd_f32) for hidden size H can be resident at once on the current device: its compute-unit
 * count x the occupancy the runtime reports for the compiled kernels (the smaller of forward and backward), at most
 * max_per_cu per compute unit when max_per_cu > 0 (callers that share the device with other kernels, e.g. RCCL's,
 * leave room that way).  No counterpart in the reference. */
int64_t dvae_lstm_resident_workgroups(int kind, int64_t H, int64_t max_per_cu);
int64_t dvae_lstm_seq_ws_floats(int64_t nchains, int64_t B, int64_t H);
int dvae_lstm_seq_fwd_f32(int64_t nchains, const float* const* xproj, const float* const* w_rec,
                          const float* const* b_a, const float* const* b_b, const float* const* h_init,
                          int64_t ld_hinit, const float* const* c_init, int64_t ld_cinit, float* const* gates,
                          float* const* hseq, int64_t ld_h, int64_t step_h, float* const* cseq, const int* reverse,
                          int64_t T, int64_t B, int64_t H, float* ws, dvae_stream_t stream);

/* ---- backpropagation through time of the same recurrences in ONE launch (autograd of nn.LSTM at
 *      models/encoders.py:161 and of the decoder loop, models/decoders.py:129-132,264-267).  Same partition,
 *      residency rule (dvae_lstm_seq_supported) and chains as the forward call; steps are walked against
 *      the forward order.  Per chain and step t:
 *        dh  = dh_ext[t] + dG_{next} * w_rec      (next: the step walked just before; none for the first)
 *        dc  = dc_{next} * f_{next} + dh * o * (1 - tanh^2 c_t)
 *        dG_t = [dc g i(1-i), dc c_{prev} f(1-f), dc i (1-g^2), dh tanh(c_t) o(1-o)]   -> dgates [T][B][4H]
 *      with c_prev the cell state of the chronologically previous step (c_init rows, stride ld_cinit, or
 *      zeros before the chain's first step).  dh_ext: h-gradient arriving from outside the chain, step t at
 *      dh_ext + t*step_dh + b*ld_dh.  dh_init (optional, NULL array or entry): receives (or, with
 *      accumulate_dh_init, is increased by) the gradient wrt the hidden state before the chain's first step,
 *      dc_init the one wrt the cell state.  The partial products of the workgroups of a row group are added
 *      in a fixed order: results are bit-identical from run to run.
 *      ws: dvae_lstm_seq_bwd_ws_floats(nchains, B, H) floats, 256-byte aligned; ws[0] as in the forward call. */
int64_t dvae_lstm_seq_bwd_ws_floats(int64_t nchains, int64_t B, int64_t H);
int dvae_lstm_seq_bwd_f32(int64_t nchains, const float* const* w_rec, const float* const* gates,
                          const float* const* cseq, const float* const* c_init, int64_t ld_cinit,
                          const float* const* dh_ext, int64_t ld_dh, int64_t step_dh, float* const* dgates,
                          float* const* dh_init, int64_t ld_dhinit, float* const* dc_init, int64_t ld_dcinit,
                          int accumulate_dh_init, const int* reverse, int64_t T, int64_t B, int64_t H, float* ws,
                          dvae_stream_t stream);

/* ---- the stacked (two-layer) autoregressive decoder loop of LSTMDecoder (models/decoders.py:110-137) in ONE launch.
 *      Cell (t, l): pre = x W_ih_l^T + h_{l,t-1} W_hh_l^T + b_ih_l + b_hh_l with x = h_{1,t-1} for layer 0 (init, the
 *      same [H] vector for every row, at t = 0) and x = h_{0,t} for layer 1; initial states h0 / c0 [B][2H] (layer l
 *      in columns l*H..).  Host arrays of two device pointers (layer 0, layer 1), weights [4H][H], biases [4H].
 *      Outputs gates [2][T][B][4H] (activated), hs / cs [2][T][B][H].  dvae_lstm_dec2_supported(): H in {128, 256}
 *      and ceil(B/32)*(H/8) <= dvae_lstm_resident_workgroups(1, H, 1) (every workgroup resident, one per compute
 *      unit: 256 on a full MI355X); otherwise -EINVAL (use the per-step calls).  ws: dvae_lstm_dec2_ws_floats(B, H) floats, 256-byte aligned, ws[0] as in dvae_lstm_seq_fwd_f32. */
int dvae_lstm_dec2_supported(int64_t T, int64_t B, int64_t H);
int64_t dvae_lstm_dec2_ws_floats(int64_t B, int64_t H);
int dvae_lstm_dec2_fwd_f32(const float* const* w_ih, const float* const* w_hh, const float* const* b_ih,
                           const float* const* b_hh, const float* h0, int64_t ld_h0, const float* c0, int64_t ld_c0,
                           const float* init, float* gates, float* hs, float* cs, int64_t T, int64_t B, int64_t H,
                           float* ws, dvae_stream_t stream);
/* its backward through the chain of cells in ONE launch.  gates / cs: the forward call's outputs; dstates: the
 * gradient reaching the top layer's h of step t at dstates + t*step_ds + b*ld_ds; outputs dgates [2][T][B][4H]
 * (gradient wrt the gate pre-activations: the operand of every weight-gradient GEMM), dh0 / dc0 [B][2H] (initial
 * states) and dx0 [B][H] (layer 0's first input, per row).  Partial products are added in a fixed order. */
int dvae_lstm_dec2_bwd_f32(const float* const* w_ih, const float* const* w_hh, const float* gates, const float* cs,
                           const float* c0, int64_t ld_c0, const float* dstates, int64_t ld_ds, int64_t step_ds,
                           float* dgates, float* dh0, int64_t ld_dh0, float* dc0, int64_t ld_dc0, float* dx0,
                           int64_t T, int64_t B, int64_t H, float* ws, dvae_stream_t stream);
/* Every wait inside the one-launch recurrence kernels is bounded (200 ms: a workgroup that never becomes resident
 * must not hang the GPU); a launch in which one ran out leaves invalid outputs.  Number of workgroups that gave up
 * on the current GPU since the library was loaded (or since the last dvae_lstm_seq_timeouts_clear()); synchronises
 * with the device, so call it where the caller synchronises anyway (end of an epoch, a logging interval). */
int64_t dvae_lstm_seq_timeouts(void);
int dvae_lstm_seq_timeouts_clear(void);

/* ---- optimiser (torch.optim.Adam with L2 weight decay and nn.utils.clip_grad_norm_ as called at
 *      training/trainer.py:56-61,198-204).  Host arrays of n device pointers / sizes.
 *      dvae_grad_norm: total 2-norm of all gradients -> norm_out[0], and
 *      norm_out[1] = min(1, max_norm/(norm+1e-6)) (1 when max_norm <= 0).
 *      ws: dvae_grad_norm_ws_floats(n, numel). */
int64_t dvae_grad_norm_ws_floats(int64_t n, const int64_t* numel);
int dvae_grad_norm_f32(int64_t n, const float* const* grads, const int64_t* numel, float max_norm,
                       float* norm_out, float* ws, dvae_stream_t stream);
/* Scales every gradient in place by *coef (device scalar). */
int dvae_scale_tensors_f32(int64_t n, float* const* tensors, const int64_t* numel, const float* coef,
                           dvae_stream_t stream);
/* dst[i] <- src[i] for n tensors of numel[i] floats in one launch per 48 tensors (the gradients of one backward
 * stage into their run of a data-parallel step's flat buffer).  No counterpart in the reference (it has no
 * data parallelism). */
int dvae_copy_tensors_f32(int64_t n, float* const* dst, const float* const* src, const int64_t* numel,
                          dvae_stream_t stream);
/* One Adam step on n tensors; gradients are multiplied by *grad_coef (device scalar, NULL = 1) as
 * they are read.  bias_correction1 = 1-beta1^t, bias_correction2 = 1-beta2^t computed by the caller,
 * unless dyn is given: four device floats {step count t, learning rate, 1-beta1^t, sqrt(1-beta2^t)};
 * then lr and both corrections are read from dyn inside the kernel, which is what lets a captured
 * hipGraph replay the step while t and the learning rate change.  dvae_adam_advance_f32 does
 * dyn[0] += 1 on the stream and refreshes dyn[2], dyn[3] for the new count (computed in double from the
 * double betas, rounded to fp32 once, like a host-side caller's float arguments); the caller owns
 * dyn[1]. */
int dvae_adam_step_f32(int64_t n, float* const* params, const float* const* grads, float* const* exp_avg,
                       float* const* exp_avg_sq, const int64_t* numel, float lr, float beta1, float beta2,
                       float eps, float weight_decay, float bias_correction1, float bias_correction2,
                       const float* grad_coef, const float* dyn, dvae_stream_t stream);
int dvae_adam_advance_f32(float* dyn, double beta1, double beta2, dvae_stream_t stream);

/* ---- launch tracing for benchmarks.  No counterpart in the reference (it times whole calls with a host
 *      clock, evaluation/metrics.py:143-177).  Between dvae_trace_begin and dvae_trace_end every kernel
 *      launch of the library is bracketed by two events on the stream it is launched on and recorded with
 *      its symbol (demangled, as rocprofv3 prints it) and the algorithmic work its launcher declares:
 *      FLOPs of a GEMM-shaped kernel (2*M*N*K of the layer, no padding), bytes an HBM-bound kernel must
 *      read and write once; unit 0 = no declaration (helper kernels).  At most max_launches records are
 *      kept.  dvae_trace_end returns the number recorded; dvae_trace_read waits for record i's events
 *      and returns its duration in milliseconds.  Do not trace under stream capture. */
#define DVAE_TRACE_UNIT_NONE 0
#define DVAE_TRACE_UNIT_FLOP 1
#define DVAE_TRACE_UNIT_BYTE 2
int dvae_trace_begin(int64_t max_launches);
int64_t dvae_trace_end(void);
int dvae_trace_read(int64_t i, char* name, int64_t name_cap, float* ms, double* work, int* unit);

#ifdef __cplusplus
}
#endif
#endif
