/*
 * libmgx — C-ABI of the MI355X (gfx950) kernels behind the progressive-GAN training step of
 * yefeza/music-generator (trainer/models.py, driven by trainer/task.py).
 *
 * The reference has no FFI of its own: its arithmetic is delegated to TensorFlow/Keras layer
 * objects.  Each entry point below therefore replaces the TensorFlow kernel(s) behind one Keras
 * layer / TF op that trainer/models.py instantiates, and cites that call site.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (row-major NHWC float32 unless said
 *     otherwise); nothing is allocated, freed or synchronised inside the library;
 *   - `stream` is a hipStream_t passed as void*; all work is enqueued on it and the call returns;
 *   - return value: 0 = ok (MGX_OK), 1 = bad argument, 2 = launch failure, 3 = workspace too small;
 *   - calls are stateless and re-entrant per stream; results are bitwise reproducible (no float
 *     atomics: every cross-workgroup sum is a fixed-order two-stage reduction through a
 *     caller-provided workspace).
 */
#ifndef MGX_H
#define MGX_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MGX_OK 0
#define MGX_ERR_ARG 1
#define MGX_ERR_LAUNCH 2
#define MGX_ERR_WORKSPACE 3

int mgx_version(void);

/* ------------------------------------------------------------------------------------------
 * Contractions: one implicit-GEMM descriptor covers Conv2D ('same' stride 1, 'valid' strided),
 * Conv2DTranspose with kernel == stride, Dense on the last axis, and the data-gradient of each.
 *
 *   rows    r = (img, ho, wo)  over nimg*hout*wout
 *   columns n = (a', b', co)   over ngemm = dkh*dkw*cout          (depth-to-space on store)
 *   acc[r][n] = sum_{a<kh, b<kw, k<cin} X[img, ho*sh+a-pt, wo*sw+b-pl, k] * W[tap(a,b)][k][n]
 *               (out-of-range input positions contribute 0; if xmul != NULL, X is multiplied
 *                element-wise by xmul on load: used for dY (.) f'(pre-activation))
 *   v = acc + bias[n % nbias];  if act_alpha > 0: v' = tanh(v) + v/act_alpha, and
 *       dact (if != NULL) receives 1 - tanh(v)^2 + 1/act_alpha
 *   Y[img, ho*dkh+a', wo*dkw+b', co] (+)= v'
 *   W[tap][k][n] is read at  w + tap*wtap + k*wsk + n*wsn  with one of wsk, wsn equal to 1;
 *   flip = 1 uses tap (kh-1-a, kw-1-b) for gather position (a,b) (data-gradient of a conv).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t nimg, hin, win, cin;
    int32_t hout, wout;
    int32_t kh, kw, sh, sw, pt, pl;
    int32_t flip;
    int32_t ngemm;
    int32_t dkh, dkw, cout;
    int32_t wsk, wsn;
    int64_t wtap;
    int32_t nbias;
    int32_t accumulate;
    float act_alpha;
} mgx_conv_desc;

/* Forward / data-gradient implicit GEMM on v_mfma_f32_32x32x2_f32.
 * Replaces the TF kernels behind Conv2D (models.py:256-281, 314-335, 363, 416), Conv2DTranspose
 * (models.py:365-393, 418-444), Dense on rank-4 tensors (models.py:366-399, 419-450), the
 * SoftRectifier that follows every discriminator conv (models.py:205-206, fused as epilogue), and
 * their tf.GradientTape data-gradients (models.py:118, 144). */
int mgx_conv_gemm_f32(const mgx_conv_desc* d, const float* x, const float* xmul, const float* w,
                      const float* bias, float* y, float* dact, void* stream);

/* Weight/bias gradient of the same contraction (tf.GradientTape, models.py:118, 144):
 *   dW[tap][k][n] += sum_r X[gather(r, tap)][k] * dY[scatter(r, n)]   (dY multiplied by dymul if set)
 *   dB[n % nbias] += sum_r dY[scatter(r, n)]
 * `workspace` holds per-split partial tiles; its size in bytes comes from
 * mgx_conv_wgrad_workspace().  Fixed-order reduction: bitwise reproducible. */
int64_t mgx_conv_wgrad_workspace(const mgx_conv_desc* d);
int mgx_conv_wgrad_f32(const mgx_conv_desc* d, const float* x, const float* dy, const float* dymul,
                       float* dw, float* dbias, void* workspace, int64_t workspace_bytes, void* stream);

/* Which kernel family serves the contraction: 0 = generic implicit GEMM, 1 = LDS tile kernels
 * (stride-1 'same' 4x4 / 3x3 / 2x2 / 1x1 convs over 32-channel slices), 2 = vector-ALU 1x1 kernel for 2-channel
 * inputs (forward only, when no input multiplier is passed), 3 = Winograd F(2x2,3x3) kernel (3x3
 * 'same' convolutions with 32 output channels, forward or data-gradient weight layout, and their
 * weight gradients; with an input multiplier the forward / data-gradient form only for 32 input
 * channels and no activation, then also with 128 output channels; the tile kernels otherwise;
 * MGX_WINOGRAD=0 in the environment turns it off), 4 = Dense-chain GEMM with the weight matrix in LDS
 * (1x1, 64 / 128 / 256 channels on both sides, no multiplier, activation or accumulation: the
 * stage-0 generator's Dense layers and their data-gradients), 5 = tap-parallel GEMM for the data-gradient of a
 * Conv2DTranspose with kernel (1,T) = stride, 64 filters -> 32 channels (models.py:422).  `wgrad` = 0 asks about mgx_conv_gemm_f32
 * without an input multiplier, 2 about mgx_conv_gemm_f32 with one, 1 about mgx_conv_wgrad_f32.
 * d->act_alpha is taken into account.  No counterpart in the reference: it lets a caller label its
 * timings with the kernel that actually ran. */
int mgx_conv_route(const mgx_conv_desc* d, int wgrad);

/* ------------------------------------------------------------------------------------------
 * Fused forward of one generator grow block (add_generator_block, models.py:362-399): given the
 * previous block's 2-channel output x (nimg, hl, wl, 2) it produces the new block's 2-channel output
 * y (nimg, 2*hl, 2*wl, 2) = Dense(2)(Dense(8)(Dense(64)(Dense(256)(Add_k Dense(128)(ConvT(1,2)(Dense(32)(
 * ConvT(2,1)(Conv2D(2,2,s2)(UpSampling2D(x)))))))))) without writing any intermediate tensor.
 * Weights are the Keras-layout variables of the block's layers; `pack` is scratch of
 * mgx_gblock_pack_floats() floats that the call refills from them.  Used for forwards that keep no
 * tape (the generator passes of models.py:110, 125, 151 and inference, utils.py:81, 104).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const float *k0, *b0;            /* Conv2D(128,(2,2),s2): kernel (2,2,2,128), bias (128)      */
    const float *k1[6], *b1[6];      /* Conv2DTranspose(16,(2,1)): kernel (2,1,16,128), bias (16)  */
    const float *d1[6], *c1[6];      /* Dense(32): kernel (16,32), bias (32)                       */
    const float *k2[6], *b2[6];      /* Conv2DTranspose(64,(1,2)): kernel (1,2,64,32), bias (64)   */
    const float *d2[6], *c2[6];      /* Dense(128): kernel (64,128), bias (128)                    */
    const float *w1, *e1;            /* Dense(256): (128,256), (256)                               */
    const float *w2, *e2;            /* Dense(64):  (256,64),  (64)                                */
    const float *w3, *e3;            /* Dense(8):   (64,8),    (8)                                 */
    const float *w4, *e4;            /* Dense(2):   (8,2),     (2)                                 */
} mgx_gblock_weights;

int64_t mgx_gblock_pack_floats(void);
/* All three entry points work on a range of low-resolution positions [pos0, pos0 + npos) of the
 * flattened (nimg, hl, wl) grid (pos0 a multiple of 128; npos <= 0: through the end), so a caller can
 * bound the size of the side buffers: tile t of a side buffer is positions pos0 + 32t ...; a side
 * buffer for n positions holds mgx_gblock_side_floats_n(n) floats.  A sequence of calls over
 * ranges must start with pos0 = 0: that call refills `pack` / `packt` from the weights.
 *
 * Forward: `act` = NULL writes nothing but y.  Otherwise `act` receives every intermediate tensor
 * the weight gradients need (the tape of tf.GradientTape, models.py:141-144) in the tile-major
 * layout the backward entry points read; y may then be NULL (re-computation of the tape). */
int64_t mgx_gblock_side_floats_n(int64_t npos);
int64_t mgx_gblock_side_floats(int64_t nimg, int hl, int wl);
int mgx_gblock_fwd_f32(const mgx_gblock_weights* w, float* pack, const float* x, float* y, float* act,
                       int64_t nimg, int hl, int wl, int64_t pos0, int64_t npos, void* stream);

/* Backward of the same block (the generator part of tape.gradient, models.py:144-146).
 * mgx_gblock_dgrad_f32: dy (nimg, 2*hl, 2*wl, 2) -> dx (nimg, hl, wl, 2) for the positions of the
 * range, and every intermediate gradient into `delta` (side buffer of the range); `packt` is scratch
 * of mgx_gblock_packt_floats() floats.
 * mgx_gblock_wgrad_f32: ADDS the weight and bias gradients of all 58 tensors, computed from the side
 * buffers `act` and `delta` of the same npos positions, into the Keras-layout gradient tensors of
 * `g` (same field order as mgx_gblock_weights); `workspace` holds mgx_gblock_wgrad_workspace_n(npos)
 * bytes of per-split partial sums that are added in fixed order (bitwise reproducible). */
typedef struct {
    float *k0, *b0;
    float *k1[6], *b1[6];
    float *d1[6], *c1[6];
    float *k2[6], *b2[6];
    float *d2[6], *c2[6];
    float *w1, *e1, *w2, *e2, *w3, *e3, *w4, *e4;
} mgx_gblock_grads;

int64_t mgx_gblock_packt_floats(void);
int64_t mgx_gblock_wgrad_workspace_n(int64_t npos);
int64_t mgx_gblock_wgrad_workspace(int64_t nimg, int hl, int wl);
int mgx_gblock_dgrad_f32(const mgx_gblock_weights* w, float* packt, const float* dy, float* delta, float* dx,
                         int64_t nimg, int hl, int wl, int64_t pos0, int64_t npos, void* stream);
int mgx_gblock_wgrad_f32(const float* act, const float* delta, const mgx_gblock_grads* g, void* workspace,
                         int64_t workspace_bytes, int64_t npos, void* stream);

/* ------------------------------------------------------------------------------------------
 * Head of the stage-0 generator (define_generator, models.py:412-444): from the latent tensor z
 * (nimg, 1, 100, 2) through featured = Conv2D(128, (1,2), strides (1,2)) (models.py:416) and, per branch
 * b = 0..3, Conv2DTranspose(4,(2,1)) -> Dense(8) -> Conv2DTranspose(16,(2,1)) -> Dense(32)
 * (models.py:418-421, 425-428, 432-435, 439-442) to the four tensors y[b] (nimg, 4, 50, 32) that the
 * Conv2DTranspose(64, (1,15)) layers read: seventeen layers without an activation in one launch.
 * Weights are the Keras-layout variables of the layers.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
    const float *k0, *b0;            /* Conv2D(128,(1,2),s(1,2)): kernel (1,2,2,128), bias (128)   */
    const float *k1[4], *b1[4];      /* Conv2DTranspose(4,(2,1)): kernel (2,1,4,128), bias (4)      */
    const float *d1[4], *c1[4];      /* Dense(8): kernel (4,8), bias (8)                            */
    const float *k2[4], *b2[4];      /* Conv2DTranspose(16,(2,1)): kernel (2,1,16,8), bias (16)     */
    const float *d2[4], *c2[4];      /* Dense(32): kernel (16,32), bias (32)                        */
} mgx_ghead_weights;

typedef struct {
    float *k0, *b0;
    float *k1[4], *b1[4];
    float *d1[4], *c1[4];
    float *k2[4], *b2[4];
    float *d2[4], *c2[4];
} mgx_ghead_grads;

typedef struct {
    float* y[4];                     /* per branch: (nimg, 4, 50, 32)                               */
} mgx_ghead_tensors;

int mgx_ghead_fwd_f32(const mgx_ghead_weights* w, const float* z, int64_t nimg, const mgx_ghead_tensors* out,
                      void* stream);
/* The head's part of tape.gradient (models.py:144-146): `dout->y[b]` holds d(loss)/d(y[b]) (read only); the
 * weight and bias gradients of all 34 tensors are ADDED into `g` (per-wave partial sums in `workspace`,
 * mgx_ghead_bwd_workspace(nimg) bytes, added in a fixed order: bitwise reproducible); g = NULL skips them.
 * dz (nimg, 1, 100, 2), if not NULL, receives d(loss)/dz.  The intermediate tensors are re-computed from z. */
int64_t mgx_ghead_bwd_workspace(int64_t nimg);
int mgx_ghead_bwd_f32(const mgx_ghead_weights* w, const float* z, int64_t nimg, const mgx_ghead_tensors* dout,
                      const mgx_ghead_grads* g, float* dz, void* workspace, int64_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Non-contraction layers (HBM-bound, coalesced, wavefront reductions)
 * ------------------------------------------------------------------------------------------ */

/* UpSampling2D() nearest x2 (models.py:362) and its gradient (sum of the 2x2 block). */
int mgx_upsample2x_fwd(const float* x, float* y, int64_t nimg, int h, int w, int c, void* stream);
int mgx_upsample2x_bwd(const float* dy, float* dx, int64_t nimg, int h, int w, int c, int accumulate, void* stream);

/* AveragePooling2D() 2x2/2 (models.py:292) and its gradient; h, w are the INPUT sizes. */
int mgx_avgpool2x_fwd(const float* x, float* y, int64_t nimg, int h, int w, int c, void* stream);
int mgx_avgpool2x_bwd(const float* dy, float* dx, int64_t nimg, int h, int w, int c, int accumulate, void* stream);

/* keras Add of up to 8 tensors (models.py:395, 446). */
int mgx_add_n(const float* const* xs, int nin, float* y, int64_t n, void* stream);
/* dst += src (gradient fan-in). */
int mgx_add_inplace(float* dst, const float* src, int64_t n, void* stream);
/* One micro-batch of whole MinibatchStdDev groups out of a batch of g * m samples (the groups of
 * models.py:179-195 are the members m', m + m', 2m + m', ... of the batch): output sample a * cnt + b is input
 * sample a * m + m0 + b for a < g, b < cnt, i.e. g contiguous runs of cnt samples of row_floats floats each
 * (a multiple of 4).  Splitting train_step (models.py:72-158) into micro-batches has no counterpart in the
 * reference; it is what lets the upper stages fit in HBM. */
int mgx_gather_groups(const float* src, float* dst, int64_t row_floats, int g, int m, int m0, int cnt, void* stream);

/* Generator inference through the collapsed affine map - an opt-in inference path with no counterpart in the
 * reference, whose generator (models.py:358-399, 412-450) has no non-linearity in front of its final
 * LayerNormalization: the tensor entering the normalisation is T[p][ch][0] + sum_k z[c][k] T[p][ch][1+k] for latent
 * column pair c (the 4 numbers z[0, 2c..2c+1, :]) and position p = (row, j) of the h x wp patch the pair produces.
 * mgx_collapse_table builds T ((h*wp) x 2 x 5 floats) from `probe` = that tensor for five latents (zero, then the
 * four unit vectors set at every column pair; 5 x h x (ncols*wp) x 2 floats) and `gram` (20 doubles: sums and
 * products of the five table vectors).  mgx_collapse_apply writes out[n] = gamma * (pre[n] - mean_n) * rstd_n + beta
 * (nclips x h x (ncols*wp) x 2 floats) for latents z (nclips x ncols x 4 floats); mean and rstd = 1/sqrt(var + eps)
 * come from `gram` (no pass over the clip) and are left in stats (nclips x 2 floats). */
int mgx_collapse_table(const float* probe, float* table, double* gram, int h, int wp, int ncols, void* stream);
int mgx_collapse_apply(const float* table, const double* gram, const float* z, const float* gamma, const float* beta,
                       float* out, float* stats, int nclips, int h, int wp, int ncols, float eps, void* stream);

/* WeightedSum (models.py:30-35): y = (1-alpha)*a + alpha*b, alpha read from device memory.
 * Backward: da (+)= (1-alpha)*dy, db (+)= alpha*dy, dalpha += sum(dy*(b-a)) (if dalpha != NULL;
 * workspace of mgx_reduce_workspace() bytes). da/db may be NULL. */
int mgx_weighted_sum_fwd(const float* a, const float* b, const float* alpha, float* y, int64_t n, void* stream);
int mgx_weighted_sum_bwd(const float* dy, const float* a, const float* b, const float* alpha,
                         float* da, int acc_a, float* db, int acc_b, float* dalpha,
                         void* workspace, int64_t workspace_bytes, int64_t n, void* stream);
int64_t mgx_reduce_workspace(void);

/* Per-clip checksums of a chunk of generated audio x (nclips, elems) (the tensors generar_ejemplo receives from
 * g_model, utils.py:80-81 / 103-104): out[clip] = (sum, sum of squares, absolute peak) in float64, one read of
 * the chunk, parts added in a fixed order.  elems a multiple of 4, x 16-byte aligned; workspace of
 * mgx_clip_checksum_workspace(nclips) bytes.  No counterpart in the reference, which writes the audio out: it
 * lets a sampling run or a benchmark consume a chunk on the device. */
int64_t mgx_clip_checksum_workspace(int64_t nclips);
int mgx_clip_checksum(const float* x, int64_t nclips, int64_t elems, double* out, void* workspace,
                      int64_t workspace_bytes, void* stream);

/* LayerNormalization(axis=[1,2,3]) (models.py:400, 405, 451): per-sample mean / biased variance
 * over e = H*W*C elements, epsilon eps, gamma/beta of e elements.  stats receives (mean, rstd)
 * per sample (2*nimg floats).  Backward: dx (+)=..., dgamma += , dbeta += (sums over the batch). */
int64_t mgx_layernorm_workspace(int64_t nimg, int64_t e);
int mgx_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* stats,
                      int64_t nimg, int64_t e, float eps, void* workspace, int64_t workspace_bytes, void* stream);
int mgx_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* stats,
                      float* dx, int acc_x, float* dgamma, float* dbeta,
                      int64_t nimg, int64_t e, void* workspace, int64_t workspace_bytes, void* stream);

/* mgx_layernorm_fwd without a workspace argument.  No counterpart in the reference: it is for callers
 * that cannot allocate device memory (code under HIP graph capture, tight serving loops, harnesses that
 * forbid hipMalloc).  The partial sums go to a 32 KiB array that lives in the library's device image
 * (one copy per GPU, held until the process ends); the call allocates and frees nothing.  Same kernels and
 * bit-identical results as mgx_layernorm_fwd.  Limit: nimg <= mgx_layernorm_fwd_static_max_samples()
 * (2048), otherwise MGX_ERR_WORKSPACE.  All calls on one GPU share that array: calls that overlap on one
 * GPU (different streams) race; calls on one stream are ordered. */
int64_t mgx_layernorm_fwd_static_max_samples(void);
int mgx_layernorm_fwd_static(const float* x, const float* gamma, const float* beta, float* y, float* stats,
                             int64_t nimg, int64_t e, float eps, void* stream);

/* MinibatchStdDev (models.py:167-192): x (n,h,w,c) -> y (n,h+1,w,c); group g = min(4,n), sample i
 * belongs to group i % (n/g); the group statistic fills row h.  stat receives n/g floats.
 * Backward: dx = dy[:, :h] + d(stat)/dx * sum(dy[:, h]). */
int64_t mgx_mbstd_workspace(int64_t n, int64_t e);
int mgx_mbstd_fwd(const float* x, float* y, float* stat, int64_t n, int h, int w, int c,
                  void* workspace, int64_t workspace_bytes, void* stream);
int mgx_mbstd_bwd(const float* dy, const float* x, float* dx, int64_t n, int h, int w, int c,
                  void* workspace, int64_t workspace_bytes, void* stream);

/* SoftRectifier / StaticOptTanh stand-alone (models.py:205-206, 219-220): y = tanh(x) + x/alpha;
 * backward dx (+)= dy * (1 - tanh(x)^2 + 1/alpha). */
int mgx_softrect_fwd(const float* x, float* y, int64_t n, float alpha, void* stream);
int mgx_softrect_bwd(const float* dy, const float* x, float* dx, int accumulate, int64_t n, float alpha, void* stream);

/* Row softmax: y[r, :] = softmax(x[r, :]), the activation of the evaluator's Dense(9) head
 * (trainer/evaluacion.py:97), whose outputs feed calculate_inception_score (trainer/utils.py:65-76). */
int mgx_softmax_rows(const float* x, float* y, int64_t rows, int cols, void* stream);

/* Fitting the evaluator (trainer/evaluacion.py:158-161, compile(optimizer="Adam",
 * loss='categorical_crossentropy')): softmax probabilities, per-row loss -sum_c y log softmax(x)_c taken from
 * the logits (as tf.keras does for the output of a Softmax op: no probability clipping) and (dx may be
 * NULL) the gradient of the batch-mean loss with respect to the logits, (p - y)/rows. */
int mgx_softmax_xent(const float* x, const float* onehot, float* prob, float* loss_rows, float* dx,
                     int64_t rows, int cols, void* stream);

/* Flatten + Dense(1) (models.py:339-340): y[i] = sum_e x[i,e]*w[e] + b.  Backward: dx[i,e] =
 * dy[i]*w[e] (if dx != NULL), dw[e] += sum_i dy[i]*x[i,e], db += sum_i dy[i] (if dw != NULL). */
int mgx_dense1_fwd(const float* x, const float* w, const float* b, float* y, int64_t n, int64_t e, void* stream);
int mgx_dense1_bwd(const float* dy, const float* x, const float* w, float* dx, float* dw, float* db,
                   int64_t n, int64_t e, void* stream);

/* Losses (models.py:468-484) and the 6 metrics (models.py:128-130, 154-156).
 * mgx_logit_sums: sums[0] (+)= sum(fake), sums[1] (+)= sum(real) over n local logits (the pair that
 *   a data-parallel run all-reduces; accumulate = 1 adds over micro-batches).
 * mgx_wgan_loss: from the GLOBAL sums and global batch n_global: ci, cu, sign = (cu-ci)/|cu-ci|
 *   (0 if equal); mode 0 = discriminator_loss, 1 = generator_loss; out[0]=loss, out[1]=cu,
 *   out[2]=ci, out[3]=|cu-ci|; dfake/dreal (n_local each, may be NULL) receive dloss/dlogit:
 *   mode 0: dreal = (sign+1e-4)/n_global, dfake = -(sign+9.99e-5)/n_global;
 *   mode 1: dfake = sign/n_global (sign treated as a constant). */
int mgx_logit_sums(const float* fake, const float* real, int64_t n, float* sums, int accumulate, void* stream);
int mgx_wgan_loss(const float* sums, int64_t n_global, int mode, float* out4,
                  float* dfake, float* dreal, int64_t n_local, void* stream);

/* out = ca[0]*a + cb[0]*b (b, cb may be NULL): the loss gradients dloss/dlogit are uniform over the
 * batch (see mgx_wgan_loss), so the step back-propagates unit seeds per micro-batch, separately for
 * the fake and the real pass, and scales / combines the two flat gradient buffers here once the
 * batch-wide sign is known.  ca, cb are device scalars (dfake[0], dreal[0] of mgx_wgan_loss).  out may
 * be the same buffer as a. */
int mgx_combine_grads(float* out, const float* a, const float* b, const float* ca, const float* cb, int64_t n,
                      void* stream);

/* Adam.apply_gradients with the hyper-parameters of models.py:502-503 (beta_1 = 0 so m == g):
 *   v = beta2*v + (1-beta2)*g*g;  p -= lr_t * g / (sqrt(v) + eps).
 * Multi-tensor: tensor i has n[i] elements at p[i], its gradient and second moment at offset
 * off[i] of the flat buffers g and v.  The three tables live in device memory; blk_tensor /
 * blk_start (nblocks entries, device) map each 1024-thread-block's chunk of 4096 elements. */
int mgx_adam_multi(float* const* p, const int64_t* n, const int64_t* off, const float* g, float* v,
                   const int32_t* blk_tensor, const int64_t* blk_start, int64_t nblocks,
                   float lr_t, float beta2, float eps, void* stream);
/* Adam with a first moment (keras' default beta_1 = 0.9, the optimiser of evaluacion.py:160):
 *   m = beta1*m + (1-beta1)*g;  v = beta2*v + (1-beta2)*g*g;  p -= lr_t * m / (sqrt(v) + eps),
 * lr_t = lr*sqrt(1-beta2^t)/(1-beta1^t) computed by the caller.  Tables as in mgx_adam_multi. */
int mgx_adam_multi_m(float* const* p, const int64_t* n, const int64_t* off, const float* g, float* m, float* v,
                     const int32_t* blk_tensor, const int64_t* blk_start, int64_t nblocks,
                     float lr_t, float beta1, float beta2, float eps, void* stream);
/* The same update with the step size lr_t = lr*sqrt(1-beta2^t) (keras OptimizerV2, the optimisers of
 * models.py:502-503) read from device memory at execution time: a train step captured once into a HIP
 * graph is replayed with a new t by rewriting that one float before the replay. */
int mgx_adam_multi_devlr(float* const* p, const int64_t* n, const int64_t* off, const float* g, float* v,
                         const int32_t* blk_tensor, const int64_t* blk_start, int64_t nblocks,
                         const float* lr_t, float beta2, float eps, void* stream);

/* ------------------------------------------------------------------------------------------
 * bf16 path of the discriminator's grow blocks (add_discriminator_block, models.py:256-281, and its
 * tf.GradientTape gradients, models.py:118, 144): activations with 32 / 128 channels are stored as
 * NHWC bfloat16, every contraction runs on v_mfma_f32_32x32x16_bf16 with float32 accumulation, weights,
 * biases and all gradients of variables stay float32 (master copies; the kernels round the weights to
 * bf16 as they stage them).  2-channel tensors (audio, its gradient) stay float32.
 * Pointers typed void* are bf16 device tensors.
 * ------------------------------------------------------------------------------------------ */

/* 1 if mgx_conv_gemm_bf16 serves the descriptor: stride-1 'same' 3x3 / 2x2 / 1x1 convolutions (forward or
 * data-gradient tap order) between 32 and 32 or 128 channels, the 2x2 stride-2 'valid' convolution
 * 32 -> 32 and its depth-to-space data-gradient. */
int mgx_conv_bf16_supported(const mgx_conv_desc* d);

/* The contraction of mgx_conv_gemm_f32 on bf16 activations (Conv2D + SoftRectifier of models.py:259-281):
 * x, y bf16; w, bias float32.  With d->act_alpha > 0 (and no ymul) the epilogue is tanh(v)+v/alpha and dact
 * (may be NULL) receives the derivative as bf16.  ymul (may be NULL; with it no bias and no dact): the result
 * is multiplied element-wise, before it is stored, by a factor taken from this bf16 tensor of the output's
 * shape: the data-gradient of a convolution whose input came out of an activation, dX = (W^T dY) (.) f'.
 * With d->act_alpha == 0 ymul holds the factor itself (a stored derivative); with d->act_alpha > 0 ymul holds
 * that activation's stored OUTPUT y = tanh(v)+v/alpha and the factor is the derivative recovered from it,
 * max(1 + 1/alpha - y^2, 1/alpha): backward passes that keep no derivative tensor (the layer input the
 * weight gradient needs anyway is that output).  d->accumulate must be 0. */
int mgx_conv_gemm_bf16(const mgx_conv_desc* d, const void* x, const float* w, const float* bias, void* y,
                       void* dact, const void* ymul, void* stream);
/* The 1x1 convolution + SoftRectifier that ends a discriminator block (models.py:280-281) and the WeightedSum
 * behind it (models.py:297, 30-35) in one launch, for passes that keep no tape: y = (1 - alpha) old + alpha
 * act(conv(x) + bias); x, old, y bf16 (old of y's shape), alpha one float in device memory, act_alpha in the
 * descriptor.  Rounded where the two launches it replaces round (results are bit-identical to them). */
int mgx_conv1x1_blend_bf16(const mgx_conv_desc* d, const void* x, const float* w, const float* bias, const void* old,
                           const float* alpha, void* y, void* stream);
/* Data gradient of the block's first 3x3 convolution (128 -> 32 channels) and of the from-audio Conv2D(128, 1x1) +
 * SoftRectifier in front of it (models.py:256-259) in one launch, for backward sweeps that need no weight gradients
 * (tape.gradient(g_loss, generator variables) through the discriminator, models.py:144): d is the 3x3 layer's
 * data-gradient descriptor (32 -> 128 channels, d->act_alpha the from-audio layer's alpha), g the gradient with respect
 * to the 3x3 layer's pre-activation (bf16), y128 the from-audio layer's stored output (bf16), audio_w its kernel
 * (2, 128); dx (nimg, h, w, 2) float32 receives the gradient of the audio (is added to when accumulate != 0).  The
 * values mgx_conv_gemm_bf16 (ymul) would store as bf16 are contracted with audio_w in registers: equal to that call
 * followed by mgx_pointwise_to2_bf16 up to the order of the float32 sums. */
int mgx_conv_dgrad_to_audio_bf16(const mgx_conv_desc* d, const void* g, const float* w, const void* y128,
                                 const float* audio_w, float* dx, int accumulate, void* stream);
/* The same launch with `old` given as the layer that produces it in the fade-in graph (models.py:292-296): the old
 * stage's from-audio Conv2D(128, 1x1) + SoftRectifier(audio_alpha) applied to the average-pooled 2-channel audio
 * (float32, (nimg, h, w, 2) of the output's height and width), kernel audio_w (2, 128) and bias audio_b (128).  Every
 * element of `old` is re-computed in the epilogue with the arithmetic of mgx_pointwise2_bf16_fwd and rounded to bf16
 * as that kernel stores it (8 bytes a position read instead of 256, and no launch that writes them): results are
 * bit-identical to mgx_pointwise2_bf16_fwd + mgx_conv_gemm_bf16 + mgx_weighted_sum_bf16_fwd.  128 output channels. */
int mgx_conv1x1_blend_audio_bf16(const mgx_conv_desc* d, const void* x, const float* w, const float* bias,
                                 const float* audio, const float* audio_w, const float* audio_b, float audio_alpha,
                                 const float* alpha, void* y, void* stream);
/* The same for float32 tensors (the 1x1 tile kernel, 32 -> 128 channels): mgx_conv1x1_blend_supported tells whether
 * the descriptor is served. */
int mgx_conv1x1_blend_supported(const mgx_conv_desc* d);
int mgx_conv1x1_blend_f32(const mgx_conv_desc* d, const float* x, const float* w, const float* bias, const float* old,
                          const float* alpha, float* y, void* stream);

/* Weight / bias gradient for a forward descriptor mgx_conv_bf16_supported accepts: x the layer input,
 * g the gradient with respect to the pre-activation (both bf16); dw (Keras layout through the
 * descriptor's strides) and dbias (may be NULL) are float32 and are ADDED to.  Fixed-order two-stage
 * reduction through `workspace` (mgx_conv_wgrad_bf16_workspace bytes). */
int64_t mgx_conv_wgrad_bf16_workspace(const mgx_conv_desc* d);
int mgx_conv_wgrad_bf16(const mgx_conv_desc* d, const void* x, const void* g, float* dw, float* dbias,
                        void* workspace, int64_t workspace_bytes, void* stream);

/* Two consecutive Conv2D + SoftRectifier layers of a discriminator block (trainer/models.py:259-276) in one launch,
 * for forward passes that keep no tape: Conv2D(32, 3x3, 'same') + SoftRectifier(alpha_a), then Conv2D(32, kb x kb,
 * 'same', kb = 3 or 2) + SoftRectifier(alpha_b), 32 -> 32 -> 32 channels.  x and y are bf16 NHWC (nimg, h, w, 32);
 * wa / wb the float32 Keras kernels (3, 3, 32, 32) / (kb, kb, 32, 32), ba / bb the float32 biases.  The first
 * layer's output stays in LDS; results are bit-identical to two calls of mgx_conv_gemm_bf16.  Needs h >= 16
 * (mgx_conv_pair_bf16_supported). */
int mgx_conv_pair_bf16_supported(int64_t nimg, int h, int w, int kb);
int mgx_conv_pair_bf16(int64_t nimg, int h, int w, const void* x, const float* wa, const float* ba, float alpha_a,
                       const float* wb, const float* bb, float alpha_b, int kb, void* y, void* stream);

/* The first two layers of a discriminator block (trainer/models.py:256-260) in one launch, for forward passes that
 * keep no tape: Conv2D(128, 1x1) + SoftRectifier(alpha_a) on the float32 audio x (nimg, h, w, 2), then Conv2D(32,
 * 3x3, 'same') + SoftRectifier(alpha_b); wa (2, 128), ba (128), wb (3, 3, 128, 32), bb (32) float32, y bf16
 * (nimg, h, w, 32).  The 128-channel tensor between the two is re-computed per tile and never stored.  Results are
 * bit-identical to mgx_pointwise2_bf16_fwd followed by mgx_conv_gemm_bf16.  Needs h a multiple of 16
 * (mgx_audio_conv3_bf16_supported). */
int mgx_audio_conv3_bf16_supported(int64_t nimg, int h, int w);
int mgx_audio_conv3_bf16(int64_t nimg, int h, int w, const float* x, const float* wa, const float* ba, float alpha_a,
                         const float* wb, const float* bb, float alpha_b, void* y, void* stream);

/* A chain of two or three consecutive Conv2D(32, k x k, 'same') + SoftRectifier layers of a discriminator block
 * (trainer/models.py:259-276) in one launch, for forward passes that keep no tape.  ksizes = {3,3,2}, {3,3} or
 * {3,2} (mgx_conv_chain_bf16_supported); x and y are bf16 NHWC (nimg, h, w, 32); weights[i] / biases[i] the
 * float32 Keras kernel (k, k, 32, 32) / bias (32) of layer i, alphas[i] its SoftRectifier alpha (> 0).  A wave
 * streams a strip of 29-31 output columns down the rows of an image with the inner layers' rows in LDS rings:
 * no intermediate reaches HBM and no row is computed twice.  Results are bit-identical to nlayers calls of
 * mgx_conv_gemm_bf16. */
int mgx_conv_chain_bf16_supported(int64_t nimg, int h, int w, int nlayers, const int* ksizes);
int mgx_conv_chain_bf16(int64_t nimg, int h, int w, int nlayers, const int* ksizes, const void* x,
                        const float* const* weights, const float* const* biases, const float* alphas, void* y,
                        void* stream);
/* The same launch for forward passes under tf.GradientTape (models.py:108-118, 133-144), whose backward reads
 * every layer's output: ys[i] (bf16, shape of x) receives the output of layer i, the inner ones as the rows leave the
 * activation on their way into the rings.  Bit-identical to nlayers calls of mgx_conv_gemm_bf16; the chain's input is
 * read once and no inner output is read back. */
int mgx_conv_chain_bf16_taped(int64_t nimg, int h, int w, int nlayers, const int* ksizes, const void* x,
                              const float* const* weights, const float* const* biases, const float* alphas,
                              void* const* ys, void* stream);

/* From-audio 1x1 convolution (models.py:256): x float32 (npos, 2) -> y bf16 (npos, 128) with kernel w
 * (2,128) and bias (128) float32, optional tanh(v)+v/alpha epilogue (act_alpha > 0) and derivative
 * store (dact, bf16, may be NULL).  Its weight gradient (dw (2,128), db (128) ADDED to; g = gradient
 * with respect to the pre-activation, bf16) and its data-gradient dx (npos, 2) float32 (+)= g W^T. */
int mgx_pointwise2_bf16_fwd(const float* x, const float* w, const float* bias, void* y, void* dact, int64_t npos,
                            float act_alpha, void* stream);
int64_t mgx_pointwise2_bf16_wgrad_workspace(int64_t npos);
int mgx_pointwise2_bf16_wgrad(const float* x, const void* g, float* dw, float* db, void* workspace,
                              int64_t workspace_bytes, int64_t npos, void* stream);
int mgx_pointwise_to2_bf16(const void* g, const float* w, float* dx, int64_t npos, int accumulate, void* stream);

/* WeightedSum (models.py:30-35) on bf16 tensors of n elements (n a multiple of 8); dalpha += sum dy (b - a)
 * (float32, workspace of mgx_reduce_workspace() bytes); da, db, dalpha may be NULL. */
int mgx_weighted_sum_bf16_fwd(const void* a, const void* b, const float* alpha, void* y, int64_t n, void* stream);
int mgx_weighted_sum_bf16_bwd(const void* dy, const void* a, const void* b, const float* alpha, void* da, void* db,
                              float* dalpha, void* workspace, int64_t workspace_bytes, int64_t n, void* stream);
/* The same with the gradients multiplied by the saved activation derivatives of the layers that produced a and b
 * (ma, mb: bf16, n elements, either may be NULL): da = bf16(bf16((1-alpha) dy) * ma), db likewise, which is what a
 * multiply pass of its own over da / db would leave. */
int mgx_weighted_sum_bf16_bwd_mul(const void* dy, const void* a, const void* b, const float* alpha, void* da, void* db,
                                  float* dalpha, const void* ma, const void* mb, void* workspace,
                                  int64_t workspace_bytes, int64_t n, void* stream);
/* The same for producers that keep no derivative tensor: a / b ARE the stored outputs of the SoftRectifiers
 * (models.py:295, 281) that produced them, and the factors are the derivatives recovered from them with
 * act_alpha_a / act_alpha_b (0 = that side is not multiplied), as in mgx_conv_gemm_bf16. */
int mgx_weighted_sum_bf16_bwd_act(const void* dy, const void* a, const void* b, const float* alpha, void* da, void* db,
                                  float* dalpha, float act_alpha_a, float act_alpha_b, void* workspace,
                                  int64_t workspace_bytes, int64_t n, void* stream);

/* The generator grow block of mgx_gblock_fwd_f32 / _dgrad_f32 / _wgrad_f32 (add_generator_block,
 * models.py:362-399, and its part of tape.gradient, models.py:144-146) on bf16 MFMAs with float32
 * accumulation.  x, y, dy, dx (2 channels) and all weights / weight gradients are float32; every
 * intermediate activation and gradient is rounded to bf16 once, in registers, and the side buffers (`act`,
 * `delta`: mgx_gblock_side_elems_bf16(npos) bf16 elements each, same tile-major row map as the float32
 * ones) hold bf16.  Position ranges, the scratch buffers `packb` / `packt` (mgx_gblock_pack_bf16_bytes /
 * mgx_gblock_packt_bf16_bytes bytes, refilled by the call with pos0 = 0) and the weight-gradient workspace
 * (mgx_gblock_wgrad_workspace_n bytes) work as in the float32 entry points. */
int64_t mgx_gblock_pack_bf16_bytes(void);
int64_t mgx_gblock_packt_bf16_bytes(void);
int64_t mgx_gblock_side_elems_bf16(int64_t npos);
int mgx_gblock_fwd_bf16(const mgx_gblock_weights* w, void* packb, const float* x, float* y, void* act, int64_t nimg,
                        int hl, int wl, int64_t pos0, int64_t npos, void* stream);
int mgx_gblock_dgrad_bf16(const mgx_gblock_weights* w, void* packt, const float* dy, void* delta, float* dx,
                          int64_t nimg, int hl, int wl, int64_t pos0, int64_t npos, void* stream);
int mgx_gblock_wgrad_bf16(const void* act, const void* delta, const mgx_gblock_grads* g, void* workspace,
                          int64_t workspace_bytes, int64_t npos, void* stream);
/* The same three steps with a LEAN tape: the output of the six Conv2DTranspose(64) and of Dense(256)
 * (models.py:377-378, 386) and the gradients that arrive at them make up 5120 of the 8138 rows that a 32-position
 * tile of the two side buffers holds, and each is one linear layer away from rows the tape keeps.  The _lean
 * entry points use side buffers without those rows (mgx_gblock_side_elems_bf16_lean(npos) bf16 elements each, 38 % of
 * the full size; their own row map, so the three must be used together): the forward and the data gradient do not
 * write them, and mgx_gblock_wgrad_bf16_lean recomputes them tile by tile from the neighbouring rows and the block's
 * weights `w` (the values the forward ran with).  mgx_gblock_fwd_bf16_lean requires `act`.  Same scratch buffers and
 * workspace size (mgx_gblock_wgrad_workspace_n covers both forms); results differ from the full-tape entry points by
 * the order in which an MFMA adds its products (a few flipped bf16 roundings). */
int64_t mgx_gblock_side_elems_bf16_lean(int64_t npos);
int mgx_gblock_fwd_bf16_lean(const mgx_gblock_weights* w, void* packb, const float* x, float* y, void* act, int64_t nimg,
                             int hl, int wl, int64_t pos0, int64_t npos, void* stream);
int mgx_gblock_dgrad_bf16_lean(const mgx_gblock_weights* w, void* packt, const float* dy, void* delta, float* dx,
                               int64_t nimg, int hl, int wl, int64_t pos0, int64_t npos, void* stream);
int mgx_gblock_wgrad_bf16_lean(const mgx_gblock_weights* w, const void* act, const void* delta, const mgx_gblock_grads* g,
                               void* workspace, int64_t workspace_bytes, int64_t npos, void* stream);

/* Data gradient and weight gradient of one bf16 'same' convolution (32 -> 32 channels 3x3 or 2x2, or 128 -> 32
 * channels 3x3; stride 1) in ONE pass:
 * the calls mgx_conv_gemm_bf16 (data-gradient descriptor, ymul = x, act_alpha) and mgx_conv_wgrad_bf16 of the same
 * layer read the same two tensors, the layer input x (= the stored output of the layer before it) and g, the
 * gradient with respect to this layer's pre-activation (tf.GradientTape through Conv2D + SoftRectifier,
 * models.py:118, 256-281).  `d` is the layer's FORWARD descriptor, `w` its kernel (Keras layout); dw / dbias are
 * ADDED to as by mgx_conv_wgrad_bf16; dx (bf16, shape of x) receives the input gradient, multiplied by the
 * derivative 1 - tanh^2 + 1/act_alpha recovered from x when act_alpha > 0 (x is then the output of a SoftRectifier
 * layer).  128 B read + 64 B written per position instead of 384 (320 + 256 instead of 896 from 128 channels).
 * Workspace: mgx_conv_bwd_bf16_workspace(d)
 * bytes; fixed-order reduction, bitwise reproducible. */
int mgx_conv_bwd_bf16_supported(const mgx_conv_desc* d);
int64_t mgx_conv_bwd_bf16_workspace(const mgx_conv_desc* d);
int mgx_conv_bwd_bf16(const mgx_conv_desc* d, const void* x, const void* g, const float* w, float* dw, float* dbias,
                      void* dx, float act_alpha, void* workspace, int64_t workspace_bytes, void* stream);

/* out = a (.) b over n elements (n a multiple of 8): b and out bf16, a bf16 or float32 (a_is_f32): the
 * product dY (.) f' of tf.GradientTape through a SoftRectifier (models.py:205-206) where the kernel that
 * produced dY could not apply the saved derivative itself. */
int mgx_mul_bf16(const void* a, int a_is_f32, const void* b, void* out, int64_t n, void* stream);
/* out = a (.) f'(y): the same product where no derivative tensor is kept; y is the SoftRectifier's stored bf16
 * output and f' is recovered from it as in mgx_conv_gemm_bf16 (act_alpha > 0). */
int mgx_mul_dact_bf16(const void* a, int a_is_f32, const void* y, float act_alpha, void* out, int64_t n, void* stream);

/* Casts at the boundary between the bf16 grow blocks and the float32 base block (no counterpart in the
 * reference, which runs float32 throughout). */
int mgx_cast_f32_to_bf16(const float* x, void* y, int64_t n, void* stream);
int mgx_cast_bf16_to_f32(const void* x, float* y, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif
