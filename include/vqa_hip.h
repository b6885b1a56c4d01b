/* libvqa_hip.so -- C ABI of the MI355X (gfx950) VQA forward/backward kernels.
 *
 * Conventions:
 *   - raw DEVICE pointers + explicit sizes + a hipStream_t; the caller (PyTorch) owns every buffer, including
 *     workspaces and tensors saved for backward; functions never allocate, never synchronise, never throw;
 *   - return value: 0 = launched, 1000 = argument/shape error (nothing launched), otherwise a hipError_t;
 *   - dtype: 0 = float32 (fp32 MFMA, parity path), 1 = bfloat16 (bf16 MFMA, fp32 accumulate); statistics,
 *     coefficients, weight gradients and optimizer state are always float32;
 *   - CNN activations are NHWC, token tensors are [rows][D]; conv weights are [Cout][R][S][Cin];
 *   - "+=" outputs (weight / bias gradients) accumulate into caller-zeroed fp32 buffers.
 * Each entry cites the reference interface (path relative to the reference checkout) it replaces.
 */
#ifndef VQA_HIP_H
#define VQA_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
typedef struct ihipStream_t* hipStream_t;

/* ---- implicit-GEMM convolution / linear -------------------------------------------------------------------
 * Replaces nn.Conv2d forward + its autograd backward (models/cnn_backbone.py:148-158 conv1/conv2, :243-247 shortcut,
 * :350 stem) and every nn.Linear (models/text_encoder.py:176-179,309-310; models/cross_attention.py:98-101,257-263;
 * models/fusion.py:68; models/vqa_model.py:74-82).
 * loader 0: NHWC activations, 1: 7x7/2 stem straight from the NCHW fp32 image.
 * out[M][N] = gather(a)[M][Kw] * w[N][Kw]^T, then +bias, ReLU, dropout(drop_p, drop_seed), + addend*(addmask>0);
 * stats != NULL: per-M-tile column sums / sums of squares ([vqa_igemm_mtiles][2][N]) for train-mode BatchNorm.
 * transposed = 1 gathers for the data gradient (a = dY [B,H,W,C], rows index dX [B,Ho,Wo]). */
int vqa_igemm_mtiles(int M, int N, int loader);
/* which template instantiation vqa_igemm launches for this problem (host-only query, no launch):
 * BM*10000 + BN*10 + flavour (0 plain LDS-DMA double buffer, 1 window loader) */
int vqa_igemm_variant(int dtype, int loader, int M, int N, int Kw, int B, int H, int W, int C, int Ho, int Wo,
                      int R, int S, int stride, int pad);
int vqa_igemm(int dtype, int loader, const void* a, const void* w, void* out, const float* bias, const void* addend,
              const void* addmask, const void* outmask /* out *= (outmask > 0), applied last */, float* stats, int M, int N, int Kw, int B, int H, int W, int C, int Ho, int Wo,
              int R, int S, int stride, int pad, int transposed, int relu, float drop_p, unsigned long long drop_seed,
              int stats_mode /* 0: stats = float slab [mtiles][2][N]; 1: stats = fixed-point accumulator (see vqa_bn_apply_acc) */,
              hipStream_t stream);
/* Data gradient of a Linear whose INPUT was relu(+dropout(p)) of the previous Linear, with the consumer's mask applied by the producer
 * (round 4): dx[M][Kin] = (dz[M][N] * wt[Kin][N]^T + addend) * (outact > 0) / (1 - drop_p); the keep scale multiplies the
 * value already rounded to the compute dtype, so dx is bit-equal to vqa_igemm followed by vqa_bias_act_bwd(outact, p).  Replaces the
 * autograd nodes of nn.ReLU + nn.Dropout between two nn.Linear (models/text_encoder.py:309-317 FeedForward, models/cross_attention.py:257-263,
 * models/vqa_model.py:74-82 AnswerHead).  wt = the [Kin][N] transposed weight (vqa_pack_transpose).  outact required. */
int vqa_linear_dgrad_act(int dtype, const void* dz, const void* wt, void* dx, const void* addend, const void* outact, float drop_p,
                         int M, int Kin, int N, hipStream_t stream);
/* dw[N][Kw] += dy[M][N]^T * gather(x)[M][Kw], split over the M pixels.
 * With a workspace (`ws`, caller-owned scratch of >= the ws_floats vqa_wgrad_plan reports; contents undefined afterwards) every
 * split writes its fp32 tile to its own slab and a second launch adds the slabs to dw in a FIXED order: dw is bit-reproducible,
 * no float atomics.  ws == NULL (or too small): fp32 atomics.  vqa_wgrad_plan is a host-only query: *kind 1 = the 8-wave LDS-DMA
 * kernel (bf16, the large CNN convs), 0 = the 4-wave kernel; tile, split count, workspace floats (0 = single split, no reduce). */
int vqa_wgrad_plan(int dtype, int loader, int M, int N, int Kw, int B, int H, int W, int C, int R, int S, long long* ws_floats,
                   int* kind, int* tile_n, int* tile_k, int* nsplit);
int vqa_wgrad(int dtype, int loader, const void* dy, const void* x, float* dw, int M, int N, int Kw, int B, int H, int W,
              int C, int Ho, int Wo, int R, int S, int stride, int pad, float* ws, long long ws_floats, hipStream_t stream);
int vqa_pack_rows(int dtype, const float* in, void* out, int N, int K, int Kp, hipStream_t stream);          /* cast + row pad   */
int vqa_pack_transpose(int dtype, const float* in, void* out, int N, int T, int C, int ldo, int col0, int flip, hipStream_t stream); /* out[c][col0+t*N+n] = in[n][flip?T-1-t:t][c] */
int vqa_pack_transpose_batch(int dtype, const float* flat, void* out, const long long* desc, int nd, int total_blocks, hipStream_t stream); /* nd pieces, device table desc[nd][10] = {src_off, dst_off, N, T, C, ldo, col0, flip, blk0, 0}, blk0 = running sum of T*ceil(N/32)*ceil(C/32): every data-gradient operand of a step in one launch */
/* Eval-mode Conv+BN folding, all convs in one launch (inference path, api/inference.py:196-323 via VQAModel.predict):
 * w'[n][k] = w[n][k]*gamma[n]/sqrt(running_var[n]+eps) cast to dtype, b'[n] = beta[n] - running_mean[n]*scale[n].
 * desc: device table [nd][10] int64 {w_off, gamma_off, beta_off (floats from flat), running_mean ptr, running_var ptr, N, K,
 * dst_off (elements of wout), bias_off (floats of bout), blk0}.  vqa_igemm's relu argument: 1 = ReLU before the addend,
 * 2 = ReLU after the addend (relu(conv + bias + residual)). */
int vqa_fold_bn_batch(int dtype, const float* flat, void* wout, float* bout, const long long* desc, int nd, int total_blocks, float eps, hipStream_t stream);
/* stage-1 3x3/1 conv, 64->64 channels, bf16, LDS-resident input patch (models/cnn_backbone.py:182-187 at Cin=Cout=64):
 * forward (w = [Cout][R][S][Cin]) and data gradient (w = flipped+transposed pack, out += addend*(addmask>0)); weight gradient. */
int vqa_conv3x3_c64_blocks(int B, int H, int W);
int vqa_conv3x3_c64(const void* x, const void* w, void* out, float* stats, const void* addend, const void* addmask,
                    int B, int H, int W, hipStream_t stream);
/* same conv (forward, or data gradient with the flipped pack) without epilogue inputs: 8-wave persistent kernel, 8 output rows per
 * block, input patches by LDS-DMA, weights in registers; stats [vqa_conv3x3_c64p_blocks][2][64] or NULL.  H % 8 == 0, W % 8 == 0. */
int vqa_conv3x3_c64p_blocks(int B, int H, int W);
/* ... and the data gradient of a residual block's conv1 (w = flipped + transposed pack) with the identity path in the epilogue:
 * out = (conv + addend * (addmask > 0)) * (outmask > 0) on the bf16 conv value, as vqa_igemm / vqa_conv8p; addend required, masks or NULL. */
/* ... and conv2's data gradient that also leaves bn1's BatchNorm-backward column sums (as vqa_conv8p's bn_y / bn_coef / bn_facc): the
 * stored tile is the gradient entering relu(BatchNorm(bn_y)); bn_coef = coef[4][64], bn_facc = zeroed vqa_bn_acc_words(3, 64). */
int vqa_conv3x3_c64p_bnred(const void* x, const void* w, void* out, const void* bn_y, const float* bn_coef, unsigned long long* bn_facc,
                           int B, int H, int W, hipStream_t stream);
int vqa_conv3x3_c64p_epi(const void* x, const void* w, void* out, const void* addend, const void* addmask, const void* outmask,
                         int B, int H, int W, hipStream_t stream);
int vqa_conv3x3_c64p(const void* x, const void* w, void* out, float* stats, int B, int H, int W,
                     int stats_mode /* 1: stats is the fixed-point accumulator u64 [2*64 + 1] of vqa_bn_apply_acc */, hipStream_t stream);
/* Round 4 -- the 256 x 256 x 64 8-phase GEMM core (csrc/gemm8p.hip): C[M][N] = A[M][K] . B[N][K]^T, bf16 operands, fp32 accumulation,
 * bf16 result; M % 256 == 0, N % 256 == 0, K % 64 == 0.  The dense form of the tile that vqa_conv8p runs as an implicit GEMM. */
int vqa_gemm8p(const void* A, const void* B, void* C, int M, int N, int K, hipStream_t stream);
/* the same product on the four-wave kernel (one wave per SIMD, 128 x 128 wave tiles in the accumulator registers; same shape rules) */
int vqa_gemm4w(const void* A, const void* B, void* C, int M, int N, int K, hipStream_t stream);
/* vqa_conv8p: the same tile as an implicit-GEMM 3x3 / stride 1 / pad 1 convolution over NHWC bf16 (models/cnn_backbone.py:182-187 at
 * 256 / 512 channels): out[B*H*W][N] = conv(x [B][H][W][C], w [N][(r, s, c)]) with transposed = 0; with transposed = 1 the stride-1 data
 * gradient (x = dy [B][H][W][Cout], w = the [Cin][(tap, Cout)] pack of vqa_pack_transpose, taps mirrored).  Tiles hold 196 valid output
 * pixels of 224 when 196 divides B*H*W (one 14 x 14 image / four 7 x 7 images: exactly 2 / 1 rounds of 256 CUs at B = 512).
 * stats: NULL, or the fixed-point BatchNorm accumulator (vqa_bn_acc_words(2, N), caller-zeroed) for sum y | sum y^2 of the stored values.
 * vqa_conv8p_ok: 1 when the shape is taken (C a power-of-two multiple of 64, N a multiple of 256). */
int vqa_conv8p_ok(int B, int H, int W, int C, int N);
int vqa_conv8p(const void* x, const void* w, void* out, unsigned long long* stats,
               const void* addend, const void* addmask, const void* outmask /* [B*H*W][N] bf16 or NULL: out = (conv + addend * (addmask > 0)) * (outmask > 0) */,
               const void* bn_y, const float* bn_coef, unsigned long long* bn_facc /* all or none: also the BatchNorm-backward column sums of the stored
                  tile as the gradient entering relu(BatchNorm(bn_y)): sum g | sum g*xhat, g = out * [bn_y*scale + shift > 0], into bn_facc
                  (vqa_bn_acc_words(3, N)); the caller then skips vqa_bn_bwd_reduce and runs vqa_bn_bwd_apply_acc on it */,
               int bn_selfmask /* 1: as above (ReLU directly behind that BatchNorm, mask recomputed from bn_y); 0: g = out, the tile already carries
                  its mask (outmask = the output of the block whose bn2 this is) */,
               const void* bn_y2, const float* bn_coef2 /* both or none, bn_selfmask 0 only: the 1x1 shortcut's BatchNorm sharing g -> third row
                  sum g*xhat(bn_y2), as vqa_bn_bwd_reduce with y2 / coef2 */,
               int B, int H, int W, int C, int N, int transposed, int stride /* 1, or 2: forward 3x3 / 2 convs (H, W = input map) */, hipStream_t stream);
/* Round 4 -- training-mode "Conv3x3 + BN + ReLU" without the normalised tensor (models/cnn_backbone.py:182-187: conv1 -> bn1 -> relu ->
 * conv2 at 64 channels).  vqa_conv3x3_c64p_bn is vqa_conv3x3_c64p applied to relu(BatchNorm(y)): y = the previous conv's raw output,
 * acc = its fixed-point statistics (the launch that wrote y ran with stats_mode = 1).  Every workgroup finalizes the 64 coefficients in
 * its prologue, workgroup 0 publishes coef_out[4][64] (scale | shift | mean | invstd) and updates the running statistics (what
 * vqa_bn_apply_acc does), and each input patch is normalised + ReLU'd IN LDS after its DMA landed -- zero padding stays zero; the conv
 * result is bit-identical to vqa_conv3x3_c64p(vqa_bn_apply_acc(y)).  vqa_wgrad3x3_c64_bn is the matching weight gradient
 * dw += dy^T gather(relu(y * coef[c] + coef[64 + c])) on the 8-wave kernel (vqa_wgrad3x3_c64_bn_ok: 1 when it takes the shape). */
int vqa_conv3x3_c64p_bn(const void* y, const unsigned long long* acc, const float* gamma, const float* beta, float* running_mean,
                        float* running_var, long long* num_batches_tracked, float* coef_out, const void* w, void* out, float* stats,
                        int B, int H, int W, int stats_mode, double count, float momentum, float eps, hipStream_t stream);
int vqa_wgrad3x3_c64_bn_ok(int B, int H, int W);
int vqa_wgrad3x3_c64_bn(const void* y, const float* coef, const void* dy, float* dw /* [64][576] += */, int B, int H, int W,
                        float* ws /* vqa_wgrad3x3_c64_blocks * 64*576 floats */, long long ws_floats, hipStream_t stream);
/* stage-2 weight gradient (3x3 / 1 / pad 1, 128 -> 128 channels, 28 x 28 maps, bf16; models/cnn_backbone.py:182-187 backward): 8-wave
   LDS-DMA kernel + fixed-order slab reduce.  vqa_wgrad3x3_c128_blocks: slabs of 128*576 floats the workspace must hold, 0 = shape not
   supported (the caller uses vqa_wgrad). */
int vqa_wgrad3x3_c128_blocks(int B, int H, int W);
int vqa_wgrad3x3_c128(const void* x, const void* dy, float* dw, int B, int H, int W, float* ws, long long ws_floats, hipStream_t stream);
int vqa_wgrad3x3_c64_blocks(int B, int H, int W);   /* [64][576] slabs of scratch vqa_wgrad3x3_c64 wants for this shape (0: unsupported) */
int vqa_wgrad3x3_c64(const void* x, const void* dy, float* dw /* [64][576] += */, int B, int H, int W,
                     float* ws /* >= vqa_wgrad3x3_c64_blocks * 64*576 floats of scratch, or NULL: 4-wave kernel with atomics */,
                     long long ws_floats, hipStream_t stream);
/* Up to 8 Linear weight gradients dw_j[N_j][K_j] += dy_j[M_j][N_j]^T x_j[M_j][K_j] in ONE launch + ONE fixed-order reduce launch
   (training/train.py:196 loss.backward() -> the token-side nn.Linear weights of models/text_encoder.py, cross_attention.py, fusion.py,
   answer_head.py).  Each dw_j is bit-identical to its own vqa_wgrad call.  vqa_wgrad_group_ws: floats of the shared workspace, or -1
   when a job is not one the planner gives the 4-wave 128x128 split kernel (the caller then uses vqa_wgrad). */
long long vqa_wgrad_group_ws(int dtype, int njobs, const int* M, const int* N, const int* Kw);
int vqa_wgrad_group(int dtype, int njobs, const void* const* dy, const void* const* x, float* const* dw, const int* M, const int* N,
                    const int* Kw, float* ws, long long ws_floats, hipStream_t stream);
/* second pass of the deterministic split weight gradients: dw[i] += sum_s ws[s][i] in slab order (n % 4 == 0) */
int vqa_slab_reduce(const float* ws, float* dw, int nslabs, long long n, hipStream_t stream);
/* data gradient of a stride-2 conv (+ the block's 1x1/2 shortcut, models/cnn_backbone.py:243-247) in one launch; rows are
 * grouped by output parity class so only the valid taps are issued.  dy/dyd [B][H][W][C], out [B][Ho=2H][Wo=2W][N] */
int vqa_dgrad_s2(int dtype, const void* dy, const void* dyd, const void* wt, void* out, int B, int H, int W, int C,
                 int Ho, int Wo, int N, int R, int pad, hipStream_t stream);
/* dedicated bf16 stem conv 7x7/2 (models/cnn_backbone.py:350): image patch + weights resident in LDS */
int vqa_stem_conv_blocks(int B, int H, int W);
int vqa_stem_pack(const float* w_krsc, void* wstem, hipStream_t stream);
int vqa_stem_conv(const float* img, const void* wstem, void* out, float* stats, int B, int H, int W, hipStream_t stream);
/* Inference stem in ONE launch: conv7x7/2 + BatchNorm with running statistics (coef = scale[64] | shift[64], vqa_bn_eval_coef) + ReLU +
   MaxPool3x3/2 p1 (models/cnn_backbone.py:349-354 in eval mode).  out: the pooled activation NHWC bf16 [B][Hp][Wp][64]; the conv
   output itself is never stored (no argmax either: no backward).  vqa_stem_conv_pool_ok: 1 when the shape is supported. */
int vqa_stem_conv_pool_ok(int B, int H, int W);
int vqa_stem_conv_pool(const float* img, const void* wstem, const float* coef, void* out, int B, int H, int W, hipStream_t stream);
int vqa_stem_wgrad_blocks(int B, int H, int W);   /* workgroups of the two stem weight-gradient kernels; their scratch: blocks * 64*147 floats */
int vqa_stem_wgrad(const float* img, const void* dy, float* dw /* [64][7][7][3] += */, int B, int H, int W,
                   float* ws /* scratch or NULL: atomics */, long long ws_floats, hipStream_t stream);
/* same, with the stem BN+ReLU+MaxPool backward apply fused in: dy is rebuilt per row from y, dpool, idx, coef, bcoef */
int vqa_stem_wgrad_fused(const float* img, const void* y, const void* dpool, const uint8_t* idx, const float* coef,
                         const float* bcoef, float* dw, int B, int H, int W, float* ws, long long ws_floats, hipStream_t stream);

/* ---- BatchNorm2d (nn.BatchNorm2d defaults; models/cnn_backbone.py:151,158,246,351) -----------------------------
 * coef = scale | shift | mean | invstd (4*C floats).  finalize also updates running_mean/var (momentum, unbiased var)
 * and num_batches_tracked when those pointers are non-NULL. */
int vqa_bn_stats_finalize(const float* part, int tiles, int C, double count, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, long long* num_batches_tracked, float momentum, float eps,
                          double* scratch /* >= 64*2*C */, float* coef, hipStream_t stream);
int vqa_bn_eval_coef(int C, const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                     float eps, float* coef, hipStream_t stream);
/* out = [relu](y*scale+shift [+ res | + res*rscale+rshift])   -- BN + residual add + ReLU (cnn_backbone.py:186-195) */
int vqa_bn_apply(int dtype, const void* y, const float* coef, const void* res, const float* rcoef, void* out,
                 long long numel, int C, int relu, hipStream_t stream);
/* the same for the LAST residual block of a stage, with the squeeze-excitation global-average-pool sums folded in: also writes the
   column sums of the stored values per (sample, row chunk) to part[B][vqa_bn_apply_pool_chunks()][C]; vqa_se_fwd(pool_part = part)
   folds them instead of re-reading the stage output (models/cnn_backbone.py:194-197 -> models/attention_modules.py:109-112) */
int vqa_bn_apply_pool_chunks(int dtype, int HW, int C);
int vqa_bn_apply_pool(int dtype, const void* y, const float* coef, const void* res, const float* rcoef, void* out, int B, int HW, int C,
                      int relu, float* part, hipStream_t stream);
/* Fixed-point statistics (round 3).  A producer launched with stats_mode = 1 (vqa_igemm, vqa_conv3x3_c64p) adds its per-workgroup
   fp32 partial sums  sum y | sum y^2  to acc[vqa_bn_acc_words(2, C)] (unsigned 64-bit, each sum split exactly into a 2^-4 plane and a 2^-50 plane, caller-zeroed; the flag
   word between the planes counts partials that were non-finite or beyond 2^41 -> NaN statistics; the total cannot wrap, csrc/common.h) with integer atomics: order-independent, hence bit-reproducible, and complete
   when the producer ends.  vqa_bn_apply_acc then does finalize (fp64, nn.BatchNorm2d training formulas, running-statistics update,
   coef_out [4][C] = scale | shift | mean | invstd for the backward) + apply (+ res | + BatchNorm(res) from racc, + ReLU) in ONE
   launch -- the finalize launches between conv and apply are gone.  pool_part != NULL: also the SE pooling sums (vqa_bn_apply_pool).
   Backward: vqa_bn_bwd_reduce(acc_mode = 1) / vqa_se_bwd(bn_acc_mode = 1) add  sum g | sum g*xhat | sum g*xhat2  (same format) to
   facc[vqa_bn_acc_words(3, C)]; vqa_bn_bwd_apply_acc derives the apply coefficients in its prologue and adds d gamma / d beta. */
int vqa_bn_acc_words(int K, int C);   /* 64-bit words of an accumulator for K sums x C channels: R = clamp(512/C, 1, 8) replicas
                                         (same-address atomics are serialised, ~21 ns each) + the flag word; K = 2 forward, 3 backward */
int vqa_bn_apply_acc(int dtype, const void* y, const unsigned long long* acc, const float* gamma, const float* beta, float* running_mean,
                     float* running_var, long long* num_batches_tracked, float* coef_out, const void* res,
                     const unsigned long long* racc, const float* rgamma, const float* rbeta, float* rrunning_mean, float* rrunning_var,
                     long long* rnum_batches_tracked, float* rcoef_out, void* out, int B, int HW, int C, int relu, double count,
                     float momentum, float eps, float* pool_part, hipStream_t stream);
int vqa_bn_bwd_blocks(long long rows);
int vqa_bn_bwd_reduce(int dtype, const void* dout, const void* outact, const void* y, const float* coef, const void* y2,
                      const float* coef2, float* slab /* [blocks][3][C], or the u64 accumulator [3*C + 1] when acc_mode = 1 */, long long rows,
                      int C, int self_mask /* mask = relu(bn(y))>0 from y */, int acc_mode, hipStream_t stream);
int vqa_bn_bwd_apply_acc(int dtype, const void* dout, const void* outact, const void* y, const unsigned long long* facc, const float* gamma,
                         const float* coef, float* dgamma, float* dbeta, void* dy, const void* y2, const float* gamma2, const float* coef2,
                         float* dgamma2, float* dbeta2, void* dy2, long long numel, int C, double count, int self_mask, hipStream_t stream);
int vqa_bn_bwd_finalize(const float* slab, int nblk, int C, int which, double count, const float* gamma, const float* coef,
                        int training, float* dgamma, float* dbeta, float* bcoef /* 3*C */, hipStream_t stream);
int vqa_bn_bwd_apply(int dtype, const void* dout, const void* outact, const void* y, const float* bcoef, void* dy,
                     const void* y2, const float* bcoef2, void* dy2, long long numel, int C, const float* mask_coef /* or NULL */,
                     hipStream_t stream);

/* ---- stem tail: BN + ReLU + MaxPool2d(3,2,1) fused (models/cnn_backbone.py:351-353) ----------------------------- */
int vqa_stem_pool_fwd(int dtype, const void* y, const float* coef, void* out, uint8_t* idx, int B, int H, int W, int C, hipStream_t stream);
int vqa_stem_bwd_reduce(int dtype, const void* dpool, const uint8_t* idx, const void* y, const float* coef, float* slab,
                        int B, int H, int W, int C, hipStream_t stream);
int vqa_stem_bwd_apply(int dtype, const void* dpool, const uint8_t* idx, const void* y, const float* coef, const float* bcoef,
                       void* dy, int B, int H, int W, int C, hipStream_t stream);

/* ---- SEAttention.forward (models/attention_modules.py:109-136) and its backward ---------------------------------- */
int vqa_se_fwd(int dtype, const void* x, const float* w1, const float* w2, float* pooled, float* hidden, float* scale,
               void* out, int B, int HW, int C, int Cr,
               const float* pool_part /* or NULL: pool x here */, int pool_chunks /* = vqa_bn_apply_pool_chunks */, hipStream_t stream);
/* bn_y / bn_coef / bn_slab (all or none): dx is the gradient entering the BatchNorm whose conv output is bn_y (the stage's last bn2);
   its backward column sums sum dx | sum dx*xhat(bn_y) are written to bn_slab[vqa_se_bwd_blocks()][3][C] (the layout
   vqa_bn_bwd_finalize reads) in the same pass, so the caller skips vqa_bn_bwd_reduce for that BatchNorm */
int vqa_se_bwd_blocks(int dtype, int B, int HW, int C);
long long vqa_se_bwd_scratch(int dtype, int B, int HW, int C, int Cr);   /* floats of `scratch` below */
/* bn_acc_mode 1: bn_slab is the u64 fixed-point accumulator (vqa_bn_acc_words(3, C)) instead of a float slab */
int vqa_se_bwd(int dtype, const void* dout, const void* x, const float* w1, const float* w2, const float* pooled,
               const float* hidden, const float* scale, float* scratch /* vqa_se_bwd_scratch() floats */, void* dx, float* dw1, float* dw2,
               int B, int HW, int C, int Cr, int mask_out /* dx *= (x > 0): x is a post-ReLU activation */,
               const void* bn_y, const float* bn_coef, float* bn_slab /* or the u64 accumulator when bn_acc_mode = 1 */, int bn_acc_mode,
               hipStream_t stream);
/* ---- SpatialAttention.forward (models/attention_modules.py:223-243) and its backward ----------------------------- */
int vqa_spatial_fwd(int dtype, const void* x, const float* w /* (1,2,7,7) */, float* pooled2, int* argmax, float* amap,
                    void* out, int B, int H, int W, int C, hipStream_t stream);
long long vqa_spatial_bwd_scratch(int B, int H, int W);   /* floats of `scratch` below (3*B*H*W + the conv-weight partial sums) */
int vqa_spatial_bwd(int dtype, const void* dout, const void* x, const float* w, const float* pooled2, const int* argmax,
                    const float* amap, float* scratch /* vqa_spatial_bwd_scratch() floats */, void* dx, float* dw, int B, int H, int W, int C,
                    hipStream_t stream);
int vqa_nhwc_to_nchw(int dtype, const void* in, float* out, int B, int HW, int C, hipStream_t stream);   /* aux['image_features'] */
int vqa_nchw_to_nhwc(int dtype, const float* in, void* out, int B, int HW, int C, hipStream_t stream);

/* ---- token side ------------------------------------------------------------------------------------------------------
 * embedding*sqrt(d) + sinusoidal PE + dropout (models/text_encoder.py:504-510,112-114); padding_idx 0 gets no gradient */
int vqa_embed_fwd(int dtype, const long long* ids, const float* emb, const float* pe, void* out, int rows, int L, int D, int V,
                  float scale, float p, unsigned long long seed, hipStream_t stream);
int vqa_embed_bwd(int dtype, const long long* ids, const void* dout, float* demb, int rows, int D, int V, float scale, float p,
                  unsigned long long seed, hipStream_t stream);
/* nn.LayerNorm(eps 1e-5) (+dropout, + addrow[row % period]: ImageFeatureProjector, models/fusion.py:98-112) */
int vqa_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* out, float* mean_rstd, int rows, int D,
                      float eps, float p, unsigned long long seed, const float* addrow, int period, hipStream_t stream);
/* Fixed-order reductions.  The entries below that reduce over workgroups take `ws` (uninitialised float scratch, sized by the
 * matching *_ws query): every workgroup stores its partial sums to its own row and a second launch issued by the same entry
 * folds the rows in index order (bit-reproducible run to run).  With ws == NULL they fall back to float atomics (same value up
 * to rounding order). */
long long vqa_layernorm_bwd_ws(int dtype, int rows, int D, int period /* 0 when dadd == NULL */);
int vqa_layernorm_bwd(int dtype, const void* dout, const void* x, const float* gamma, const float* mean_rstd, const void* addend,
                      void* dx, float* dgamma, float* dbeta, int rows, int D, float p, unsigned long long seed, float* dadd,
                      int period, float* ws, int defer_fold /* 1: skip the fold launch(es), the caller runs vqa_fold_group later */,
                      hipStream_t stream);
/* folds of a deferred call: returns their number, 5 values per fold in out[10]: {ws offset (floats), rows, row stride, columns, n0};
   fold 0 -> (dgamma | dbeta) split at n0 = D, fold 1 (position-embedding sum) -> dadd */
int vqa_layernorm_bwd_folds(int dtype, int rows, int D, int period, long long* out);
/* softmax(QK^T/sqrt(hd) [keys with kmask==0 -> -inf]) (dropout) V, one wave per (batch, head)
 * (models/text_encoder.py:237-258, models/cross_attention.py:176-198); probs = softmax before dropout [B][H][Lq][Lk] */
int vqa_attention_fwd(int dtype, const void* q, const void* k, const void* v, int ldq, int ldk, int ldv, const float* kmask,
                      float* probs, void* ctx, int ldc, int B, int H, int Lq, int Lk, int hd, float p, unsigned long long seed,
                      hipStream_t stream);
/* bf16 MFMA variant (Lq <= 32, Lk <= 64, hd 32|64): QK^T and PV as v_mfma_f32_32x32x16_bf16 tiles, softmax in registers */
int vqa_attention_fwd_mfma(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv, const float* kmask, float* probs,
                           void* ctx, int ldc, int B, int H, int Lq, int Lk, int hd, float p, unsigned long long seed, hipStream_t stream);
int vqa_attention_bwd(int dtype, const void* dctx, int ldc, const void* q, const void* k, const void* v, int ldq, int ldk, int ldv,
                      const float* probs, void* dq, void* dk, void* dv, int lddq, int lddk, int lddv, int B, int H, int Lq, int Lk,
                      int hd, float p, unsigned long long seed, hipStream_t stream);
/* bf16 MFMA form of vqa_attention_bwd (Lq <= 32, Lk <= 64, hd in {32, 64}, row strides multiples of 8); same arguments minus dtype */
int vqa_attention_bwd_mfma(const void* dctx, int ldc, const void* q, const void* k, const void* v, int ldq, int ldk, int ldv,
                           const float* probs, void* dq, void* dk, void* dv, int lddq, int lddk, int lddv, int B, int H, int Lq, int Lk,
                           int hd, float p, unsigned long long seed, hipStream_t stream);
/* Device-side accuracy counters: counters[3] (u64) += {top-1 correct, top-5 correct, samples} for fp32 logits [B][N] and i64 targets.
 * Replaces the argmax/topk + .cpu() + .item() of VQAAccuracy.update (utils/metrics.py:55-94); ties resolve to the lowest index. */
int vqa_accuracy_update(const float* logits, const long long* targets, unsigned long long* counters, int B, int N, hipStream_t stream);
/* masked mean over tokens (models/fusion.py:303-313, models/text_encoder.py:522-527) */
/* both masked means of the fusion tail in one launch (round 4): out[B][2D] = [mean_m(x0) | mean_m(x1)] with the same mask, and the
 * matching backward dx{0,1}[b][l][:] = dcat[b][{0,D}:] * m[b][l] / cnt (models/fusion.py:281-296); per-element arithmetic of
 * vqa_masked_pool_fwd / _bwd (bit-equal). */
int vqa_masked_pool_pair_fwd(int dtype, const void* x0, const void* x1, const float* mask, void* out, int B, int L, int D, hipStream_t stream);
int vqa_masked_pool_pair_bwd(int dtype, const void* dcat, const float* mask, void* dx0, void* dx1, int B, int L, int D, hipStream_t stream);
int vqa_masked_pool_fwd(int dtype, const void* x, const float* mask, void* out, int ldo, int col0, int B, int L, int D, hipStream_t stream);
int vqa_masked_pool_bwd(int dtype, const void* dpool, int ldo, int col0, const float* mask, const void* addend, void* dx,
                        int B, int L, int D, hipStream_t stream);
/* GatingMechanism.forward (models/fusion.py:160-166): fused = g*att + (1-g)*txt, g = sigmoid(z), cat = [att|txt] */
int vqa_gate_fwd(int dtype, const void* z, const void* cat, void* fused, int B, int D, hipStream_t stream);
int vqa_gate_bwd(int dtype, const void* dfused, const void* z, const void* cat, void* dz, void* dcat, int B, int D, hipStream_t stream);
int vqa_add(int dtype, const void* a, const void* b, void* out, long long n, hipStream_t stream);
/* gradient at the pre-activation of linear(+bias)(+ReLU)(+dropout); dbias += column sums */
long long vqa_bias_act_bwd_ws(int dtype, int M, int N);
int vqa_bias_act_bwd(int dtype, const void* dout, const void* outact, void* dz, float* dbias, int M, int N, float p,
                     unsigned long long seed, float* ws, int defer_fold, hipStream_t stream);
int vqa_bias_act_bwd_fold_rows(int dtype, int M, int N);   /* deferred fold: {offset 0, this many rows, stride N, N columns, n0 = N} -> dbias */
/* the deferred folds of any number of calls in one launch per 48 jobs: dst0_j[c] (c < n0_j) / dst1_j[c - n0_j] += sum_r part_j[r*stride_j + c] */
int vqa_fold_group(int njobs, const float* const* part, const int* nrows, const long long* stride, const int* ncols, float* const* dst0,
                   const int* n0, float* const* dst1, hipStream_t stream);
/* nn.CrossEntropyLoss() mean (training/train.py:120): loss += mean NLL, dlogits = (softmax-onehot)*gscale/B.
   err (device int, may be NULL): += number of rows whose target is outside [0, N) -- the reference raises there; such rows are
   never read out of bounds, they add NaN to the loss and get a NaN gradient row. */
int vqa_cross_entropy(int dtype, const void* logits, const long long* targets, float* loss, void* dlogits, float* logits_f32,
                      int B, int N, float gscale, int* err, float* ws /* B floats, or NULL: float atomics on *loss */, hipStream_t stream);
int vqa_convert(int dtype_in, int dtype_out, const void* in, void* out, long long n, hipStream_t stream);
/* clip_grad_norm_(max_norm) + AdamW over flat fp32 buffers (training/train.py:204-208,127-132).
   skip (device int, may be NULL): when skip[0] != 0 the launch changes NOTHING (parameters, moments) and adds skip[0] to
   skipped[0], 1 to skipped[1] and 1 to skipped[2] (device int[3], may be NULL; [0] / [1] are the caller's to reset, [2] never) -- a
   step whose CrossEntropy saw an out-of-range target raises in the reference before optimizer.step(), so the model must survive it.
   calls: how many times the caller has launched vqa_adamw on this state, this launch included (>= 1).  Adam's step number
   t = calls - skipped[2] is formed ON THE DEVICE (bias corrections 1 - beta^t in double, like torch.optim.AdamW), so a skipped launch
   never advances it, whatever the host knows. */
int vqa_sumsq(const float* g, long long n, float* out /* >= 2049 floats: [0] result (bit-reproducible), rest scratch */, hipStream_t stream);
int vqa_adamw(float* p, const float* g, float* m, float* v, long long n, float lr, float beta1, float beta2, float eps,
              float weight_decay, long long calls, const float* sumsq, float max_norm, float gscale,
              const int* skip, int* skipped,
              void* p_bf16 /* or NULL: also write the bf16 copy of the updated parameters (the operand buffer of the next forward) */,
              hipStream_t stream);

/* ---- input pipeline on the GPU -----------------------------------------------------------------
 * vqa_image_normalize: torchvision ToTensor + Normalize of data/preprocess.py:34-35,117-121 -- uint8 HWC [B][H][W][3] ->
 * float32 NCHW, (u/255 - mean[c]) / std[c] in torch's operation order (bit-identical), optional per-sample horizontal flip
 * (flip[b] != 0; RandomHorizontalFlip, data/preprocess.py:73).  W % 4 == 0.
 * vqa_pack_tokens: Tokenizer.encode (utils/tokenizer.py:196-250) for a batch -- ragged vocabulary indices words[offsets[b] ..
 * offsets[b+1]) -> ids / mask int64 [B][L]: START + words + END, truncated to L with END forced onto the last slot, PAD after. */
/* vqa_image_resize: transforms.Resize((RH, RW)) of a PIL image (data/preprocess.py:70,90,118; api/inference.py:140-170), i.e.
 * PIL.Image.resize(BILINEAR) -- Pillow Resample.c restated bit-exactly (double-precision triangle weights scaled by the down-scale
 * factor, 22-bit fixed point, horizontal pass to a uint8 intermediate, then vertical) -- for a RAGGED batch: image i is uint8 HWC
 * [H[i]][W[i]][3] at in + in_off[i].  Optional RandomCrop window (crop_yx[2i], crop_yx[2i+1]) of size OH x OW inside the resized
 * RH x RW image (data/preprocess.py:70-71; NULL: no crop, then OH == RH and OW == RW) and per-sample horizontal flip (device
 * uint8[n] or NULL).  Outputs (either may be NULL): out_u8 [n][OH][OW][3] (what PIL returns) and out_nchw float32 [n][3][OH][OW] =
 * ToTensor + Normalize of it.  in_off / H / W / crop_yx are HOST arrays (they become kernel arguments, 32 images per launch);
 * `ws` is device scratch of vqa_image_resize_ws() bytes (coefficient tables + the horizontal intermediates). */
long long vqa_image_resize_ws(int n, const int* H, const int* W, int RH, int RW, int OW);
int vqa_image_resize(const uint8_t* in, const long long* in_off, const int* H, const int* W, const int* crop_yx, int n, int RH, int RW,
                     int OH, int OW, uint8_t* out_u8, float* out_nchw, const uint8_t* flip, float mean0, float mean1, float mean2,
                     float std0, float std1, float std2, void* ws, long long ws_bytes, hipStream_t stream);
/* vqa_image_color_jitter: transforms.ColorJitter (data/preprocess.py:77-82) of uint8 HWC images [B][H][W][3], fused with ToTensor +
 * Normalize.  On PIL images torchvision's adjustments are Pillow code -- ImageEnhance.Brightness / Contrast / Color =
 * Image.blend(black | rounded mean of the L band | L band, image, factor), hue = RGB -> HSV, H += delta (uint8 wrap), HSV -> RGB --
 * restated bit-exactly (libImaging Blend.c, Convert.c rgb2l / rgb2hsv_row / hsv2rgb: same float / double mix, no FMA contraction).
 * order: device uint8 [B][4], the permutation ColorJitter.forward draws (0 brightness, 1 contrast, 2 saturation, 3 hue; other values
 * are skipped).  factors: device float [B][4] = brightness, contrast, saturation factor and the hue DELTA int(hue_factor * 255) mod 256
 * as a float (the caller computes it in double, as torchvision does); NaN switches that adjustment off.  Outputs (either may be
 * NULL): out_u8 [B][H][W][3] (what PIL returns) and out_nchw float32 [B][3][H][W] = ToTensor + Normalize of it.
 * sums: device scratch of B 64-bit words (the L-band sums the contrast step needs; zeroed inside). */
int vqa_image_color_jitter(const uint8_t* in_hwc, const uint8_t* order, const float* factors, int B, int H, int W, uint8_t* out_u8,
                           float* out_nchw, float mean0, float mean1, float mean2, float std0, float std1, float std2,
                           unsigned long long* sums, hipStream_t stream);
int vqa_image_normalize(const uint8_t* in_hwc, float* out_nchw, const uint8_t* flip, int B, int H, int W, float mean0, float mean1,
                        float mean2, float std0, float std1, float std2, hipStream_t stream);
int vqa_pack_tokens(const int* words, const long long* offsets, long long* ids, long long* mask, int B, int L, int add_special,
                    int start_idx, int end_idx, int pad_idx, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif
