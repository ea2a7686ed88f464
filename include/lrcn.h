/*
 * lrcn.h -- C ABI of liblrcn_hip.so: the MI355X (gfx950) implementation of the LRCN hot path of
 * ekinakyurek/Long-Term-Recurrent-Convolutional-NN (reference file: lrcn.jl).
 *
 * The reference has no FFI; its seam is a handful of Julia functions over dense arrays.  Each entry point
 * below replaces one of them (cited per function) and takes the SAME arrays: dense COLUMN-MAJOR float32 with
 * the reference's shapes, so a Julia `ccall` passes KnetArray/Array pointers without copying.  All array
 * pointers are DEVICE pointers (hipMalloc'd, or obtained from lrcn_malloc) unless marked "host".
 *
 * Conventions
 *   - Token ids are int32 and 0-based: eos=0, bos=1, unk=2 (= the reference's 1,2,3, lrcn.jl:248-255, minus 1).
 *     tokens are laid out [T][B] (the reference's sequence[t][i]).
 *   - "params" / "grads" / "mom" / "var" are arrays of 9 device pointers in initweights order (lrcn.jl:489-510):
 *       [0] W1 (E+H1) x 4H1   [1] b1 1 x 4H1   [2] W2 (2h+H2) x 4H2   [3] b2 1 x 4H2   [4] Wproj H1 x h
 *       [5] Wcnn 4096 x h     [6] Wembed V x E [7] Wout H2 x V        [8] bout 1 x V          h = ceil(H2/2), 2h == H2
 *     gate column blocks are [forget | in | out | change] (lrcn.jl:531-534).
 *   - The caller owns every array it passes; the context owns only scratch.  No pointer is retained across
 *     calls except the VGG weights repacked (copied) by lrcn_vgg_load.
 *   - Every function returns 0 on success or a negative LRCN_E* code; lrcn_last_error() gives the message.
 *     Nothing aborts or throws across the boundary (the reference signals errors with Julia exceptions,
 *     lrcn.jl:395, 603; a binding turns non-zero into error(msg)).
 *   - A context is bound to one device and is not thread-safe.  Work is queued on the context's stream
 *     (lrcn_set_stream; default = the device's null stream) and a call returns without synchronising unless it
 *     hands back a host scalar (loss_host != NULL, lrcn_beam_search, lrcn_last_loss, lrcn_sync).
 *   - Arithmetic type: LRCN_F32 = exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) everywhere; LRCN_BF16 = bf16
 *     operands with fp32 accumulation in the GEMMs/convolutions, fp32 master weights, fp32 cell state,
 *     fp32 softmax/loss, fp32 Adam.
 */
#ifndef LRCN_H
#define LRCN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LRCN_CNNOUT 4096 /* lrcn.jl:28 */
#define LRCN_EOS 0
#define LRCN_BOS 1
#define LRCN_UNK 2
#define LRCN_MAX_T 28 /* captions longer than 28 tokens are skipped by the reference, lrcn.jl:353, 438 */

enum { LRCN_OK = 0, LRCN_EINVAL = -1, LRCN_ENOMEM = -2, LRCN_EHIP = -3, LRCN_ESTATE = -4 };
enum { LRCN_F32 = 0, LRCN_BF16 = 1, LRCN_FP8 = 2 }; /* LRCN_FP8: vgg_dtype only (BASELINE config 5) */

typedef struct lrcn_ctx lrcn_ctx;

typedef struct {
    int device;     /* HIP device ordinal */
    int E, H1, H2;  /* --embed, --hidden (lrcn.jl:39-40); H2 must be even (lrcn.jl:496-505) */
    int V;          /* vocabulary size incl. eos/bos/unk */
    int max_B;      /* largest per-call batch (rows on this device) */
    int max_T;      /* largest caption length T (<= LRCN_MAX_T); the loop runs T+1 steps */
    int lstm_dtype; /* LRCN_F32 | LRCN_BF16 */
    int vgg_dtype;  /* LRCN_F32 | LRCN_BF16 | LRCN_FP8 (conv2_2..conv5_3 in OCP e4m3 after lrcn_vgg_calibrate; the rest bf16) */
    int max_images; /* VGG batch capacity; 0 = no VGG in this context */
    int n_layers;   /* 0 or 2: the reference's two-layer LRCN-2f (lrcn.jl:540-551).  1: LRCN-1f (BASELINE configs[1] "1-layer
                     * LSTM", SURVEY 8d -- this repo's definition, the reference hard-wires two layers): LSTM-1 and Wproj are
                     * dropped and ONE LSTM of width H1 == H2 reads dropout(hcat(embedding, x_cnn)) at every step.  Its model
                     * keeps the 9-slot order with W1 (E+h+H) x 4H in slot 0, b1 in slot 1, slots 2..4 (W2, b2, Wproj) unused
                     * (NULL / size 0); state = {h, c}; lrcn_dropout.mask1 holds (T+1) blocks of B x (E+h), mask2 is unused. */
} lrcn_config;

/* ---- lifetime / plumbing ---- */
int lrcn_create(const lrcn_config *cfg, lrcn_ctx **out);
void lrcn_destroy(lrcn_ctx *ctx);
const char *lrcn_last_error(const lrcn_ctx *ctx); /* ctx may be NULL: last creation error */
int lrcn_set_stream(lrcn_ctx *ctx, void *hip_stream);  /* hipStream_t; NULL = null stream */
/* Cap the grids of the VGG convolution kernels at `cap` workgroups (0 = one per tile / CU): the kernels then walk their
 * tiles persistently.  The data-parallel step sets a cap below the CU count at small per-GPU batches so that the LSTM
 * stream's chain of small dependent launches finds idle CUs while the VGG forward of the next step runs beside it. */
int lrcn_vgg_set_wg_cap(lrcn_ctx *ctx, int cap);
/* The stream on which lrcn_loss_grad runs its weight / bias gradient GEMMs beside the reverse recurrences (rev 4; default: a stream of the
 * context's own).  Like lrcn_comm_set_stream: a host that runs other work on side streams hands in one that shares a hardware queue with
 * none of them (HIP multiplexes streams onto GPU_MAX_HW_QUEUES queues; two streams on one queue run in order). */
int lrcn_set_wg_stream(lrcn_ctx *ctx, void *hip_stream);
int lrcn_sync(lrcn_ctx *ctx);
int lrcn_malloc(void **dev_ptr, size_t bytes);
int lrcn_free(void *dev_ptr);
int lrcn_memcpy_h2d(void *dst_dev, const void *src_host, size_t bytes);
int lrcn_memcpy_d2h(void *dst_host, const void *src_dev, size_t bytes);
const char *lrcn_version(void);
/* ABI revision of this header.  It changes whenever a struct layout or an existing signature changes (lrcn_config gained its trailing
 * n_layers field at revision 2; revision 3 added the entry points marked "rev 3", revision 4 those marked "rev 4", revision 5 those marked
 * "rev 5": lrcn_avg_loss_batch, lrcn_profile_segment, lrcn_refresh_shadows_group).  A binding compiled
 * against another revision must refuse to run: lrcn_create reads sizeof(lrcn_config) bytes of the caller's struct. */
#define LRCN_ABI_VERSION 5
int lrcn_abi_version(void);
/* Options (rev 3).  Returns LRCN_EINVAL for an unknown option or a value outside its range.
 *   LRCN_OPT_FUSED_UPDATE (0 | 1, default 0): lrcn_train_step / lrcn_train_step_dp write the NEXT step's K-contiguous shadow weights
 *     (the bf16 / f32, direct and transposed copies every lossgradient otherwise makes from the f32 parameters first) from inside the Adam
 *     kernel, into a second set of shadow buffers, and the next loss / lossgradient / train step with the SAME nine parameter pointers
 *     skips its shadow pass.  Contract: between those two calls the caller does not write the parameter arrays except through this
 *     library (lrcn_init_weights, lrcn_adam_update* invalidate the shadows themselves) -- or says so with lrcn_params_touched().
 *   LRCN_OPT_DETERMINISTIC (0 | 1, default 0): every floating-point sum on the lossgradient route is taken in a fixed order (the
 *     embedding-gradient scatter, the bias column sums, the split-K contractions and the loss accumulator use ordered partial sums
 *     instead of float atomics): two calls on the same inputs give bit-identical gradients.  Slower.
 *   LRCN_OPT_CONV_CHUNK_BYTES (development knob; 0 = default 0xF0000000): input bytes above which a convolution is cut into launches of
 *     whole images (the direct-to-LDS kernels address their input with 32-bit offsets). */
enum { LRCN_OPT_FUSED_UPDATE = 1, LRCN_OPT_DETERMINISTIC = 2, LRCN_OPT_CONV_CHUNK_BYTES = 3 };
int lrcn_set_option(lrcn_ctx *ctx, int option, int64_t value);
/* Shadow weights of ONE gradient group (LRCN_GRAD_GROUPS order), made on `stream` from the caller's f32 parameters into the second shadow set
 * (rev 5; needs LRCN_OPT_FUSED_UPDATE = 1): what a host that updates the parameters ITSELF per group -- the sharded update: reduce-scatter ->
 * lrcn_adam_update_flat on its slice -> all-gather -- calls right after a group's all-gather, so that the shadow pass of the next step runs
 * beside the rest of the backward pass instead of at the head of the next lrcn_loss_grad.  Once all five groups of a step have been
 * refreshed the set becomes current, under the same contract as LRCN_OPT_FUSED_UPDATE (same nine pointers next call, no foreign writes).
 * The caller orders `stream` after the writes of the group's parameters and the context's stream after `stream`. */
int lrcn_refresh_shadows_group(lrcn_ctx *ctx, const float *const params[9], int group, void *stream);
/* The caller wrote parameter arrays itself (loaded a checkpoint, clipped, ...): the next call makes its shadow weights afresh (rev 3). */
int lrcn_params_touched(lrcn_ctx *ctx);

/* ---- model ---- */
/* Element counts of the 9 tensors (two-layer model), and for a given lrcn_config.n_layers (0 for the slots a model lacks). */
int lrcn_param_sizes(int E, int H1, int H2, int V, int64_t sizes[9]);
int lrcn_param_sizes_n(int n_layers, int E, int H1, int H2, int V, int64_t sizes[9]);
/* initweights (lrcn.jl:489-510): xavier-uniform +-sqrt(2/(fanin+fanout)), zero biases, forget-gate bias 1.
 * Julia's RNG stream is not reproducible; the generator is a counter-based hash keyed by (seed, tensor, index). */
int lrcn_init_weights(lrcn_ctx *ctx, float *const params[9], uint64_t seed);

/* lstm (lrcn.jl:528-538). x: B x X, h/c: B x H, W: (X+H) x 4H, b: 1 x 4H -> h_out/c_out: B x H (may alias h/c). */
int lrcn_lstm(lrcn_ctx *ctx, const float *W, const float *b, int X, int H, int B, const float *x, const float *h,
              const float *c, float *h_out, float *c_out);

/* lrcn (lrcn.jl:540-551): one timestep. state[4] = {h1,c1,h2,c2} (B x H, updated in place as s[1..4] are);
 * x_cnn: B x h, x_lstm: B x E, mask1 (B x E) / mask2 (B x H2): dropout multipliers or NULL; logits: B x V. */
int lrcn_step(lrcn_ctx *ctx, const float *const params[9], float *const state[4], int B, const float *x_cnn,
              const float *x_lstm, const float *mask1, const float *mask2, float *logits);

/* Dropout specification for loss/lossgradient. pdrop == 0: none (what average_loss uses, lrcn.jl:233).
 * pdrop > 0 and mask1 == NULL: masks are generated on device from (seed) [Philox-style counter hash];
 * mask1/mask2 != NULL: caller-supplied multipliers, (T+1) blocks of B x E / B x H2 (for parity runs). */
typedef struct {
    float pdrop;
    uint64_t seed;
    const float *mask1;
    const float *mask2;
} lrcn_dropout;

/* loss (lrcn.jl:553-581). feats: B x 4096. norm_B: the reference's global `batchsize` that the loss is divided by
 * (lrcn.jl:564-568): B for one device, the global batch under data parallelism.
 * loss_host (host double*, may be NULL): -sum logp / (norm_B*(T+1)). */
int lrcn_loss(lrcn_ctx *ctx, const float *const params[9], const float *feats, const int32_t *tokens, int T, int B,
              int norm_B, const lrcn_dropout *drop, double *loss_host);

/* lossgradient = grad(loss) (lrcn.jl:583): as lrcn_loss, plus d loss / d params into grads[9] (overwritten). */
int lrcn_loss_grad(lrcn_ctx *ctx, const float *const params[9], const float *feats, const int32_t *tokens, int T,
                   int B, int norm_B, const lrcn_dropout *drop, float *const grads[9], double *loss_host);

/* The body of average_loss's batch loop (lrcn.jl:452-475; rev 5 -- the entry point SURVEY 8(b) lists by this name): the forward pass of
 * one batch with pdrop = 0, its loss divided by the batch's OWN size (average_loss takes the batch size from the data, lrcn.jl:412, not
 * from the global `batchsize`): -sum logp / (B*(T+1)) into *loss_host.  Identical to lrcn_loss(..., norm_B = B, drop = NULL, ...); the
 * host averages the per-batch values over the split (train.py, lrcn.jl:477-485). */
int lrcn_avg_loss_batch(lrcn_ctx *ctx, const float *const params[9], const float *feats, const int32_t *tokens, int T, int B,
                        double *loss_host);

/* Gradient-ready events (new: lets a data-parallel host start the all-reduce of a gradient group while lossgradient's
 * backward is still running).  lrcn_loss_grad finalises the nine gradients in this order of GROUPS:
 *   0: Wout, bout (params 7, 8)   1: W2, b2 (2, 3)   2: Wproj, Wcnn (4, 5)   3: W1, b1 (0, 1)   4: Wembed (6)
 * and records an event on the context's stream after each.  lrcn_grad_group_wait makes `stream` (a hipStream_t) wait for
 * the event of `group` of the most recent lrcn_loss_grad / lrcn_train_step call. */
#define LRCN_GRAD_GROUPS 5
int lrcn_grad_group_wait(lrcn_ctx *ctx, int group, void *stream);

/* The loss of the most recent lrcn_loss/_loss_grad/_train_step call (synchronises). */
int lrcn_last_loss(lrcn_ctx *ctx, double *loss_host);

/* Per-step logits of the loss forward pass (parity probe): logits_out = (T+1) blocks of B x V. */
int lrcn_forward_logits(lrcn_ctx *ctx, const float *const params[9], const float *feats, const int32_t *tokens,
                        int T, int B, float *logits_out);

/* update!(param, gloss, optim) with one Adam() per tensor (lrcn.jl:394, 399-405; Knet defaults lr 1e-3,
 * beta1 0.9, beta2 0.999, eps 1e-8). step = 1-based count of this update. One launch for all 9 tensors. */
int lrcn_adam_update(lrcn_ctx *ctx, float *const params[9], const float *const grads[9], float *const mom[9],
                     float *const var[9], int step, float lr, float beta1, float beta2, float eps);
/* The same update restricted to the tensors of gradient group `group` (0 .. LRCN_GRAD_GROUPS-1, the groups of
 * lrcn_grad_group_wait: {Wout,bout}, {W2,b2}, {Wproj,Wcnn}, {W1,b1}, {Wembed}), enqueued on `hip_stream` (NULL = the
 * context's stream).  Lets a data-parallel host run each group's Adam as soon as that group's gradients are final
 * (and reduced) while the rest of the backward pass is still running: the backward reads the K-contiguous shadows made at
 * the start of lrcn_loss_grad, never the f32 parameters.  Call once per group with the same `step`. */
int lrcn_adam_update_group(lrcn_ctx *ctx, float *const params[9], const float *const grads[9], float *const mom[9],
                           float *const var[9], int group, int step, float lr, float beta1, float beta2, float eps,
                           void *hip_stream);

/* The same Adam arithmetic on ONE flat run of n floats (w, g, m, v: device pointers to n elements each), enqueued on `hip_stream` (NULL =
 * the context's stream) (rev 3).  For a data-parallel host that shards the update over the ranks: reduce-scatter of a gradient group -> this
 * call on the rank's 1/N slice of the flat parameter buffer (with the rank's own slices of the moments) -> all-gather of the parameters.
 * Same bytes on the wire as the all-reduce, 1/N of the update's HBM traffic per rank; the moments then exist only rank-sharded. */
int lrcn_adam_update_flat(lrcn_ctx *ctx, float *w, const float *g, float *m, float *v, int64_t n, int step, float lr, float beta1,
                          float beta2, float eps, void *hip_stream);

/* Body of train1's loop (lrcn.jl:369-394) on one device: lossgradient + update!.  feats: B x 4096. */
int lrcn_train_step(lrcn_ctx *ctx, float *const params[9], float *const grads[9], float *const mom[9],
                    float *const var[9], const float *feats, const int32_t *tokens, int T, int B, int norm_B,
                    const lrcn_dropout *drop, int step, float lr, float beta1, float beta2, float eps,
                    double *loss_host);

/* ---- data parallelism over the GPUs of a node (new -- the reference is single-device).  One process (or thread) and
 * one context per GPU.  Rows of a global batch are split over the ranks; every rank passes the GLOBAL batch as norm_B, so the
 * sum of the ranks' gradients is exactly the single-device gradient of lrcn.jl:564-580, and an identical Adam step follows
 * everywhere.  Transport: RCCL over xGMI (librccl.so.1 is opened at run time), fp32 on the wire, one all-reduce(SUM) per gradient
 * group, issued on the context's own per-group streams as soon as that group's gradients are final. ---- */
#define LRCN_UNIQUE_ID_BYTES 128
/* Rank 0 creates the id (host buffer of LRCN_UNIQUE_ID_BYTES); the host program hands it to the other ranks by any channel. */
int lrcn_comm_unique_id(void *id_out);
/* LOCAL, non-collective check that lrcn_comm_init can be entered on this rank: librccl is loadable, every entry point resolves, the
 * context has no communicator yet (rev 3).  lrcn_comm_init is a collective: a rank that fails BEFORE entering it would leave the others
 * blocked inside it, so a host program probes on every rank first, agrees on the result, and only then calls lrcn_comm_init. */
int lrcn_comm_probe(lrcn_ctx *ctx);
/* Collective over the `world` contexts: binds ctx to rank `rank` of the communicator named by the id. */
int lrcn_comm_init(lrcn_ctx *ctx, int world, int rank, const void *unique_id);
int lrcn_comm_destroy(lrcn_ctx *ctx);
/* Sparse exchange of the embedding gradient (rev 4).  A rank's contribution to d Wembed is its (T+1) B rows of d(x_lstm) -- 1.5 MB at 32
 * rows per GPU against the 42.6 MB dense V x E gradient, which is also the one gradient group that becomes final LAST and whose all-reduce
 * therefore cannot hide behind the backward pass.  lrcn_set_embed_rows_buffer(rows, tok, capacity): from now on lrcn_loss_grad writes those
 * rows ((T+1) B x E, row-major, dropout multiplier applied) and their token ids into the caller's buffers and does NOT write grads[6]
 * (NULL, NULL, 0 turns it off).  The host all-gathers rows and ids over the ranks (rank order) and calls lrcn_embed_grad_from_rows on
 * every rank: the n_rows <= 8192 rows are summed per token in ONE fixed order (sorted (token, row) keys), so every rank obtains the same
 * bits an all-reduce would have delivered, as the dense V x E column-major gradient `grad_wembed`, on `hip_stream` (NULL = the context's). */
int lrcn_set_embed_rows_buffer(lrcn_ctx *ctx, float *rows, int32_t *tok, int capacity_rows);
int lrcn_embed_grad_from_rows(lrcn_ctx *ctx, const float *rows, const int32_t *tok, int n_rows, float *grad_wembed, void *hip_stream);
/* The stream on which the context issues every collective and every per-group update of lrcn_allreduce_grads / lrcn_train_step_dp (rev 4;
 * default: a stream of its own).  HIP multiplexes streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4) and two streams on one
 * queue run in order: a host that also runs the VGG forward of the next step on a side stream hands in an update stream that it has
 * checked NOT to share that side stream's queue (lrcn_amd/dp.py streams_share_a_queue), or a group's Adam waits for the whole forward. */
int lrcn_comm_set_stream(lrcn_ctx *ctx, void *hip_stream);
/* In-place all-reduce(SUM) of gradient group `group` (0 .. LRCN_GRAD_GROUPS-1; -1 = all groups) of the most recent lrcn_loss_grad:
 * the group's stream waits for its gradient-ready event, then runs the collective (a no-op without a communicator / with one rank).
 * Asynchronous; lrcn_comm_join makes the context's stream wait for everything issued on the group streams. */
int lrcn_allreduce_grads(lrcn_ctx *ctx, float *const grads[9], int group);
int lrcn_comm_join(lrcn_ctx *ctx);
/* The whole data-parallel step of SURVEY 8(b) in one call (body of train1's loop, lrcn.jl:369-394, on one rank's rows):
 *   [img_u8 != NULL: VGG-16 forward of this rank's B crops into feats (B x 4096, caller's buffer), optionally normalised (lrcn.jl:597)]
 *   -> lossgradient -> per gradient group: all-reduce(SUM) over the ranks -> Adam of that group, overlapped with the rest of the
 *   backward pass -> the context's stream joins.  img_u8 == NULL: feats is the input.  Without a communicator it is lrcn_train_step. */
int lrcn_train_step_dp(lrcn_ctx *ctx, float *const params[9], float *const grads[9], float *const mom[9], float *const var[9],
                       const uint8_t *img_u8, const float mean[3], int normalize, float *feats, const int32_t *tokens, int T, int B,
                       int norm_B, const lrcn_dropout *drop, int step, float lr, float beta1, float beta2, float eps,
                       double *loss_host);

/* generate + beam_search (lrcn.jl:585-678): feat 1 x 4096 (normalise beforehand if wanted, lrcn.jl:597).
 * out_tokens (host, >= nword+2 ints) receives the best hypothesis INCLUDING the leading bos; *out_len its length;
 * *out_prob its probability (linear float32 product, no length normalisation). */
int lrcn_beam_search(lrcn_ctx *ctx, const float *const params[9], const float *feat, int K, int nword,
                     int32_t *out_tokens, int *out_len, float *out_prob);
/* The same decode for N images at once (new: the reference decodes image by image): feats N x 4096 column-major, N*K <= max_B.
 * The N*K hypotheses are the rows of one batched lrcn() step; softmax, top-K, candidate ordering (stable, descending), history
 * update and the stop test run on the device.  Host outputs: out_tokens [N][nword + 2] (bos first), out_len [N], out_prob [N]
 * (may be NULL).  Per image identical to lrcn_beam_search.
 * From 256 hypotheses in bf16 the step takes its memory-for-FLOPs form: the context allocates, at the first such call,
 * a table of input-side pre-activations per TOKEN (V x 4 H1 f32: 170 MB at V = 10640, H = 1000) and per hypothesis row capacity
 * (max_B x 4 H2 f32), the gate GEMMs contract the hidden state only, and the N*K x V logits are reduced to per-tile records inside the
 * logits GEMM (they never reach memory).  Ranking is lrcn.jl:652-656's -- float32 probabilities, stable, lower column first in a tie group. */
#define LRCN_BEAM_MAXLEN 258
int lrcn_beam_search_batch(lrcn_ctx *ctx, const float *const params[9], const float *feats, int N, int K, int nword,
                           int32_t *out_tokens, int *out_len, float *out_prob);

/* ---- VGG-16 to fc7 (lrcn.jl:697-748) ---- */
/* get_params_cnn (lrcn.jl:697-721): conv_w[l] (3,3,Cin,Cout), conv_b[l] Cout, fc6_w 4096 x 25088, fc7_w 4096 x 4096
 * (the transposed `mat` of :712), biases 4096.  Weights are repacked (copied) into the context. */
int lrcn_vgg_load(lrcn_ctx *ctx, const float *const conv_w[13], const float *const conv_b[13], const float *fc6_w,
                  const float *fc6_b, const float *fc7_w, const float *fc7_b);
/* convnet (lrcn.jl:733-748): x (224,224,3,N) preprocessed -> feats N x 4096 (pre-ReLU fc7). */
int lrcn_vgg_forward(lrcn_ctx *ctx, const float *x, int N, float *feats);
/* read_image_data's arithmetic (lrcn.jl:766-772) on decoded 224x224 RGB uint8 crops img[n][row][col][c]:
 * out (224,224,3,N), out(i,j,c,n) = pixel(row i, col j, c) - mean[c].  mean: host float[3] (NULL after lrcn_set_average_image). */
int lrcn_preprocess_u8(lrcn_ctx *ctx, const uint8_t *img, int N, const float mean[3], float *out);
/* Both of the above fused (no (224,224,3,N) float round trip): the training-path entry. */
int lrcn_vgg_forward_u8(lrcn_ctx *ctx, const uint8_t *img, int N, const float mean[3], float *feats);
/* The same forward for the crops of SEVERAL training batches at once (rev 4): N = m * block_rows images in, m consecutive
 * block_rows x 4096 column-major feature arrays out (block b at feats + b * block_rows * 4096), each optionally divided by its row sums
 * (lrcn.jl:595-597).  The frozen extractor does not depend on the LSTM parameters, so a data-parallel rank whose own batch is small
 * (32 rows of a 256 batch on 8 GPUs) runs the convolutions for the next m steps in one forward at the efficiency of a large batch
 * (VGG alone on one MI355X: 32 images 1.13 ms = 880 TFLOP/s, 128 images 3.38 ms = 1172 TFLOP/s) and feeds one block per step. */
int lrcn_vgg_forward_u8_blocks(lrcn_ctx *ctx, const uint8_t *img, int N, const float mean[3], int block_rows, int normalize, float *feats);
/* The full VGG averageImage (lrcn.jl:113: vgg["meta"]["normalization"]["averageImage"], (224,224,3) column-major, device pointer;
 * copied).  Once set, the *_u8 entry points subtract it instead of mean[3] (mean may then be NULL), exactly where the reference does:
 * before its last H <-> W permutedims (lrcn.jl:770-771), i.e. pixel (row r, col q, c) meets averageImage(q, r, c).  NULL turns it off. */
int lrcn_set_average_image(lrcn_ctx *ctx, const float *average_image);
/* read_image_data's geometry on the device for a batch of decoded images of different sizes (lrcn.jl:755-765): resize so that
 * the shorter side is 224 and the other div(side * 224, shorter), centre crop with div offsets, grey -> 3 channels (alpha dropped).
 * src: device buffer with the N images back to back, image n at byte offsets[n], row-major [h][w][channels] uint8;
 * offsets / heights / widths / channels (1, 3 or 4): HOST arrays.  out: device uint8 crops [N][224][224][3] = the input of
 * lrcn_vgg_forward_u8 / lrcn_preprocess_u8.  Resampling is bilinear between pixel centres in exact integer arithmetic
 * (round half up); Images.imresize's own kernel is not pinned by the reference. */
int lrcn_resize_crop_u8(lrcn_ctx *ctx, const uint8_t *src, const int64_t *offsets, const int *heights, const int *widths,
                        const int *channels, int N, uint8_t *out);
/* ---- input feed (rev 4): the per-batch host -> device copy of the reference's training loop (lrcn.jl:369-376 uploads every batch's
 * inputs), moved off the critical path.  lrcn_upload_crops copies N decoded crops (uint8 [N][224][224][3], HOST memory; page-locked memory
 * -- lrcn_host_alloc -- makes the copy a true asynchronous DMA) on the context's own copy stream into one of THREE device staging buffers and
 * returns that buffer in *dev_out.  The pointer is accepted wherever device crops are (lrcn_vgg_forward_u8, lrcn_train_step_dp,
 * lrcn_vgg_calibrate): such a forward waits ON THE DEVICE for the upload, and a later upload into the same staging buffer waits for the
 * forward's first kernel (the only one that reads the crops).  Uploads may run at most three batches ahead of the forwards that consume them
 * (LRCN_ESTATE otherwise); the call returns at once unless the host is more than two batches ahead of the DEVICE: it then waits, on the
 * calling thread, until the forward that last read the staging buffer has started (a bound on the run-ahead, not a cost; the copy stream
 * never carries a device-side wait for a future event, which would stall compute streams sharing its hardware queue).  The host buffer must stay valid and unchanged until the copy has run:
 * lrcn_upload_wait blocks until every upload issued so far has finished (what a loader thread calls before refilling its buffer). */
int lrcn_host_alloc(void **host_ptr, size_t bytes);
int lrcn_host_free(void *host_ptr);
int lrcn_upload_crops(lrcn_ctx *ctx, const uint8_t *host_u8, int N, const uint8_t **dev_out);
int lrcn_upload_wait(lrcn_ctx *ctx);
/* feats (N x 4096 column-major, device) <- every row divided by its sum: generate's `input/sum(input)` (lrcn.jl:595-597) and
 * what the reference's training features (`featsn`, lrcn.jl:121-123) hold. */
int lrcn_normalize_features(lrcn_ctx *ctx, float *feats, int N);

/* Parity probes for the VGG operators (lrcn.jl:724-728), reference layouts, any small size:
 * x (W,H,Cin,N) -> y (W,H,Cout,N) [or (W/2,H/2,Cout,N) with pool]; Cin, Cout multiples of 32. */
int lrcn_conv3x3(lrcn_ctx *ctx, const float *x, int W, int H, int Cin, int N, const float *w, const float *b,
                 int Cout, int relu, int pool, float *y);

/* Parity probe of the product path's FIRST launch in bf16 (conv64f.hip: read_image_data's mean subtraction, conv1_1 + ReLU, conv1_2 + ReLU
 * and the 2x2 max-pool in one kernel; lrcn.jl:770 + 724-726 twice): img = N decoded crops [n][S][S][3] uint8 as lrcn_vgg_forward_u8 takes
 * them (device memory), S a multiple of 16, w11 (3,3,3,64) / w12 (3,3,64,64) and the biases in the reference layouts (device float) ->
 * y (S/2,S/2,64,N).  Rounds where the stack rounds: crops - mean, conv1_1's output, the pooled output (bf16), f32 accumulation. */
int lrcn_conv1_fused(lrcn_ctx *ctx, const uint8_t *img, int N, int S, const float mean[3], const float *w11, const float *b11,
                     const float *w12, const float *b12, float *y);

/* ---- fp8 convolution stack (BASELINE config 5; the reference has no reduced-precision path, lrcn.jl:724-728 is Float32) ----
 * vgg_dtype = LRCN_FP8: conv2_2 .. conv5_3 (Cin % 128 == 0, 79 % of the VGG FLOPs) run as v_mfma_f32_16x16x128_f8f6f4 on
 * OCP e4m3 operands: weights e4m3(w / sw[co]) with sw[co] = amax_co / 448, activations e4m3(x / sa) with one sa per layer,
 * f32 accumulation, epilogue e4m3(relu(acc * sa_in sw[co] / sa_out + b[co] / sa_out)) (+ fused 2x2 max-pool).  conv1_1,
 * conv1_2, conv2_1, fc6 and fc7 stay bf16.
 * lrcn_vgg_calibrate runs `img` (decoded uint8 crops, as lrcn_vgg_forward_u8) through the bf16 stack once, records every
 * layer's output amax and sets sa = margin * amax / 448 (margin in [1,16]; values above saturate at 448 sa).  It must be
 * called before the first forward (LRCN_ESTATE otherwise) and may be called again to recalibrate. */
int lrcn_vgg_calibrate(lrcn_ctx *ctx, const uint8_t *img, int N, const float mean[3], float margin);
/* Parity probe of one e4m3 layer, reference layouts as lrcn_conv3x3: x is quantised with sa_in, w per output channel, the
 * result is returned dequantised (e4m3 * sa_out).  Cin % 128 == 0, Cout >= 128 and % 16 == 0, N*W*H >= 256.
 * sw_out (device float[Cout], may be NULL) receives the weight scales so a caller can emulate the arithmetic exactly. */
int lrcn_conv3x3_fp8(lrcn_ctx *ctx, const float *x, int W, int H, int Cin, int N, const float *w, const float *b, int Cout,
                     int relu, int pool, float sa_in, float sa_out, float *y, float *sw_out);

/* ---- measurement (bench.py "roofline") ----
 * While enabled, every VGG forward brackets its 12 convolution launches (bf16: conv1_1+conv1_2 fused and conv2_1 on
 * conv64_kernel, conv2_2..conv5_3 on gemm8p_kernel<..,CONV3,..>; back to back on the context's stream) with a pair of HIP events.
 * lrcn_profile_get synchronises and returns the accumulated milliseconds and launch count since lrcn_profile(ctx,1). */
int lrcn_profile(lrcn_ctx *ctx, int enable);
int lrcn_profile_get(lrcn_ctx *ctx, double *conv_ms, int64_t *conv_launches);
/* Segment timing for the HBM-bound sub-reports of SURVEY 8(d) (rev 5).  lrcn_profile(ctx, 2) = as 1, and additionally a pair of HIP
 * events -- on the stream the work is launched on -- around each of these segments of every call, with the segment's ALGORITHMIC bytes
 * (SURVEY 8d's per-unit figures x the units processed) summed beside the time:
 *   LRCN_SEG_UPDATE        update! (lrcn.jl:394): 28 B per parameter (w, m, v read + written, g read)
 *   LRCN_SEG_REC_FWD / _BWD  the per-timestep recurrence launches of one layer pass (lrcn.jl:529 and its dual): one read of the
 *                          recurrent weight block (4H x H elements) per timestep
 *   LRCN_SEG_EMBED_GATHER  param[end-2][idx,:] (lrcn.jl:556, 569): (T+1) B rows of E elements read and written
 *   LRCN_SEG_EMBED_GRAD    its dual: (T+1) B rows of E f32 in, the dense V x E f32 gradient out
 *   LRCN_SEG_PREPROCESS    read_image_data's arithmetic (lrcn.jl:768-772): 1 B in + 1 activation element out per pixel value
 *   LRCN_SEG_UPLOAD        the per-batch H2D copy of train1 (lrcn.jl:369-376): N x 150528 bytes (PCIe, not HBM)
 * The event records sit between dependent launches and cost a few microseconds each: level 2 is for a separate, untimed pass.
 * lrcn_profile_segment synchronises the device and returns what accumulated since lrcn_profile(ctx, 2). */
enum { LRCN_SEG_UPDATE = 0, LRCN_SEG_REC_FWD = 1, LRCN_SEG_REC_BWD = 2, LRCN_SEG_EMBED_GATHER = 3, LRCN_SEG_EMBED_GRAD = 4,
       LRCN_SEG_PREPROCESS = 5, LRCN_SEG_UPLOAD = 6, LRCN_SEG_COUNT = 7 };
int lrcn_profile_segment(lrcn_ctx *ctx, int segment, double *ms, int64_t *brackets, double *algorithmic_bytes);
/* Diagnostic: average milliseconds of one bf16 3x3 convolution layer (N images of S x S x Cin -> Cout, optional fused
 * pool) on random data, `iters` back-to-back launches timed with HIP events.  Kernel-development aid, not the product path. */
int lrcn_bench_conv(lrcn_ctx *ctx, int N, int S, int Cin, int Cout, int pool, int iters, double *ms_out);
/* Kernel-development aid: with LRCN_STAMPS=1 in the environment, lrcn_bench_conv's launches of the phase-interleaved kernel record, per
 * output tile, the shader clock at the boundaries of its segments (8 x uint64 per tile: [0] tile start, [1] first K-tile's DMA issued,
 * [2] ... landed, [3] main loop done, [4] accumulators staged in LDS, [5] stores issued, [6] / [7] 100 MHz wall counter at tile end /
 * start).  lrcn_debug_stamps copies the first n values of the most recent launch to the host.  Not the product path. */
int lrcn_debug_stamps(lrcn_ctx *ctx, unsigned long long *host_out, int64_t n);
/* Same for one bf16 NT contraction C[M][N] = A[M][K] B[N][K]^T (K a multiple of 64, N of 8) through the library's dispatch. */
int lrcn_bench_gemm(lrcn_ctx *ctx, int M, int N, int K, int iters, double *ms_out);

/* Test / development aid: which kernel family did the work?  which = 0: the most recent contraction / convolution launched by
 * the calling thread ("8p:0" = phase-interleaved 256x256 tile, "8p:1" = 256x128, "8p:2" = 512x128, "8p-splitk:<slices>", "glds",
 * "skinny", "gemm_nt", "conv64", "conv64-fused11", ...); which = 1: the comma-separated routes of the layers of ctx's most recent
 * VGG forward (conv1_1 [+conv1_2], ..., conv5_3, fc6, fc7).  The string is valid until the next call that launches work. */
const char *lrcn_debug_route(lrcn_ctx *ctx, int which);

#ifdef __cplusplus
}
#endif
#endif
