/*
 * lrcn_oracle.h -- CPU restatement of the arithmetic of the reference's LRCN hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker / the CPU baseline.  The product path is the HIP library declared in
 * include/lrcn.h and it never links or calls this code.
 *
 * PARITY UNPINNED.  The reference (ekinakyurek/Long-Term-Recurrent-Convolutional-NN) is
 * Julia 0.5/0.6 + Knet.jl; Julia is not installed here, Knet/AutoGrad are un-vendored and
 * un-pinned (lrcn.jl:1-3 does an unversioned Pkg.add), and the repository holds no tests,
 * golden tensors or saved weights.  This file therefore restates the published algorithm
 * of lrcn.jl line by line (citations below) and of the Knet 0.8.x primitives it calls
 * (sigm, tanh, logp, dropout, xavier, Adam/update!, conv4 mode=1, pool, relu, mat), and is
 * checked by (1) finite differences, (2) an independent torch-CPU autograd transcription
 * (tests/golden/make_golden.py -> tests/golden/ npz files), (3) analytic known answers
 * (zero-logit loss = ln V, which is what the reference's deck plots at epoch 0).
 *
 * Array convention: every matrix is dense COLUMN-MAJOR with exactly the reference's shapes
 * (an R x C matrix stores element (i,j) at i + j*R), as Julia does, so that a maintainer can
 * hand Julia arrays straight to these functions.  Token ids are 0-based here
 * (eos=0, bos=1, unk=2) = the reference's 1/2/3 (lrcn.jl:248-255) minus one.
 */
#ifndef LRCN_ORACLE_H
#define LRCN_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_CNNOUT 4096 /* lrcn.jl:28  const cnnout = 4096 */
#define ORC_EOS 0
#define ORC_BOS 1
#define ORC_UNK 2

/* The 9-tensor model of initweights (lrcn.jl:489-510), reference order and shapes:
 *   W1    (E+H1) x 4H1     b1   1 x 4H1      gate column blocks [forget|in|out|change] (lrcn.jl:531-534)
 *   W2    (2h+H2) x 4H2    b2   1 x 4H2      LSTM-2 input = [h1*Wproj (h) , x_cnn (h)]  (lrcn.jl:545-546)
 *   Wproj H1 x h           Wcnn 4096 x h     h = ceil(H2/2)                            (lrcn.jl:504-505)
 *   Wembed V x E           Wout H2 x V       bout 1 x V                                (lrcn.jl:506-508)
 * initweights sizes W2's input as hidden[end] (lrcn.jl:496-498), so 2h must equal H2 (H2 even). */
typedef struct {
    int E, H1, H2, V;
    float *W1, *b1, *W2, *b2, *Wproj, *Wcnn, *Wembed, *Wout, *bout;
} orc_model;

int64_t orc_param_count(int E, int H1, int H2, int V);
/* sizes[9] <- element counts of the 9 tensors in reference order. */
void orc_param_sizes(int E, int H1, int H2, int V, int64_t sizes[9]);

/* initweights (lrcn.jl:489-510): xavier uniform +-sqrt(2/(fanin+fanout)) drawn in double then cast,
 * zero biases, forget-gate bias (first H entries of b) = 1.  Julia's MersenneTwister stream is not
 * reproduced (only the distribution is specified by the reference); the generator is splitmix64. */
void orc_init_weights(orc_model *m, uint64_t seed);

/* lstm (lrcn.jl:528-538).  x: B x X, h,c: B x H, W: (X+H) x 4H, b: 1 x 4H; writes h_out,c_out (B x H).
 * If gates_out != NULL it receives the activated gates [f|i|o|g] (B x 4H). */
void orc_lstm(const float *W, const float *b, int X, int H, int B, const float *x, const float *h,
              const float *c, float *h_out, float *c_out, float *gates_out);

/* lrcn (lrcn.jl:540-551): one timestep.  state = {h1,c1,h2,c2} (B x H each, updated in place).
 * mask1 (B x E) / mask2 (B x H2) are the dropout multipliers (0 or 1/(1-p)) applied at lrcn.jl:542 / :547,
 * NULL = no dropout (pdrop = 0).  logits: B x V. */
void orc_lrcn_step(const orc_model *m, int B, float *h1, float *c1, float *h2, float *c2,
                   const float *x_cnn, const float *x_lstm, const float *mask1, const float *mask2,
                   float *logits);

/* loss (lrcn.jl:553-581).  feats: B x 4096.  tokens: T vectors of B ids laid out [T][B] (= sequence[t][i]).
 * The loop runs T+1 steps: step 0 input = embedding of bos, step t input = embedding of tokens[t-1],
 * targets tokens[0..T-1] then eos.  Returns -sum(logp target) / (norm_B * (T+1)) with norm_B the
 * reference's GLOBAL batchsize (lrcn.jl:564-568); pass norm_B = B for a single shard.
 * mask1: (T+1) blocks of B x E, mask2: (T+1) blocks of B x H2, or NULL.
 * If grads != NULL it receives d loss / d param in the 9 reference-shaped buffers (lossgradient, lrcn.jl:583),
 * overwritten (not accumulated). */
double orc_loss(const orc_model *m, const float *feats, const int32_t *tokens, int T, int B, int norm_B,
                const float *mask1, const float *mask2, orc_model *grads);

/* Per-step logits of the same forward pass (for parity probes): logits_out is (T+1) blocks of B x V. */
void orc_forward_logits(const orc_model *m, const float *feats, const int32_t *tokens, int T, int B,
                        float *logits_out);

/* Knet Adam()/update! defaults (lrcn.jl:394, 399-405): t is the 1-based step count AFTER increment.
 *   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g ; w -= lr * (m/(1-b1^t)) / (sqrt(v/(1-b2^t)) + eps) */
void orc_adam(float *w, const float *g, float *mom, float *var, int64_t n, int t, float lr, float beta1,
              float beta2, float eps);

/* generate + beam_search (lrcn.jl:585-678), exactly as SURVEY A.3: probabilities multiplied in linear float32,
 * no length normalisation, stable descending sorts (lower index wins ties), step 1 expands hypothesis 1 only,
 * stop test looks at the best beam only, depth <= nword+1.  feat: 1 x 4096 (already normalised by the caller
 * if wanted, lrcn.jl:597).  out_tokens receives the best sequence INCLUDING the leading bos (max nword+2 ids);
 * returns its length.  out_prob = its probability. */
int orc_beam_search(const orc_model *m, const float *feat, int K, int nword, int32_t *out_tokens,
                    float *out_prob);

/* ---- LRCN-1f: BASELINE configs[1] "1-layer LSTM" (SURVEY 8d).  This repo's definition, NOT reference code (the reference
 * hard-wires two layers): drop LSTM-1 and Wproj from lrcn() and feed dropout(hcat(x_lstm, x_cnn)) to ONE lstm of width
 * H = m->H1 = m->H2, then the same output layer.  The model lives in the W1 ((E+h+H) x 4H), b1, Wcnn (4096 x h), Wembed, Wout,
 * bout members of orc_model (h = ceil(H/2)); W2, b2, Wproj are not touched.  mask: (T+1) blocks of B x (E+h), or NULL. ---- */
void orc1_param_sizes(int E, int H, int V, int64_t sizes[9]); /* 0 for the absent tensors */
void orc1_init_weights(orc_model *m, uint64_t seed);
void orc1_step(const orc_model *m, int B, float *h, float *c, const float *x_cnn, const float *x_lstm, const float *mask,
               float *logits);
double orc1_loss(const orc_model *m, const float *feats, const int32_t *tokens, int T, int B, int norm_B, const float *mask,
                 orc_model *grads);
void orc1_forward_logits(const orc_model *m, const float *feats, const int32_t *tokens, int T, int B, float *logits_out);
int orc1_beam_search(const orc_model *m, const float *feat, int K, int nword, int32_t *out_tokens, float *out_prob);

/* ---- VGG-16 to fc7 (lrcn.jl:697-748) ---- */
/* convx (lrcn.jl:724): 3x3, pad 1, stride 1, cross-correlation (mode=1) + bias.
 * x: (W,H,Cin,N) col-major; w: (3,3,Cin,Cout) col-major; b: Cout; y: (W,H,Cout,N). relu!=0 fuses relux (:725). */
void orc_conv3x3(const float *x, int W, int H, int Cin, int N, const float *w, const float *b, int Cout,
                 int relu, float *y);
/* poolx (lrcn.jl:726): 2x2 max, stride 2. x: (W,H,C,N) -> y: (W/2,H/2,C,N). */
void orc_pool2(const float *x, int W, int H, int C, int N, float *y);
/* fcx (lrcn.jl:728): y = w*mat(x) .+ b.  w: O x K col-major, x: K x N, y: O x N. */
void orc_fc(const float *w, const float *b, int O, int K, int N, const float *x, int relu, float *y);

/* Weights in the order get_params_cnn (lrcn.jl:697-721) yields them: 13 conv (w,b) then fc6, fc7. */
typedef struct {
    const float *conv_w[13];
    const float *conv_b[13];
    const float *fc6_w, *fc6_b, *fc7_w, *fc7_b;
} orc_vgg;
extern const int orc_vgg_cout[13];
extern const int orc_vgg_pool_after[13];
/* convnet (lrcn.jl:733-748): x (S,S,3,N) preprocessed image -> feats N x 4096 col-major (the final transpose
 * at :746).  S must be 224 for fc6's 25088 inputs.  Op list = SURVEY A.4 (no relu after fc7). */
void orc_vgg_forward(const orc_vgg *v, const float *x, int S, int N, float *feats);

/* read_image_data's arithmetic tail (lrcn.jl:768-772) for an already decoded/resized/cropped uint8 image batch:
 * img: (C=3, Wd, Ht, N) interleaved uint8 as decoders give it, i.e. img[((n*Ht + y)*Wd + x)*3 + c];
 * mean[3] per-channel average (the reference's averageImage, reduced to per-channel means);
 * out: (224,224,3,N) col-major with the H<->W swap of :771, out(i,j,c,n) = img(y=j? ...) see .c */
void orc_preprocess_u8(const uint8_t *img, int S, int N, const float mean[3], float *out);

/* ---- ORC_EMULATE_BF16: the same restatement with bfloat16 rounding where the HIP library's bf16 arithmetic has it ----
 * orc_set_emulate_bf16(1) makes every function above round (to nearest even, orc_bf16_round) exactly where liblrcn_hip.so's
 * LRCN_BF16 LSTM path / bf16 VGG path stores or feeds a bf16 value; everything else (ORC_ACC accumulation, f32 cell state, f32 logits,
 * double softmax / loss, f32 gradients) is untouched.  Off (default) the oracle is bit-for-bit the one tests/golden pins.
 * Rounding points (kernel that has them in brackets; DESIGN.md section 2 repeats the list):
 *   every contraction     both operands bf16: shadow weights = bf16(f32 master), feats, h(t-1), x, dlogits, dz ... [gemm*.hip, lstm_fused.hip]
 *   embedding gather      bf16(bf16(Wembed[tok,:]) * multiplier)                                  [embed_gather_kernel]
 *   LSTM-2 input          bf16(bf16(h1*Wproj) * multiplier) | bf16(x_cnn * multiplier); x_cnn itself is f32   [concat_x2_kernel]
 *   LRCN-1f input         bf16(bf16(Wembed[tok,:]) * multiplier) | bf16(x_cnn * multiplier)                   [concat_x2_kernel]
 *   cell forward          gates accumulate in f32, c stays f32; the activated gates kept for the reverse pass and h are bf16  [lstm_fwd_kernel]
 *   loss head             logits f32, log-softmax f32/double, dlogits = bf16((p - onehot) * scale)            [softmax_xent*_kernel]
 *   cell backward         dz = bf16(.), dc f32, dh f32                                                        [lstm_bwd_kernel]
 *   dX of LSTM-2          bf16(dz2 * W2x'), then bf16(that * multiplier); d x_cnn sums the f32 products        [dx2_mask_reduce_kernel]
 *   dX of LSTM-1          f32 (not rounded); embedding scatter in f32
 *   image-embedding grad  bf16(d x_cnn), bf16(feats) into the contraction
 *   convx / poolx         bf16 inputs and filters, f32 accumulation starting from the f32 bias, bf16(relu(.)); max of bf16 values
 *   fcx                   bf16 operands; relu6's output bf16, fc7's output f32
 * Not emulated (functions keep their f32 meaning): Adam (f32 master weights), beam search, preprocessing. */
void orc_set_emulate_bf16(int on);
int orc_get_emulate_bf16(void);
float orc_bf16_round(float x);

/* Number of OpenMP threads the library will use / set it (e.g. to the container's CPU share). */
int orc_num_threads(void);
void orc_set_num_threads(int n);

#ifdef __cplusplus
}
#endif
#endif
