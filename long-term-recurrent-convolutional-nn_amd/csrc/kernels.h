// kernels.h -- launchers of the non-GEMM kernels of liblrcn_hip (gfx950).  dtype: GEMM_T_F32 / GEMM_T_BF16 selects
// the element type "T" of activation/shadow buffers; everything marked f32 is always float.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct DropSpec {
    float p;            // drop probability (0 = identity)
    uint64_t seed;      // counter-hash key
    const float *mask;  // caller-supplied multipliers ((T+1) blocks of B x ncols, column-major) or NULL
    int which;          // 1 = the mask of lrcn.jl:542, 2 = the mask of lrcn.jl:547
};

// tok_in[s][b] = bos (s==0) | tokens[s-1][b];  tok_tgt[s][b] = tokens[s][b] (s<T) | eos.   (lrcn.jl:556,565,569,576)
void k_build_tokens(hipStream_t st, const int32_t *tokens, int T, int B, int V, int32_t *tok_in, int32_t *tok_tgt, double *zero_acc);

// Xemb[m][e] = WembT[tok_in[m]][e] * dropmask   (lrcn.jl:556/569 gather + :542 dropout), m = s*B+b.
void k_embed_gather(hipStream_t st, int dtype, const void *wembT, int64_t ld_w, const int32_t *tok_in, int S, int B,
                    int E, DropSpec d, void *xemb, int64_t ld_x);
// dWembed(tok, e) += dXemb[m][e] * dropmask   (AutoGrad dual of the gather; Wembed is V x E column-major f32).
void k_embed_scatter(hipStream_t st, const float *dxemb, int64_t ld_dx, const int32_t *tok_in, int S, int B, int E,
                     int V, DropSpec d, float *dwembed);

// The same through an E-contiguous staging array stage[V][ld_s] (f32, all zero on entry and on exit): coalesced atomics per token row,
// then one dense transpose that writes EVERY element of dwembed (no memset needed).  sort_keys != NULL ((T+1)*B <= 8192 keys of scratch):
// the rows of a token are added in row order by plain stores instead -- a fixed summation order (LRCN_OPT_DETERMINISTIC); false = too
// many rows for the one-workgroup sort, nothing was launched.
void k_embed_rows_export(hipStream_t st, const float *dxemb, int64_t ld_dx, int S, int B, int E, DropSpec d, float *out);
bool k_embed_scatter_rm(hipStream_t st, const float *dxemb, int64_t ld_dx, const int32_t *tok_in, int S, int B, int E, int V, DropSpec d,
                        float *stage, int64_t ld_s, float *dwembed, unsigned long long *sort_keys);

// LSTM cell, one timestep (lrcn.jl:531-536).  G f32 [B][4H] = pre-activations incl. bias; c_prev f32 [B][H] or NULL.
// Writes activated gates [f|i|o|g] (T), c_new (f32), h_new (T) and optionally h_new as f32.
void k_lstm_fwd(hipStream_t st, int dtype, const float *G, int64_t ld_g, const float *c_prev, int B, int H, void *acts,
                int64_t ld_a, float *c_new, void *h_new, int64_t ld_h, float *h_new_f32);
// Reverse of the cell.  dh_a (+ dh_b, may be NULL) f32 [B][H]; dc f32 [B][H] is read and replaced by
// dc_prev.  Writes dZ (T) [B][4H].
void k_lstm_bwd(hipStream_t st, int dtype, const void *acts, int64_t ld_a, const float *c_prev, const float *c_new,
                const float *dh_a, int64_t ld_dha, float *dh_b, int dh_b_read, float *dc, int dc_zero, int B, int H, void *dz,
                int64_t ld_dz, int nslab = 0);   // nslab > 0: dh_b = [nslab][B][H] partial sums, summed here

// Small-batch fused recurrent steps (lstm_fused.hip; bf16, B <= 64): one launch = the recurrent GEMM of a timestep + the cell
// update (forward) / the dh GEMM of step s + the cell backward of step s-1.
// batched beam search: histories = [bos, 0, ...], next input = bos, probabilities = 1 for all R hypotheses (lrcn.jl:608-611)
void k_beam_init(hipStream_t st, int32_t *seq, int32_t *last, float *p, int R, int Lh, int bos);
bool lstm_fused_eligible(int dtype, int B, int H, int64_t ldh, int64_t ld4);
hipError_t launch_lstm_rec_fwd(hipStream_t st, const void *h_prev, int64_t ldh, const void *Wh, const float *Gx, const float *c_prev, int B,
                               int H, void *acts, int64_t ld_a, float *c_new, void *h_new, const void *zero_page, bool alone = false);
hipError_t launch_lstm_rec_bwd(hipStream_t st, const void *dz_s, int64_t ld4, const void *WhT, const void *acts, const float *c_prev,
                               const float *c_new, const float *dh_ext, float *dc, int B, int H, void *dz_out, const void *zero_page,
                               bool alone = false);
// alone: nothing else runs on the GPU beside the LSTM step -- up to LRCN_LSTM_REC2 rows (default 64) the step kernels then take
// their 8-units-per-workgroup forms (125 / 250 workgroups instead of 63; lstm_fused.hip)

// X2[m][j<nl] *= mask ; X2[m][nl+j] = xcnn[b][j] * mask (j < nr); mask over nl+nr columns      (lrcn.jl:546-547: nl = nr = h;
// LRCN-1f: nl = E, nr = h)
void k_concat_x2(hipStream_t st, int dtype, void *x2, int64_t ld_x2, const float *xcnn, int64_t ld_xc, int S, int B,
                 int nl, int nr, DropSpec d);
// dX2[m][j] *= mask (all nl+nr columns, in place);  dxcnn[b][j] = sum_s dX2[s*B+b][nl+j]
void k_dx2_mask_reduce(hipStream_t st, int dtype, void *dx2, int64_t ld, int S, int B, int nl, int nr, DropSpec d,
                       float *dxcnn, int64_t ld_dxc);

// Row-wise log-softmax + target pick + (optional) dlogits = (softmax - onehot) * scale   (lrcn.jl:562-567 and dual).
// logits f32 [M][ld_l]; accumulates sum of log p(target) into *logp_sum (double).  dlog (T) may be NULL.
// logp_rows != NULL (LRCN_OPT_DETERMINISTIC): each row's term goes to logp_rows[m] and one workgroup adds them to *logp_sum in a
// fixed order, instead of M double atomics.
void k_softmax_xent(hipStream_t st, int dtype, const float *logits, int64_t ld_l, const int32_t *tgt, int M, int V,
                    float scale, double *logp_sum, void *dlog, int64_t ld_d, double *logp_rows = nullptr);
// prob[v] = exp(logp) for one row each (beam search, lrcn.jl:652).
void k_softmax_rows(hipStream_t st, const float *logits, int64_t ld_l, int M, int V, float *prob, int64_t ld_p);

// out[c][r + shift] = in[r][c] (0<=r<R, 0<=c<C), out[c][0..shift) = 0.  in_f32/out types: in is f32 if in_f32 else T;
// out is always T.   (builds the K-contiguous transposed operands of the weight-gradient GEMMs)
void k_transpose(hipStream_t st, int dtype, int in_f32, const void *in, int64_t ld_in, int R, int C, void *out,
                 int64_t ld_out, int shift);
// out_f32[c][r] = in[r][c], f32 -> f32 (boundary layout changes)
void k_transpose_f32(hipStream_t st, const float *in, int64_t ld_in, int R, int C, float *out, int64_t ld_out);
// One launch for all shadow weights: parameter memory image src[R][C] (f32, C contiguous) -> direct copies split at column cs
// (dA[r][c] | dB[r][c - cs]) and transposed copies (tA[c][r] | tB[c - cs][r]) in T; NULL destinations are skipped.  Padding
// columns of the destinations are left as they are (zero since allocation).
#define PREP_MAX 12
struct PrepDesc {
    const float *src;
    // k_adam_shadows only: the tensor's gradient and Adam moments (memory images like src, which is then also written)
    const float *g;
    float *m, *v;
    int R, C, cs;
    void *dA, *dB, *tA, *tB;
    int64_t ldA, ldB, ldtA, ldtB;
    int tile0;
    // optional third direct copy of the B side with the ROWS in (unit, gate)-interleaved order: source row g*giH + u -> row 4u + g
    // (the recurrent weights of gemm_8p.hip's LSTM_FWD epilogue); NULL = none
    void *dG;
    int64_t ldG;
    int giH;
    // permH > 0: the DIRECT copies dA / dB are written with their rows in (unit, gate)-interleaved order (source row g*permH + u -> row
    // 4u + g): the concatenated [x | h] decode weights of the cell-epilogue decode step (round 5)
    int permH;
};
struct PrepPlan {
    PrepDesc d[PREP_MAX];
    int n;
    int total;                      // tiles of all descriptors (set by the launchers)
    int grid_cap;                   // k_adam_shadows: > 0 = at most this many workgroups walk the tiles (an update beside the backward pass)
    float lr, b1, b2, eps, c1, c2;  // k_adam_shadows
};
void k_prepare_weights(hipStream_t st, int dtype, PrepPlan &plan);
// update! (lrcn.jl:394) and the NEXT step's shadow weights in one pass over the parameters: every descriptor's src / g / m / v are
// updated exactly as k_adam does (same arithmetic, element by element) and the new values are written to the descriptor's shadow
// destinations.  Descriptors without destinations (the biases: R = 1) are plain Adam.
void k_adam_shadows(hipStream_t st, int dtype, PrepPlan &plan, int step, float lr, float b1, float b2, float eps);
// out[r][c] = (T) in[r*ld_in + c]  (f32 -> T copy of a sub-matrix; pads [C, ld_out) with zeros)
void k_cast_rows(hipStream_t st, int dtype, const float *in, int64_t ld_in, int R, int C, void *out, int64_t ld_out);
// out[r][c] = (T) act(in[r][c] + bias[c])  (epilogue of a split-K GEMM whose partial sums were combined in f32)
void k_bias_act_cast(hipStream_t st, int dtype, const float *in, int64_t ld_in, const float *bias, int relu, int R, int C, void *out,
                     int64_t ld_out);
// out_f32[r][c] = in[r][c] (T -> f32)
void k_uncast_rows(hipStream_t st, int dtype, const void *in, int64_t ld_in, int R, int C, float *out, int64_t ld_out);

// db[n] = sum_m Z[m][n]   (Z is T [M][ld]); f32 output, overwritten.
// deterministic: one slab of rows per column block (no atomics between slabs)
void k_colsum(hipStream_t st, int dtype, const void *z, int64_t ld, int M, int N, float *out, bool deterministic = false);
// Several T -> T transposes in one launch: dst[c][shift + r] = src[r][c]; columns [0, shift) and [R + shift, ld_dst) of every
// destination row are written as zeros (K padding of the GEMM that consumes it).  R == 0 zero-fills the C destination rows.
#define TR_MAX 4
struct TrDesc {
    const void *src;
    void *dst;
    int64_t ld_src, ld_dst;
    int R, C, shift, tile0;
};
struct TrPlan {
    TrDesc d[TR_MAX];
    int n;
};
void k_transpose_multi(hipStream_t st, int dtype, TrPlan &plan);

struct AdamTensors {
    float *w[9];
    const float *g[9];
    float *m[9];
    float *v[9];
    int64_t n[9];
};
// update! with Adam (lrcn.jl:394; Knet defaults), all 9 tensors in one launch.
void k_adam(hipStream_t st, const AdamTensors &t, int step, float lr, float b1, float b2, float eps);
// xavier-uniform / constant fill (initweights, lrcn.jl:489-510)
void k_init_uniform(hipStream_t st, float *w, int64_t n, float scale, uint64_t seed, int tensor);
void k_fill(hipStream_t st, float *w, int64_t n, float v);

// ---- VGG side ----
// conv weight (3,3,Cin,Cout) column-major f32 -> [Cout][tap = b*3+a][Cin_pad] T (zero padded channels)
void k_repack_conv_w(hipStream_t st, int dtype, const float *w, int Cin, int Cout, int Cin_pad, void *out);
// conv1_1 weight -> [64][ld] T with k = tap*3 + c (27 real, rest zero)
void k_repack_conv11_w(hipStream_t st, int dtype, const float *w, int Cout, void *out, int64_t ld);
// out = (bf16)(img - mean[c]) over n = N*S*S*3 bytes, written INSIDE A FRAME of 2 zero pixels: out[n][S+4][S+4][3], pixel (x, y) of the crop at
// [x+2][y+2]; the frame is never written (the buffer is zero since allocation) -- the input of conv64 FUSE, whose raw-window DMA reads
// conv1_1's zero padding from it.  Needs (N*(S+4)*(S+4)*3 + 8) elements.
// avg != NULL: subtract the full averageImage (S,S,3) column-major, avg(col, row, c) from pixel (row, col, c) (lrcn.jl:770-771), instead
void k_img_u8_to_bf16(hipStream_t st, const uint8_t *img, int64_t n, float m0, float m1, float m2, const float *avg, int S, void *out);
// batched resize + centre crop + grey -> RGB of variable-size decoded uint8 images (lrcn.jl:755-765): meta = device array of
// {int64 byte offset into src, int h, w, channels, pad} per image; out = uint8 crops [N][S][S][3]
void k_resize_crop_u8(hipStream_t st, const uint8_t *src, const void *meta, int N, int S, uint8_t *out);
// feats (N x F column-major f32): every row divided by its sum (lrcn.jl:595-597)
void k_normalize_rows(hipStream_t st, float *feats, int N, int F);
// conv1_1 weight -> [64][32] bf16 in the K order of the fused conv1_1+conv1_2 kernel (conv64.hip, FUSE)
void k_repack_conv11_w_fused(hipStream_t st, const float *w, const float *b, void *out);  // b: conv1_1 bias (pieces at k' = 27..29), may be NULL
// fc6 weight (4096 x 25088 column-major, k_ref = x + 7y + 49c) -> [4096][25088] T with k = (y*7+x)*512 + c
void k_repack_fc6_w(hipStream_t st, int dtype, const float *w, void *out);
// conv1_1 im2col from uint8 crops img[n][row][col][3]: A[m][k = tap*3+c] (T, ld), m window-major over (y=col, x=row);
// value = pixel - mean[c], zero outside the image.                               (lrcn.jl:766-772 + :724)
void k_im2col11_u8(hipStream_t st, int dtype, const uint8_t *img, int N, int S, float m0, float m1, float m2, void *out,
                   int64_t ld);
// same from the preprocessed float tensor x (S,S,3,N) column-major
void k_im2col11_f32(hipStream_t st, int dtype, const float *x, int N, int S, void *out, int64_t ld);
// out(i,j,c,n) = img[n][i][j][c] - mean[c]   (lrcn.jl:766-772)
void k_preprocess_u8(hipStream_t st, const uint8_t *img, int N, int S, float m0, float m1, float m2, const float *avg, float *out);
// reference (W,H,C,N) column-major f32 <-> internal NHWC [n][y][x][C_ld] T   (x = dim 1, y = dim 2)
void k_ref_to_nhwc(hipStream_t st, int dtype, const float *x, int W, int H, int C, int N, void *out, int C_ld);
void k_nhwc_to_ref(hipStream_t st, int dtype, const void *in, int W, int H, int C, int N, int C_ld, float *out);

// ---- beam search (lrcn.jl:644-678) ----
// For each of R rows of prob [R][ld]: the K largest entries in descending order, ties to the lower index.
void k_topk_rows(hipStream_t st, const float *prob, int64_t ld, int R, int V, int K, int32_t *idx, float *val);
// One decode step of beam bookkeeping for N images (one workgroup each): see beam_update_kernel.  K <= 32.
void k_beam_update(hipStream_t st, const int32_t *topi, const float *topv, const int32_t *seq_in, int32_t *seq_out, float *p,
                   int32_t *parent, int32_t *last, int32_t *done, int32_t *ndone, int32_t *res_tok, int32_t *res_len, float *res_p, int N,
                   int K, int L, int current, int nword, int eos);
// out[r][0..C) = in[r / K][0..C): every image row repeated K times (rows ld apart in both)
void k_repeat_rows(hipStream_t st, int dtype, const void *in, int64_t ld, int N, int K, int C, void *out);
// out[r][0..C) = in[src_row[r]][0..C)  (beam search parent-state gather, lrcn.jl:673-676); in != out.
void k_gather_rows_f32(hipStream_t st, const float *in, int64_t ld, const int32_t *src_row, int R, int C, float *out);
// out[i] = a[i] * b[i]
void k_mul_f32(hipStream_t st, const float *a, const float *b, int64_t n, float *out);
// conv1_1 + preprocessing fused, bf16 (conv11.hip): src = uint8 crops img[n][row][col][3] or float (S,S,3,N);
// w [64][32] bf16 (k = tap*3+c), out NHWC bf16 [n][y][x][64] with bias + ReLU.
void k_conv11_fused(hipStream_t st, int src_is_u8, const void *src, int N, int S, float m0, float m1, float m2, const void *w,
                    const float *bias, void *out);

// softmax + top-K of every row in one pass (false: V too large for the register-resident form, use the two kernels)
bool k_softmax_topk_rows(hipStream_t st, const float *logits, int64_t ld, int R, int V, int K, int32_t *idx, float *val);
// out[i][r][:] = in[i][parent[r]][:] for the four recurrent states (row stride C[i]); hT[i] != NULL also receives a T copy (ld ldT[i])
// second half of the logits GEMM's softmax / top-K epilogue (gemm.h SmaxEpi): records [R][nrec][SMAX_REC] -> idx / val [R][K] as k_softmax_topk_rows;
// false = not applicable (K >= SMAX_KC, too many records)
bool k_softmax_topk_merge(hipStream_t st, const float *part, int nrec, int R, int K, int32_t *idx, float *val);
// bf16 batched decode, one launch per step: embedding of each hypothesis' last token + h1 / h2 of its parent into the [x | h] operands
void k_decode_prep(hipStream_t st, const void *wembT, int64_t ld_w, const int32_t *last, const int32_t *parent, int R, int E, const void *h1,
                   int64_t ld_h1, int H1, const void *h2, int64_t ld_h2, int H2, void *xh1, int64_t ld_xh1, int64_t off_h1, void *xh2, int64_t ld_xh2,
                   int64_t off_h2);
// decode step with input-projection tables: the parents' bf16 h into the gate GEMMs' operands; out[r] = r / K (the image of a hypothesis row)
void k_decode_prep_h(hipStream_t st, const int32_t *parent, int R, const void *h1, int64_t ld_h1, int H1, const void *h2, int64_t ld_h2, int H2,
                     void *a1, int64_t ld_a1, void *a2, int64_t ld_a2, int64_t off_h2);
void k_row_div(hipStream_t st, int32_t *out, int R, int K);
void k_gather_state(hipStream_t st, int dtype, const float *const in[4], float *const out[4], void *const hT[4], const int64_t ldT[4],
                    const int C[4], const int32_t *parent, int R);

// ---- fp8.hip: OCP e4m3 plumbing of the VGG convolution stack ----
void k_quant_conv_w_fp8(hipStream_t st, const float *w, int Cin, int Cout, void *out, float *sw);
void k_amax(hipStream_t st, int in_f32, const void *x, int64_t n, float *out);  // atomic max of |x| into *out (caller zeroes)
void k_cast_bf16_fp8(hipStream_t st, const void *x, int64_t n, float inv_scale, void *out);  // n % 8 == 0
void k_cast_fp8_bf16(hipStream_t st, const void *x, int64_t n, float scale, void *out);
void k_fp8_epilogue_params(hipStream_t st, const float *b, const float *sw, int Cout, float sa_in, float sa_out, float *escale, float *ebias);
void k_ref_to_nhwc_fp8(hipStream_t st, const float *x, int W, int H, int C, int N, float inv_scale, void *out);
void k_nhwc_fp8_to_ref(hipStream_t st, const void *in, int W, int H, int C, int N, float scale, float *out);
