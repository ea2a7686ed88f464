/*
 * lrcn_oracle.c -- CPU restatement of the reference's LRCN hot path (see lrcn_oracle.h header:
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED).  Plain C99 + OpenMP, fp32 storage, ORC_ACC
 * (double unless overridden) accumulation.  Every function cites the reference lines it restates.
 *
 * Deliberately written step-by-step (one lrcn() call per timestep, [input hidden] * W as one product),
 * exactly as lrcn.jl does -- NOT time-batched like the HIP path -- so the two are independent statements.
 */
#include "lrcn_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef ORC_ACC
#define ORC_ACC double
#endif
typedef ORC_ACC acc_t;

#define CM(A, ld, i, j) ((A)[(size_t)(i) + (size_t)(j) * (size_t)(ld)])

/* ------------------------------------------------------------------------------------------------
 * ORC_EMULATE_BF16 (orc_set_emulate_bf16): the SAME restatement with every value rounded to bfloat16 (round to nearest even)
 * at exactly the points where the HIP library's LRCN_BF16 / vgg bf16 arithmetic stores or feeds a bf16 value (lrcn_oracle.h lists
 * them); accumulation stays ORC_ACC, cell state / softmax / loss stay as they are.  Off (the default) every rb() is the identity
 * and the oracle is bit-for-bit the one tests/golden pins.  Emulation exists so that the bf16 kernels can be compared
 * ELEMENTWISE (what is left is summation order and the rare 1-ulp flip of a rounding) instead of by cosine.
 * ------------------------------------------------------------------------------------------------ */
static int g_emu = 0;
void orc_set_emulate_bf16(int on) { g_emu = on != 0; }
int orc_get_emulate_bf16(void) { return g_emu; }
float orc_bf16_round(float x) { /* RNE to 8 significant bits; NaN stays NaN, overflow rounds to inf as the hardware conversion does */
    union { float f; uint32_t u; } v;
    v.f = x;
    if ((v.u & 0x7fffffffu) > 0x7f800000u) return x;
    v.u += 0x7fffu + ((v.u >> 16) & 1u);
    v.u &= 0xffff0000u;
    return v.f;
}
static inline float rb(float x) { return g_emu ? orc_bf16_round(x) : x; }

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void orc_set_num_threads(int n) { /* a container's CPU share may be smaller than the machine's core count */
#ifdef _OPENMP
    if (n >= 1) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

static void *xmalloc(size_t n) {
    void *p = malloc(n ? n : 1);
    if (!p) abort();
    return p;
}
static float *fzeros(size_t n) {
    float *p = (float *)xmalloc(n * sizeof(float));
    memset(p, 0, n * sizeof(float));
    return p;
}

/* ------------------------------------------------------------------------------------------------
 * Column-major GEMM, C(MxN) = beta*C + op(A)*op(B).
 *   tA=0: A is M x K (lda);  tA=1: A is K x M and op(A)=A'.
 *   tB=0: B is K x N (ldb);  tB=1: B is N x K and op(B)=B'.
 * Stands in for cublasSgemm at the `*` call sites (lrcn.jl:529, 545, 550, 558) and their AutoGrad duals.
 * ------------------------------------------------------------------------------------------------ */
static void gemm_cm_plain(int tA, int tB, int M, int N, int K, const float *A, int lda, const float *B, int ldb, float beta, float *C,
                          int ldc);
static float *rounded_copy(const float *A, int rows, int cols, int ld) {
    float *r = (float *)malloc(sizeof(float) * (size_t)(rows ? rows : 1) * (size_t)(cols ? cols : 1));
    if (!r) abort();
#pragma omp parallel for schedule(static)
    for (int j = 0; j < cols; ++j)
        for (int i = 0; i < rows; ++i) r[(size_t)i + (size_t)j * rows] = orc_bf16_round(CM(A, ld, i, j));
    return r;
}
static void gemm_cm(int tA, int tB, int M, int N, int K, const float *A, int lda, const float *B, int ldb,
                    float beta, float *C, int ldc) {
    if (g_emu) { /* every contraction of the bf16 path takes bf16 operands (MFMA bf16 x bf16 -> f32): round both on load */
        const int ra = tA ? K : M, ca = tA ? M : K, rbw = tB ? N : K, cbw = tB ? K : N;
        float *Ar = rounded_copy(A, ra, ca, lda), *Br = rounded_copy(B, rbw, cbw, ldb);
        gemm_cm_plain(tA, tB, M, N, K, Ar, ra, Br, rbw, beta, C, ldc);
        free(Ar);
        free(Br);
        return;
    }
    gemm_cm_plain(tA, tB, M, N, K, A, lda, B, ldb, beta, C, ldc);
}
static void gemm_cm_plain(int tA, int tB, int M, int N, int K, const float *A, int lda, const float *B, int ldb,
                          float beta, float *C, int ldc) {
#ifdef ORC_FAST_GEMM
    /* Timed CPU-baseline build only: the two operand orders that the plain loops below walk with long strides or as
     * millions of 16-element dot products (the reverse pass of every timestep) are re-ordered so that the inner loop streams
     * contiguous memory.  Same sums, float accumulation; the checker build keeps the plain loops. */
    if (!tA && tB) { /* C(m,n) = sum_k A(m,k) B(n,k): rank-1 updates over k on a block of 64 columns that stays in L1 */
#pragma omp parallel for schedule(static)
        for (int n0 = 0; n0 < N; n0 += 64) {
            const int nn = N - n0 < 64 ? N - n0 : 64;
            for (int n = 0; n < nn; ++n) {
                float *c = &CM(C, ldc, 0, n0 + n);
                if (beta == 0.0f)
                    for (int m = 0; m < M; ++m) c[m] = 0.0f;
                else if (beta != 1.0f)
                    for (int m = 0; m < M; ++m) c[m] *= beta;
            }
            for (int k = 0; k < K; ++k) {
                const float *a = &CM(A, lda, 0, k), *b = &CM(B, ldb, n0, k);
                for (int n = 0; n < nn; ++n) {
                    const float bkn = b[n];
                    float *c = &CM(C, ldc, 0, n0 + n);
                    for (int m = 0; m < M; ++m) c[m] += a[m] * bkn;
                }
            }
        }
        return;
    }
    if (tA && !tB && K <= 256) { /* C(m,n) = sum_k A(k,m) B(k,n), short K: transpose A once, then K axpys of length M per column */
        float *At = (float *)xmalloc(sizeof(float) * (size_t)K * M);
#pragma omp parallel for schedule(static)
        for (int m = 0; m < M; ++m)
            for (int k = 0; k < K; ++k) At[(size_t)k * M + m] = CM(A, lda, k, m);
#pragma omp parallel for schedule(static)
        for (int n = 0; n < N; ++n) {
            float *c = &CM(C, ldc, 0, n);
            if (beta == 0.0f)
                for (int m = 0; m < M; ++m) c[m] = 0.0f;
            else if (beta != 1.0f)
                for (int m = 0; m < M; ++m) c[m] *= beta;
            for (int k = 0; k < K; ++k) {
                const float bkn = CM(B, ldb, k, n);
                const float *a = At + (size_t)k * M;
                for (int m = 0; m < M; ++m) c[m] += a[m] * bkn;
            }
        }
        free(At);
        return;
    }
#endif
    if (!tA) {
#pragma omp parallel
        {
            acc_t *acc = (acc_t *)xmalloc((size_t)M * sizeof(acc_t));
#pragma omp for schedule(static)
            for (int n = 0; n < N; ++n) {
                for (int m = 0; m < M; ++m) acc[m] = 0;
                for (int k = 0; k < K; ++k) {
                    const acc_t bkn = tB ? CM(B, ldb, n, k) : CM(B, ldb, k, n);
                    const float *a = &CM(A, lda, 0, k);
                    for (int m = 0; m < M; ++m) acc[m] += (acc_t)a[m] * bkn;
                }
                for (int m = 0; m < M; ++m) {
                    float *c = &CM(C, ldc, m, n);
                    *c = (beta == 0.0f) ? (float)acc[m] : (float)((acc_t)beta * (acc_t)*c + acc[m]);
                }
            }
            free(acc);
        }
    } else {
        /* op(A)(m,k) = A(k,m): column m of A is contiguous in k -> dot products. Only tB=0 is needed. */
#pragma omp parallel for schedule(static)
        for (int n = 0; n < N; ++n) {
            for (int m = 0; m < M; ++m) {
                const float *a = &CM(A, lda, 0, m);
                acc_t s = 0;
                if (!tB) {
                    const float *b = &CM(B, ldb, 0, n);
                    for (int k = 0; k < K; ++k) s += (acc_t)a[k] * (acc_t)b[k];
                } else {
                    for (int k = 0; k < K; ++k) s += (acc_t)a[k] * (acc_t)CM(B, ldb, n, k);
                }
                float *c = &CM(C, ldc, m, n);
                *c = (beta == 0.0f) ? (float)s : (float)((acc_t)beta * (acc_t)*c + s);
            }
        }
    }
}

static inline float sigm_f(float x) { return 1.0f / (1.0f + expf(-x)); } /* Knet sigm */

/* ------------------------------------------------------------------------------------------------ */
void orc_param_sizes(int E, int H1, int H2, int V, int64_t s[9]) {
    const int h = (H2 + 1) / 2; /* ceil(Int, hidden[end]/2)  lrcn.jl:504 */
    s[0] = (int64_t)(E + H1) * 4 * H1;
    s[1] = 4 * H1;
    s[2] = (int64_t)(H2 + H2) * 4 * H2; /* X = hidden[end] for k==2  lrcn.jl:496-499 */
    s[3] = 4 * H2;
    s[4] = (int64_t)H1 * h;
    s[5] = (int64_t)ORC_CNNOUT * h;
    s[6] = (int64_t)V * E;
    s[7] = (int64_t)H2 * V;
    s[8] = V;
}
int64_t orc_param_count(int E, int H1, int H2, int V) {
    int64_t s[9], t = 0;
    orc_param_sizes(E, H1, H2, V, s);
    for (int i = 0; i < 9; ++i) t += s[i];
    return t;
}

static uint64_t splitmix64(uint64_t *s) {
    uint64_t z = (*s += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
/* xavier(rows, cols) [Knet 0.8.x]: fanout = size(w,1), fanin = size(w,2), s = sqrt(2/(fanin+fanout)),
 * w = 2s*rand() - s on Float64, cast to Float32 by atype (lrcn.jl:490). */
static void xavier(float *w, int rows, int cols, uint64_t *rng) {
    const double s = sqrt(2.0 / ((double)rows + (double)cols));
    const size_t n = (size_t)rows * cols;
    for (size_t i = 0; i < n; ++i) {
        const double u = (double)(splitmix64(rng) >> 11) * (1.0 / 9007199254740992.0);
        w[i] = (float)(2.0 * s * u - s);
    }
}
void orc_init_weights(orc_model *m, uint64_t seed) { /* lrcn.jl:489-510 */
    uint64_t rng = seed;
    const int h = (m->H2 + 1) / 2;
    xavier(m->W1, m->E + m->H1, 4 * m->H1, &rng);
    memset(m->b1, 0, sizeof(float) * 4 * m->H1);
    for (int i = 0; i < m->H1; ++i) m->b1[i] = 1.0f; /* model[2k][1:H] = 1  lrcn.jl:501 */
    xavier(m->W2, m->H2 + m->H2, 4 * m->H2, &rng);
    memset(m->b2, 0, sizeof(float) * 4 * m->H2);
    for (int i = 0; i < m->H2; ++i) m->b2[i] = 1.0f;
    xavier(m->Wproj, m->H1, h, &rng);
    xavier(m->Wcnn, ORC_CNNOUT, h, &rng);
    xavier(m->Wembed, m->V, m->E, &rng);
    xavier(m->Wout, m->H2, m->V, &rng);
    memset(m->bout, 0, sizeof(float) * m->V);
}

/* ------------------------------------------------------------------------------------------------
 * lstm  (lrcn.jl:528-538)
 *   gates = hcat(input,hidden) * weight .+ bias
 *   forget,ingate,outgate = sigm(gates[:, 1:H | H+1:2H | 2H+1:3H]); change = tanh(gates[:,3H+1:end])
 *   cell = cell .* forget + ingate .* change ; hidden = outgate .* tanh(cell)
 * ------------------------------------------------------------------------------------------------ */
static void lstm_fwd(const float *W, const float *b, int X, int H, int B, const float *x, const float *h,
                     const float *c, float *h_out, float *c_out, float *gates /* B x 4H, activated */,
                     float *xh /* B x (X+H) scratch: the hcat */) {
    memcpy(xh, x, sizeof(float) * (size_t)B * X);
    memcpy(xh + (size_t)B * X, h, sizeof(float) * (size_t)B * H);
    gemm_cm(0, 0, B, 4 * H, X + H, xh, B, W, X + H, 0.0f, gates, B);
    for (int n = 0; n < 4 * H; ++n) {
        const float bn = b[n];
        float *g = &CM(gates, B, 0, n);
        if (n < 3 * H)
            for (int i = 0; i < B; ++i) g[i] = sigm_f(g[i] + bn);
        else
            for (int i = 0; i < B; ++i) g[i] = tanhf(g[i] + bn);
    }
    for (int j = 0; j < H; ++j)
        for (int i = 0; i < B; ++i) {
            const float f = CM(gates, B, i, j), in = CM(gates, B, i, H + j), o = CM(gates, B, i, 2 * H + j),
                        g = CM(gates, B, i, 3 * H + j);
            const float cn = CM(c, B, i, j) * f + in * g;
            CM(c_out, B, i, j) = cn;
            CM(h_out, B, i, j) = o * tanhf(cn);
        }
}

void orc_lstm(const float *W, const float *b, int X, int H, int B, const float *x, const float *h,
              const float *c, float *h_out, float *c_out, float *gates_out) {
    float *gates = gates_out ? gates_out : fzeros((size_t)B * 4 * H);
    float *xh = fzeros((size_t)B * (X + H));
    float *hn = fzeros((size_t)B * H), *cn = fzeros((size_t)B * H);
    lstm_fwd(W, b, X, H, B, x, h, c, hn, cn, gates, xh);
    memcpy(h_out, hn, sizeof(float) * (size_t)B * H);
    memcpy(c_out, cn, sizeof(float) * (size_t)B * H);
    free(hn);
    free(cn);
    free(xh);
    if (!gates_out) free(gates);
}

/* Everything one lrcn() call (lrcn.jl:540-551) produces that the reverse pass needs. */
typedef struct {
    float *x1;   /* B x E    dropout(x_lstm)            :542 */
    float *xh1;  /* B x (E+H1) hcat for lstm 1 (holds h1_prev) */
    float *g1;   /* B x 4H1  activated gates            */
    float *c1p;  /* B x H1   cell before                */
    float *c1;   /* B x H1   cell after                 */
    float *h1;   /* B x H1   hidden after               :543 */
    float *x2;   /* B x H2   dropout(hcat(h1*Wproj, x_cnn)) :545-547 */
    float *xh2;  /* B x (H2+H2) */
    float *g2, *c2p, *c2, *h2;
    float *logits; /* B x V  :550 */
} step_tape;

static void tape_alloc(step_tape *t, const orc_model *m, int B) {
    const int E = m->E, H1 = m->H1, H2 = m->H2, V = m->V;
    t->x1 = fzeros((size_t)B * E);
    t->xh1 = fzeros((size_t)B * (E + H1));
    t->g1 = fzeros((size_t)B * 4 * H1);
    t->c1p = fzeros((size_t)B * H1);
    t->c1 = fzeros((size_t)B * H1);
    t->h1 = fzeros((size_t)B * H1);
    t->x2 = fzeros((size_t)B * H2);
    t->xh2 = fzeros((size_t)B * 2 * H2);
    t->g2 = fzeros((size_t)B * 4 * H2);
    t->c2p = fzeros((size_t)B * H2);
    t->c2 = fzeros((size_t)B * H2);
    t->h2 = fzeros((size_t)B * H2);
    t->logits = fzeros((size_t)B * V);
}
static void tape_free(step_tape *t) {
    free(t->x1); free(t->xh1); free(t->g1); free(t->c1p); free(t->c1); free(t->h1);
    free(t->x2); free(t->xh2); free(t->g2); free(t->c2p); free(t->c2); free(t->h2); free(t->logits);
}

/* lrcn (lrcn.jl:540-551), recording the tape. h1,c1,h2,c2 are the incoming state (not modified). */
static void lrcn_fwd(const orc_model *m, int B, const float *h1, const float *c1, const float *h2,
                     const float *c2, const float *x_cnn, const float *x_lstm, const float *mask1,
                     const float *mask2, step_tape *t) {
    const int E = m->E, H1 = m->H1, H2 = m->H2, V = m->V, hh = (H2 + 1) / 2;
    /* x = dropout(x_lstm, pdrop)  :542  (Knet 0.8.x: x .* (rand .> p) ./ (1-p); the mask is supplied) */
    /* [bf16] the gather writes bf16(embedding * multiplier) (embed_gather_kernel) */
    for (size_t i = 0; i < (size_t)B * E; ++i) t->x1[i] = rb(mask1 ? x_lstm[i] * mask1[i] : x_lstm[i]);
    memcpy(t->c1p, c1, sizeof(float) * (size_t)B * H1);
    lstm_fwd(m->W1, m->b1, E, H1, B, t->x1, h1, c1, t->h1, t->c1, t->g1, t->xh1); /* :543 */
    /* x = s[1] * w[end-4]; x = hcat(x, x_cnn); x = dropout(x)  :544-547 */
    gemm_cm(0, 0, B, hh, H1, t->h1, B, m->Wproj, H1, 0.0f, t->x2, B);
    memcpy(t->x2 + (size_t)B * hh, x_cnn, sizeof(float) * (size_t)B * hh);
    /* [bf16] the projection GEMM stores bf16; concat_x2_kernel then writes bf16(value * multiplier) over all 2h columns, reading the
     * left half back as bf16 and the right half (x_cnn) as f32 */
    for (size_t i = 0; i < (size_t)B * hh; ++i) t->x2[i] = rb(t->x2[i]);
    for (size_t i = 0; i < (size_t)B * H2; ++i) t->x2[i] = rb(mask2 ? t->x2[i] * mask2[i] : t->x2[i]);
    memcpy(t->c2p, c2, sizeof(float) * (size_t)B * H2);
    lstm_fwd(m->W2, m->b2, H2, H2, B, t->x2, h2, c2, t->h2, t->c2, t->g2, t->xh2); /* :548 */
    /* return x * w[end-1] .+ w[end]  :550 */
    gemm_cm(0, 0, B, V, H2, t->h2, B, m->Wout, H2, 0.0f, t->logits, B);
    for (int v = 0; v < V; ++v) {
        const float bv = m->bout[v];
        float *l = &CM(t->logits, B, 0, v);
        for (int i = 0; i < B; ++i) l[i] += bv;
    }
}

void orc_lrcn_step(const orc_model *m, int B, float *h1, float *c1, float *h2, float *c2, const float *x_cnn,
                   const float *x_lstm, const float *mask1, const float *mask2, float *logits) {
    step_tape t;
    tape_alloc(&t, m, B);
    lrcn_fwd(m, B, h1, c1, h2, c2, x_cnn, x_lstm, mask1, mask2, &t);
    memcpy(h1, t.h1, sizeof(float) * (size_t)B * m->H1);
    memcpy(c1, t.c1, sizeof(float) * (size_t)B * m->H1);
    memcpy(h2, t.h2, sizeof(float) * (size_t)B * m->H2);
    memcpy(c2, t.c2, sizeof(float) * (size_t)B * m->H2);
    memcpy(logits, t.logits, sizeof(float) * (size_t)B * m->V);
    tape_free(&t);
}

/* param[end-2][idx,:]  (lrcn.jl:556, 569): row gather of the V x E embedding. */
static void embed_rows(const orc_model *m, const int32_t *idx, int B, float *out /* B x E */) {
    for (int e = 0; e < m->E; ++e)
        for (int i = 0; i < B; ++i) CM(out, B, i, e) = rb(CM(m->Wembed, m->V, idx[i], e)); /* [bf16] gathered from the bf16 shadow */
}

/* logp(ypred,2) (lrcn.jl:562): row-wise log-softmax; returns sum_i logp(i, target[i]) in double (the host
 * Float64 `total`, lrcn.jl:554, 567) and, if dlogits != NULL, writes (softmax - onehot) * scale. */
static double logp_pick(const float *logits, int B, int V, const int32_t *target, float *dlogits,
                        double scale) {
    double total = 0.0;
    for (int i = 0; i < B; ++i) {
        float mx = CM(logits, B, i, 0);
        for (int v = 1; v < V; ++v) mx = fmaxf(mx, CM(logits, B, i, v));
        double se = 0.0;
        for (int v = 0; v < V; ++v) se += exp((double)(CM(logits, B, i, v) - mx));
        const double lse = (double)mx + log(se);
        total += (double)CM(logits, B, i, target[i]) - lse;
        if (dlogits)
            for (int v = 0; v < V; ++v) {
                const double p = exp((double)CM(logits, B, i, v) - lse);
                CM(dlogits, B, i, v) = rb((float)((p - (v == target[i] ? 1.0 : 0.0)) * scale)); /* [bf16] dlogits stored bf16 */
            }
    }
    return total;
}

/* Reverse of lstm.  dh, dc: gradients wrt (h_out, c_out); outputs dxh = d[input hidden] (B x (X+H)),
 * dc_prev; accumulates dW, db. */
static void lstm_bwd(const float *W, int X, int H, int B, const float *xh, const float *gates, const float *c_prev,
                     const float *c_new, const float *dh, const float *dc_in, float *dxh, float *dc_prev,
                     float *dW, float *db, float *dz /* B x 4H scratch */, int dx_bf16 /* [bf16] the dX GEMM stores bf16 */) {
    for (int j = 0; j < H; ++j)
        for (int i = 0; i < B; ++i) {
            /* [bf16] the forward cell kernel keeps the activated gates in bf16 for this pass (cell state stays f32) */
            const float f = rb(CM(gates, B, i, j)), in = rb(CM(gates, B, i, H + j)), o = rb(CM(gates, B, i, 2 * H + j)),
                        g = rb(CM(gates, B, i, 3 * H + j));
            const float tc = tanhf(CM(c_new, B, i, j));
            const float dhv = CM(dh, B, i, j);
            const float dov = dhv * tc;
            const float dcv = CM(dc_in, B, i, j) + dhv * o * (1.0f - tc * tc);
            CM(dz, B, i, j) = rb(dcv * CM(c_prev, B, i, j) * f * (1.0f - f)); /* [bf16] dz stored bf16 */
            CM(dz, B, i, H + j) = rb(dcv * g * in * (1.0f - in));
            CM(dz, B, i, 2 * H + j) = rb(dov * o * (1.0f - o));
            CM(dz, B, i, 3 * H + j) = rb(dcv * in * (1.0f - g * g));
            CM(dc_prev, B, i, j) = dcv * f;
        }
    /* dW += [x h]' * dz ; db += colsum(dz) ; [dx dh_prev] = dz * W' */
    gemm_cm(1, 0, X + H, 4 * H, B, xh, B, dz, B, 1.0f, dW, X + H);
    for (int n = 0; n < 4 * H; ++n) {
        acc_t s = 0;
        for (int i = 0; i < B; ++i) s += CM(dz, B, i, n);
        db[n] += (float)s;
    }
    gemm_cm(0, 1, B, X + H, 4 * H, dz, B, W, X + H, 0.0f, dxh, B);
    if (dx_bf16)
        for (size_t i = 0; i < (size_t)B * X; ++i) dxh[i] = rb(dxh[i]);
}

static void zero_model(orc_model *g) {
    int64_t s[9];
    orc_param_sizes(g->E, g->H1, g->H2, g->V, s);
    float *p[9] = {g->W1, g->b1, g->W2, g->b2, g->Wproj, g->Wcnn, g->Wembed, g->Wout, g->bout};
    for (int i = 0; i < 9; ++i) memset(p[i], 0, sizeof(float) * (size_t)s[i]);
}

static double loss_impl(const orc_model *m, const float *feats, const int32_t *tokens, int T, int B, int norm_B,
                        const float *mask1, const float *mask2, orc_model *G, float *logits_out) {
    const int E = m->E, H1 = m->H1, H2 = m->H2, V = m->V, hh = (H2 + 1) / 2;
    const int S = T + 1; /* T words + the extra eos step  lrcn.jl:560-579 */
    step_tape *tape = (step_tape *)xmalloc(sizeof(step_tape) * S);
    for (int s = 0; s < S; ++s) tape_alloc(&tape[s], m, B);
    /* input = input * param[end-3]  :558 */
    float *x_cnn = fzeros((size_t)B * hh);
    gemm_cm(0, 0, B, hh, ORC_CNNOUT, feats, B, m->Wcnn, ORC_CNNOUT, 0.0f, x_cnn, B);
    float *zeroH1 = fzeros((size_t)B * H1), *zeroH2 = fzeros((size_t)B * H2); /* initstate :512-526 */
    float *x_lstm = fzeros((size_t)B * E);
    int32_t *inp = (int32_t *)xmalloc(sizeof(int32_t) * B), *tgt = (int32_t *)xmalloc(sizeof(int32_t) * B);
    double total = 0.0;
    long count = 0;
    for (int s = 0; s < S; ++s) {
        for (int i = 0; i < B; ++i) {
            inp[i] = (s == 0) ? ORC_BOS : tokens[(size_t)(s - 1) * B + i]; /* :556, :569 */
            tgt[i] = (s < T) ? tokens[(size_t)s * B + i] : ORC_EOS;        /* :565, :576 */
        }
        embed_rows(m, inp, B, x_lstm);
        const float *h1 = s ? tape[s - 1].h1 : zeroH1, *c1 = s ? tape[s - 1].c1 : zeroH1;
        const float *h2 = s ? tape[s - 1].h2 : zeroH2, *c2 = s ? tape[s - 1].c2 : zeroH2;
        lrcn_fwd(m, B, h1, c1, h2, c2, x_cnn, x_lstm, mask1 ? mask1 + (size_t)s * B * E : NULL,
                 mask2 ? mask2 + (size_t)s * B * H2 : NULL, &tape[s]);
        total += logp_pick(tape[s].logits, B, V, tgt, NULL, 0.0); /* :562-567 */
        count += norm_B;                                          /* :568 (global batchsize) */
        if (logits_out) memcpy(logits_out + (size_t)s * B * V, tape[s].logits, sizeof(float) * (size_t)B * V);
    }
    const double loss = -total / (double)count; /* :580 */

    if (G) {
        zero_model(G);
        const double scale = 1.0 / (double)count; /* d(-total/count)/dlogp = -1/count; folded into (p - onehot) */
        float *dlog = fzeros((size_t)B * V);
        float *dh2 = fzeros((size_t)B * H2), *dc2 = fzeros((size_t)B * H2), *dc2p = fzeros((size_t)B * H2);
        float *dh1 = fzeros((size_t)B * H1), *dc1 = fzeros((size_t)B * H1), *dc1p = fzeros((size_t)B * H1);
        float *dxh2 = fzeros((size_t)B * 2 * H2), *dxh1 = fzeros((size_t)B * (E + H1));
        float *dz2 = fzeros((size_t)B * 4 * H2), *dz1 = fzeros((size_t)B * 4 * H1);
        float *dxcnn = fzeros((size_t)B * hh), *dp = fzeros((size_t)B * hh), *dh1p = fzeros((size_t)B * H1);
        for (int s = S - 1; s >= 0; --s) {
            step_tape *t = &tape[s];
            for (int i = 0; i < B; ++i) {
                inp[i] = (s == 0) ? ORC_BOS : tokens[(size_t)(s - 1) * B + i];
                tgt[i] = (s < T) ? tokens[(size_t)s * B + i] : ORC_EOS;
            }
            logp_pick(t->logits, B, V, tgt, dlog, scale);
            /* logits = h2*Wout .+ bout */
            gemm_cm(1, 0, H2, V, B, t->h2, B, dlog, B, 1.0f, G->Wout, H2);
            for (int v = 0; v < V; ++v) {
                acc_t a = 0;
                for (int i = 0; i < B; ++i) a += CM(dlog, B, i, v);
                G->bout[v] += (float)a;
            }
            gemm_cm(0, 1, B, H2, V, dlog, B, m->Wout, H2, 1.0f, dh2, B); /* dh2 += dlog*Wout' (dh2 holds recurrent part) */
            lstm_bwd(m->W2, H2, H2, B, t->xh2, t->g2, t->c2p, t->c2, dh2, dc2, dxh2, dc2p, G->W2, G->b2, dz2, 1);
            /* dxh2 = [d x2 (H2 cols) | d h2_prev (H2 cols)] */
            memcpy(dh2, dxh2 + (size_t)B * H2, sizeof(float) * (size_t)B * H2);
            memcpy(dc2, dc2p, sizeof(float) * (size_t)B * H2);
            if (mask2) {
                const float *mk = mask2 + (size_t)s * B * H2;
                for (size_t i = 0; i < (size_t)B * H2; ++i) dxh2[i] *= mk[i];
            }
            /* left hh columns -> projection; right hh columns -> x_cnn (summed over steps).
             * [bf16] dx2_mask_reduce_kernel sums the f32 products into d x_cnn and writes the masked values back as bf16 */
            for (size_t i = 0; i < (size_t)B * hh; ++i) dp[i] = rb(dxh2[i]);
            for (size_t i = 0; i < (size_t)B * hh; ++i) dxcnn[i] += dxh2[(size_t)B * hh + i];
            gemm_cm(1, 0, H1, hh, B, t->h1, B, dp, B, 1.0f, G->Wproj, H1);
            gemm_cm(0, 1, B, H1, hh, dp, B, m->Wproj, H1, 0.0f, dh1p, B);
            for (size_t i = 0; i < (size_t)B * H1; ++i) dh1[i] += dh1p[i];
            lstm_bwd(m->W1, E, H1, B, t->xh1, t->g1, t->c1p, t->c1, dh1, dc1, dxh1, dc1p, G->W1, G->b1, dz1, 0);
            memcpy(dh1, dxh1 + (size_t)B * E, sizeof(float) * (size_t)B * H1);
            memcpy(dc1, dc1p, sizeof(float) * (size_t)B * H1);
            if (mask1) {
                const float *mk = mask1 + (size_t)s * B * E;
                for (size_t i = 0; i < (size_t)B * E; ++i) dxh1[i] *= mk[i];
            }
            /* embedding gather backward: scatter-add rows */
            for (int e = 0; e < E; ++e)
                for (int i = 0; i < B; ++i) CM(G->Wembed, V, inp[i], e) += CM(dxh1, B, i, e);
        }
        gemm_cm(1, 0, ORC_CNNOUT, hh, B, feats, B, dxcnn, B, 0.0f, G->Wcnn, ORC_CNNOUT);
        free(dlog); free(dh2); free(dc2); free(dc2p); free(dh1); free(dc1); free(dc1p);
        free(dxh2); free(dxh1); free(dz2); free(dz1); free(dxcnn); free(dp); free(dh1p);
    }
    for (int s = 0; s < S; ++s) tape_free(&tape[s]);
    free(tape); free(x_cnn); free(zeroH1); free(zeroH2); free(x_lstm); free(inp); free(tgt);
    return loss;
}

double orc_loss(const orc_model *m, const float *feats, const int32_t *tokens, int T, int B, int norm_B,
                const float *mask1, const float *mask2, orc_model *grads) {
    return loss_impl(m, feats, tokens, T, B, norm_B, mask1, mask2, grads, NULL);
}
void orc_forward_logits(const orc_model *m, const float *feats, const int32_t *tokens, int T, int B,
                        float *logits_out) {
    loss_impl(m, feats, tokens, T, B, B, NULL, NULL, NULL, logits_out);
}

void orc_adam(float *w, const float *g, float *mom, float *var, int64_t n, int t, float lr, float beta1,
              float beta2, float eps) { /* lrcn.jl:394, 399-405 ; Knet Adam defaults */
    const double c1 = 1.0 - pow((double)beta1, (double)t), c2 = 1.0 - pow((double)beta2, (double)t);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        const float gi = g[i];
        const float mi = beta1 * mom[i] + (1.0f - beta1) * gi;
        const float vi = beta2 * var[i] + (1.0f - beta2) * gi * gi;
        mom[i] = mi;
        var[i] = vi;
        w[i] -= (float)((double)lr * ((double)mi / c1) / (sqrt((double)vi / c2) + (double)eps));
    }
}

/* ================================================================================================
 * LRCN-1f -- BASELINE configs[1] "1-layer LSTM" (SURVEY 8d).  NOT in the reference, which hard-wires two layers
 * (lrcn.jl:540-551; its initweights would raise BoundsError for length(hidden) == 1, :504): this repo's definition --
 * drop LSTM-1 and Wproj of lrcn() and feed dropout(hcat(x_lstm, x_cnn)) to ONE lstm (:528-538) of width H = H1 = H2, then the
 * same output layer (:550).  Everything else (loss :553-581, generate/beam_search :585-678, Adam) is the reference's code
 * with this step in place of lrcn().  The model uses W1 ((E+h+H) x 4H), b1, Wcnn, Wembed, Wout, bout of orc_model.
 * ================================================================================================ */
void orc1_param_sizes(int E, int H, int V, int64_t s[9]) {
    const int h = (H + 1) / 2;
    s[0] = (int64_t)(E + h + H) * 4 * H;
    s[1] = 4 * H;
    s[2] = s[3] = s[4] = 0;
    s[5] = (int64_t)ORC_CNNOUT * h;
    s[6] = (int64_t)V * E;
    s[7] = (int64_t)H * V;
    s[8] = V;
}
void orc1_init_weights(orc_model *m, uint64_t seed) { /* initweights' rules (lrcn.jl:489-510) on the 1f shapes */
    uint64_t rng = seed;
    const int H = m->H1, h = (H + 1) / 2;
    xavier(m->W1, m->E + h + H, 4 * H, &rng);
    memset(m->b1, 0, sizeof(float) * 4 * H);
    for (int i = 0; i < H; ++i) m->b1[i] = 1.0f;
    xavier(m->Wcnn, ORC_CNNOUT, h, &rng);
    xavier(m->Wembed, m->V, m->E, &rng);
    xavier(m->Wout, H, m->V, &rng);
    memset(m->bout, 0, sizeof(float) * m->V);
}

typedef struct {
    float *x;   /* B x (E+h)  dropout(hcat(x_lstm, x_cnn)) */
    float *xh;  /* B x (E+h+H) */
    float *g, *cp, *c, *h;
    float *logits;
} step1_tape;
static void tape1_alloc(step1_tape *t, const orc_model *m, int B) {
    const int H = m->H1, X = m->E + (H + 1) / 2;
    t->x = fzeros((size_t)B * X);
    t->xh = fzeros((size_t)B * (X + H));
    t->g = fzeros((size_t)B * 4 * H);
    t->cp = fzeros((size_t)B * H);
    t->c = fzeros((size_t)B * H);
    t->h = fzeros((size_t)B * H);
    t->logits = fzeros((size_t)B * m->V);
}
static void tape1_free(step1_tape *t) { free(t->x); free(t->xh); free(t->g); free(t->cp); free(t->c); free(t->h); free(t->logits); }

static void lrcn1_fwd(const orc_model *m, int B, const float *h, const float *c, const float *x_cnn, const float *x_lstm,
                      const float *mask, step1_tape *t) {
    const int E = m->E, H = m->H1, V = m->V, hh = (H + 1) / 2, X = E + hh;
    memcpy(t->x, x_lstm, sizeof(float) * (size_t)B * E);                  /* hcat(x_lstm, x_cnn) */
    memcpy(t->x + (size_t)B * E, x_cnn, sizeof(float) * (size_t)B * hh);
    /* dropout.  [bf16] the embedding half was gathered as bf16, x_cnn is f32; concat_x2_kernel writes bf16(value * multiplier) */
    for (size_t i = 0; i < (size_t)B * X; ++i) t->x[i] = rb(mask ? t->x[i] * mask[i] : t->x[i]);
    memcpy(t->cp, c, sizeof(float) * (size_t)B * H);
    lstm_fwd(m->W1, m->b1, X, H, B, t->x, h, c, t->h, t->c, t->g, t->xh);
    gemm_cm(0, 0, B, V, H, t->h, B, m->Wout, H, 0.0f, t->logits, B);     /* x * w[end-1] .+ w[end]  :550 */
    for (int v = 0; v < V; ++v) {
        const float bv = m->bout[v];
        float *l = &CM(t->logits, B, 0, v);
        for (int i = 0; i < B; ++i) l[i] += bv;
    }
}

void orc1_step(const orc_model *m, int B, float *h, float *c, const float *x_cnn, const float *x_lstm, const float *mask,
               float *logits) {
    step1_tape t;
    tape1_alloc(&t, m, B);
    lrcn1_fwd(m, B, h, c, x_cnn, x_lstm, mask, &t);
    memcpy(h, t.h, sizeof(float) * (size_t)B * m->H1);
    memcpy(c, t.c, sizeof(float) * (size_t)B * m->H1);
    memcpy(logits, t.logits, sizeof(float) * (size_t)B * m->V);
    tape1_free(&t);
}

static double loss1_impl(const orc_model *m, const float *feats, const int32_t *tokens, int T, int B, int norm_B, const float *mask,
                         orc_model *G, float *logits_out) {
    const int E = m->E, H = m->H1, V = m->V, hh = (H + 1) / 2, X = E + hh;
    const int S = T + 1;
    step1_tape *tape = (step1_tape *)xmalloc(sizeof(step1_tape) * S);
    for (int s = 0; s < S; ++s) tape1_alloc(&tape[s], m, B);
    float *x_cnn = fzeros((size_t)B * hh);
    gemm_cm(0, 0, B, hh, ORC_CNNOUT, feats, B, m->Wcnn, ORC_CNNOUT, 0.0f, x_cnn, B); /* :558 */
    float *zeroH = fzeros((size_t)B * H), *x_lstm = fzeros((size_t)B * E);
    int32_t *inp = (int32_t *)xmalloc(sizeof(int32_t) * B), *tgt = (int32_t *)xmalloc(sizeof(int32_t) * B);
    double total = 0.0;
    long count = 0;
    for (int s = 0; s < S; ++s) {
        for (int i = 0; i < B; ++i) {
            inp[i] = (s == 0) ? ORC_BOS : tokens[(size_t)(s - 1) * B + i];
            tgt[i] = (s < T) ? tokens[(size_t)s * B + i] : ORC_EOS;
        }
        embed_rows(m, inp, B, x_lstm);
        lrcn1_fwd(m, B, s ? tape[s - 1].h : zeroH, s ? tape[s - 1].c : zeroH, x_cnn, x_lstm, mask ? mask + (size_t)s * B * X : NULL, &tape[s]);
        total += logp_pick(tape[s].logits, B, V, tgt, NULL, 0.0);
        count += norm_B;
        if (logits_out) memcpy(logits_out + (size_t)s * B * V, tape[s].logits, sizeof(float) * (size_t)B * V);
    }
    const double loss = -total / (double)count;
    if (G) {
        int64_t sz[9];
        orc1_param_sizes(E, H, V, sz);
        float *gp[9] = {G->W1, G->b1, NULL, NULL, NULL, G->Wcnn, G->Wembed, G->Wout, G->bout};
        for (int k = 0; k < 9; ++k)
            if (gp[k]) memset(gp[k], 0, sizeof(float) * (size_t)sz[k]);
        const double scale = 1.0 / (double)count;
        float *dlog = fzeros((size_t)B * V), *dh = fzeros((size_t)B * H), *dc = fzeros((size_t)B * H), *dcp = fzeros((size_t)B * H);
        float *dxh = fzeros((size_t)B * (X + H)), *dz = fzeros((size_t)B * 4 * H), *dxcnn = fzeros((size_t)B * hh);
        for (int s = S - 1; s >= 0; --s) {
            step1_tape *t = &tape[s];
            for (int i = 0; i < B; ++i) {
                inp[i] = (s == 0) ? ORC_BOS : tokens[(size_t)(s - 1) * B + i];
                tgt[i] = (s < T) ? tokens[(size_t)s * B + i] : ORC_EOS;
            }
            logp_pick(t->logits, B, V, tgt, dlog, scale);
            gemm_cm(1, 0, H, V, B, t->h, B, dlog, B, 1.0f, G->Wout, H);
            for (int v = 0; v < V; ++v) {
                acc_t a = 0;
                for (int i = 0; i < B; ++i) a += CM(dlog, B, i, v);
                G->bout[v] += (float)a;
            }
            gemm_cm(0, 1, B, H, V, dlog, B, m->Wout, H, 1.0f, dh, B); /* dh holds the recurrent part */
            lstm_bwd(m->W1, X, H, B, t->xh, t->g, t->cp, t->c, dh, dc, dxh, dcp, G->W1, G->b1, dz, 0);
            memcpy(dh, dxh + (size_t)B * X, sizeof(float) * (size_t)B * H);
            memcpy(dc, dcp, sizeof(float) * (size_t)B * H);
            if (mask) {
                const float *mk = mask + (size_t)s * B * X;
                for (size_t i = 0; i < (size_t)B * X; ++i) dxh[i] *= mk[i];
            }
            for (int e = 0; e < E; ++e)
                for (int i = 0; i < B; ++i) CM(G->Wembed, V, inp[i], e) += CM(dxh, B, i, e);
            for (size_t i = 0; i < (size_t)B * hh; ++i) dxcnn[i] += dxh[(size_t)B * E + i];
        }
        gemm_cm(1, 0, ORC_CNNOUT, hh, B, feats, B, dxcnn, B, 0.0f, G->Wcnn, ORC_CNNOUT);
        free(dlog); free(dh); free(dc); free(dcp); free(dxh); free(dz); free(dxcnn);
    }
    for (int s = 0; s < S; ++s) tape1_free(&tape[s]);
    free(tape); free(x_cnn); free(zeroH); free(x_lstm); free(inp); free(tgt);
    return loss;
}
double orc1_loss(const orc_model *m, const float *feats, const int32_t *tokens, int T, int B, int norm_B, const float *mask,
                 orc_model *grads) {
    return loss1_impl(m, feats, tokens, T, B, norm_B, mask, grads, NULL);
}
void orc1_forward_logits(const orc_model *m, const float *feats, const int32_t *tokens, int T, int B, float *logits_out) {
    loss1_impl(m, feats, tokens, T, B, B, NULL, NULL, logits_out);
}

/* ------------------------------------------------------------------------------------------------
 * beam_search  (lrcn.jl:644-678) driven as generate does (lrcn.jl:609-633)
 * ------------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t *seq;
    int len;
    float p;
} hyp;

/* argsort descending, stable (Julia sortperm(...; rev=true) is stable: earlier index wins ties). */
static void argsort_desc(const float *v, int n, int *perm) {
    /* merge sort on indices */
    int *tmp = (int *)xmalloc(sizeof(int) * n);
    for (int i = 0; i < n; ++i) perm[i] = i;
    for (int w = 1; w < n; w *= 2) {
        for (int lo = 0; lo < n; lo += 2 * w) {
            int mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            int a = lo, b = mid, o = lo;
            while (a < mid && b < hi) tmp[o++] = (v[perm[b]] > v[perm[a]]) ? perm[b++] : perm[a++];
            while (a < mid) tmp[o++] = perm[a++];
            while (b < hi) tmp[o++] = perm[b++];
        }
        memcpy(perm, tmp, sizeof(int) * n);
    }
    free(tmp);
}

static int beam_impl(const orc_model *m, const float *feat, int K, int nword, int32_t *out_tokens, float *out_prob, int one_layer) {
    const int E = m->E, H1 = m->H1, H2 = m->H2, V = m->V, hh = (H2 + 1) / 2;
    const int maxlen = nword + 3;
    float *x_cnn = fzeros(hh);
    gemm_cm(0, 0, 1, hh, ORC_CNNOUT, feat, 1, m->Wcnn, ORC_CNNOUT, 0.0f, x_cnn, 1); /* :611 */
    /* K hypotheses ([bos], 1.0) and K zero states  :625-630 */
    hyp *x = (hyp *)xmalloc(sizeof(hyp) * K);
    float **st = (float **)xmalloc(sizeof(float *) * K); /* each: h1|c1|h2|c2 */
    const size_t ssz = 2 * (size_t)H1 + 2 * (size_t)H2;
    for (int i = 0; i < K; ++i) {
        x[i].seq = (int32_t *)xmalloc(sizeof(int32_t) * maxlen);
        x[i].seq[0] = ORC_BOS;
        x[i].len = 1;
        x[i].p = 1.0f;
        st[i] = fzeros(ssz);
    }
    float *logits = fzeros(V), *prob = fzeros(V), *emb = fzeros(E);
    int *perm = (int *)xmalloc(sizeof(int) * V);
    hyp *cand = (hyp *)xmalloc(sizeof(hyp) * K * K);
    for (int i = 0; i < K * K; ++i) cand[i].seq = (int32_t *)xmalloc(sizeof(int32_t) * maxlen);
    float *cp = (float *)xmalloc(sizeof(float) * K * K);
    int *corder = (int *)xmalloc(sizeof(int) * K * K);
    float **nst = (float **)xmalloc(sizeof(float *) * K);
    for (int i = 0; i < K; ++i) nst[i] = fzeros(ssz);

    int current = 1;
    for (;;) {
        int ncand = 0;
        for (int i = 0; i < K; ++i) {
            const int32_t last = x[i].seq[x[i].len - 1]; /* :648 */
            for (int e = 0; e < E; ++e) emb[e] = CM(m->Wembed, V, last, e); /* :650 */
            float *s = st[i];
            if (one_layer)
                orc1_step(m, 1, s, s + H1, x_cnn, emb, NULL, logits);
            else
                orc_lrcn_step(m, 1, s, s + H1, s + 2 * H1, s + 2 * H1 + H2, x_cnn, emb, NULL, NULL, logits); /* :651 */
            /* ynorm = exp(logp(ypred,2))  :652 */
            float mx = logits[0];
            for (int v = 1; v < V; ++v) mx = fmaxf(mx, logits[v]);
            double se = 0.0;
            for (int v = 0; v < V; ++v) se += exp((double)(logits[v] - mx));
            const double lse = (double)mx + log(se);
            for (int v = 0; v < V; ++v) prob[v] = (float)exp((double)logits[v] - lse);
            argsort_desc(prob, V, perm); /* :655 */
            for (int j = 0; j < K; ++j) { /* :656-661 */
                hyp *c = &cand[ncand];
                memcpy(c->seq, x[i].seq, sizeof(int32_t) * x[i].len);
                c->seq[x[i].len] = perm[j];
                c->len = x[i].len + 1;
                c->p = prob[perm[j]] * x[i].p; /* Float32 product */
                cp[ncand] = c->p;
                ++ncand;
            }
            if (current == 1) break; /* :662-664 */
        }
        argsort_desc(cp, ncand, corder); /* :667 */
        /* xs = new_x[sorted[1:K]]  :668 -- at current==1 there are exactly K candidates */
        const int done = (cand[corder[0]].seq[cand[corder[0]].len - 1] == ORC_EOS) || (current > nword); /* :670 */
        if (!done)
            for (int i = 0; i < K; ++i) { /* :673-676: parent = ceil(sorted[i]/K) (1-based) */
                const int parent = corder[i] / K;
                memcpy(nst[i], st[parent], sizeof(float) * ssz);
            }
        for (int i = 0; i < K; ++i) {
            const hyp *c = &cand[corder[i]];
            memcpy(x[i].seq, c->seq, sizeof(int32_t) * c->len);
            x[i].len = c->len;
            x[i].p = c->p;
        }
        if (done) break;
        for (int i = 0; i < K; ++i) {
            float *t = st[i];
            st[i] = nst[i];
            nst[i] = t;
        }
        ++current;
    }
    const int len = x[0].len;
    memcpy(out_tokens, x[0].seq, sizeof(int32_t) * len);
    if (out_prob) *out_prob = x[0].p;
    for (int i = 0; i < K; ++i) { free(x[i].seq); free(st[i]); free(nst[i]); }
    for (int i = 0; i < K * K; ++i) free(cand[i].seq);
    free(x); free(st); free(nst); free(cand); free(cp); free(corder); free(perm);
    free(logits); free(prob); free(emb); free(x_cnn);
    return len;
}
int orc_beam_search(const orc_model *m, const float *feat, int K, int nword, int32_t *out_tokens, float *out_prob) {
    return beam_impl(m, feat, K, nword, out_tokens, out_prob, 0);
}
int orc1_beam_search(const orc_model *m, const float *feat, int K, int nword, int32_t *out_tokens, float *out_prob) {
    return beam_impl(m, feat, K, nword, out_tokens, out_prob, 1);
}

/* ------------------------------------------------------------------------------------------------
 * VGG-16 to fc7  (lrcn.jl:697-748)
 * ------------------------------------------------------------------------------------------------ */
const int orc_vgg_cout[13] = {64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512};
const int orc_vgg_pool_after[13] = {0, 1, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0, 1};

/* convx(x,w) = conv4(w[1], x; padding=1, mode=1) .+ w[2]   (lrcn.jl:724); mode=1 = cross-correlation:
 *   y(i,j,co,n) = b(co) + sum_{a,b,ci} x(i+a-1, j+b-1, ci, n) * w(a,b,ci,co),  a,b in 0..2, zero padding. */
#ifdef ORC_FAST_GEMM
/* The timed CPU baseline build (liblrcn_oracle_f32.so) runs convx as what a CPU library would: im2col of a block of
 * pixels + a register-blocked SGEMM (4 output channels x 32 pixels of accumulators in AVX2 registers, GCC vector extensions),
 * OpenMP over (image, pixel block, channel block).  Same arithmetic as the direct loop below (lrcn.jl:724), float accumulation,
 * another summation order; tests/test_oracle_golden.py checks the two against each other.  The CHECKER build never uses it. */
typedef float v8f __attribute__((vector_size(32), aligned(4)));
#define FG_PB 128 /* pixels per block  */
#define FG_CB 128 /* output channels per block */
#define FG_KB 288 /* contraction block = 32 input channels x 9 taps */
static void conv3x3_fast(const float *x, int W, int H, int Cin, int N, const float *w, const float *b, int Cout, int relu, float *y) {
    const int P = W * H, K = 9 * Cin;
    const int npb = (P + FG_PB - 1) / FG_PB, ncb = (Cout + FG_CB - 1) / FG_CB;
    const long ntask = (long)N * npb * ncb;
#pragma omp parallel
    {
        float *col = (float *)xmalloc(sizeof(float) * FG_KB * FG_PB);
        float *cblk = (float *)xmalloc(sizeof(float) * FG_CB * FG_PB);
#pragma omp for schedule(dynamic, 1)
        for (long t = 0; t < ntask; ++t) {
            const int cb = (int)(t % ncb), pb = (int)((t / ncb) % npb), n = (int)(t / ((long)ncb * npb));
            const int p0 = pb * FG_PB, pn = P - p0 < FG_PB ? P - p0 : FG_PB;
            const int c0 = cb * FG_CB, cn = Cout - c0 < FG_CB ? Cout - c0 : FG_CB;
            for (int c = 0; c < cn; ++c)
                for (int p = 0; p < FG_PB; ++p) cblk[c * FG_PB + p] = b[c0 + c];
            for (int k0 = 0; k0 < K; k0 += FG_KB) {
                const int kn = K - k0 < FG_KB ? K - k0 : FG_KB;
                /* im2col: col[k][p] = x(i + a - 1, j + bb - 1, ci, n), zero outside; k = a + 3 bb + 9 ci (the weight's own order) */
                for (int k = 0; k < kn; ++k) {
                    const int kk = k0 + k, ci = kk / 9, bb = (kk % 9) / 3, a = kk % 3;
                    const float *xp = x + ((size_t)n * Cin + ci) * P;
                    float *cr = col + (size_t)k * FG_PB;
                    for (int p = 0; p < FG_PB; ++p) {
                        float v = 0.0f;
                        if (p < pn) {
                            const int pp = p0 + p, j = pp / W + bb - 1, i = pp % W + a - 1;
                            if ((unsigned)j < (unsigned)H && (unsigned)i < (unsigned)W) v = xp[(size_t)j * W + i];
                        }
                        cr[p] = v;
                    }
                }
                /* cblk[c][p] += sum_k w(k, c0 + c) * col[k][p]: 4 channels x 32 pixels per register tile */
                for (int c = 0; c < cn; c += 4) {
                    const int cr4 = cn - c < 4 ? cn - c : 4;
                    const float *w0 = w + (size_t)(c0 + c) * K + k0;
                    const float *w1 = cr4 > 1 ? w0 + K : w0, *w2 = cr4 > 2 ? w0 + 2 * (size_t)K : w0, *w3 = cr4 > 3 ? w0 + 3 * (size_t)K : w0;
                    for (int p = 0; p < FG_PB; p += 32) {
                        v8f acc[4][4];
                        for (int r = 0; r < 4; ++r)
                            for (int q = 0; q < 4; ++q) acc[r][q] = *(const v8f *)(cblk + (size_t)(c + (r < cr4 ? r : 0)) * FG_PB + p + 8 * q);
                        for (int k = 0; k < kn; ++k) {
                            const float *cp = col + (size_t)k * FG_PB + p;
                            const v8f b0 = *(const v8f *)cp, b1 = *(const v8f *)(cp + 8), b2 = *(const v8f *)(cp + 16), b3 = *(const v8f *)(cp + 24);
                            const float a0 = w0[k], a1 = w1[k], a2 = w2[k], a3 = w3[k];
                            acc[0][0] += a0 * b0; acc[0][1] += a0 * b1; acc[0][2] += a0 * b2; acc[0][3] += a0 * b3;
                            acc[1][0] += a1 * b0; acc[1][1] += a1 * b1; acc[1][2] += a1 * b2; acc[1][3] += a1 * b3;
                            acc[2][0] += a2 * b0; acc[2][1] += a2 * b1; acc[2][2] += a2 * b2; acc[2][3] += a2 * b3;
                            acc[3][0] += a3 * b0; acc[3][1] += a3 * b1; acc[3][2] += a3 * b2; acc[3][3] += a3 * b3;
                        }
                        for (int r = 0; r < cr4; ++r)
                            for (int q = 0; q < 4; ++q) *(v8f *)(cblk + (size_t)(c + r) * FG_PB + p + 8 * q) = acc[r][q];
                    }
                }
            }
            for (int c = 0; c < cn; ++c) {
                float *yp = y + ((size_t)n * Cout + c0 + c) * P + p0;
                const float *sp = cblk + (size_t)c * FG_PB;
                for (int p = 0; p < pn; ++p) yp[p] = (relu && sp[p] < 0.0f) ? 0.0f : sp[p];
            }
        }
        free(col);
        free(cblk);
    }
}
#endif

void orc_conv3x3(const float *x, int W, int H, int Cin, int N, const float *w, const float *b, int Cout,
                 int relu, float *y) {
#ifdef ORC_FAST_GEMM
    conv3x3_fast(x, W, H, Cin, N, w, b, Cout, relu, y);
    return;
#endif
    const size_t plane = (size_t)W * H;
    float *xr_ = NULL, *wr_ = NULL;
    if (g_emu) { /* [bf16] NHWC bf16 activations and bf16 filters into the MFMA, f32 bias as the accumulator's start, bf16 result */
        xr_ = rounded_copy(x, (int)plane, Cin * N, (int)plane);
        wr_ = rounded_copy(w, 9, Cin * Cout, 9);
        x = xr_;
        w = wr_;
    }
#pragma omp parallel
    {
        acc_t *acc = (acc_t *)xmalloc(plane * sizeof(acc_t));
#pragma omp for collapse(2) schedule(static)
        for (int n = 0; n < N; ++n)
            for (int co = 0; co < Cout; ++co) {
                for (size_t p = 0; p < plane; ++p) acc[p] = (acc_t)b[co];
                for (int ci = 0; ci < Cin; ++ci) {
                    const float *xp = x + ((size_t)n * Cin + ci) * plane;
                    const float *wp = w + ((size_t)co * Cin + ci) * 9; /* (a,b) col-major: a + 3b */
                    for (int bb = 0; bb < 3; ++bb)
                        for (int a = 0; a < 3; ++a) {
                            const acc_t wv = (acc_t)wp[a + 3 * bb];
                            const int j0 = bb == 0 ? 1 : 0, j1 = bb == 2 ? H - 1 : H;
                            const int i0 = a == 0 ? 1 : 0, i1 = a == 2 ? W - 1 : W;
                            for (int j = j0; j < j1; ++j) {
                                const float *xr = xp + (size_t)(j + bb - 1) * W + (a - 1);
                                acc_t *ar = acc + (size_t)j * W;
                                for (int i = i0; i < i1; ++i) ar[i] += (acc_t)xr[i] * wv;
                            }
                        }
                }
                float *yp = y + ((size_t)n * Cout + co) * plane;
                for (size_t p = 0; p < plane; ++p) {
                    const float v = (float)acc[p];
                    yp[p] = rb((relu && v < 0.0f) ? 0.0f : v); /* relux  :725 */
                }
            }
        free(acc);
    }
    free(xr_);
    free(wr_);
}

void orc_pool2(const float *x, int W, int H, int C, int N, float *y) { /* poolx :726, Knet pool default 2x2/2 max */
    const int Wo = W / 2, Ho = H / 2;
#pragma omp parallel for schedule(static)
    for (int nc = 0; nc < N * C; ++nc) {
        const float *xp = x + (size_t)nc * W * H;
        float *yp = y + (size_t)nc * Wo * Ho;
        for (int j = 0; j < Ho; ++j)
            for (int i = 0; i < Wo; ++i) {
                const float a = xp[(size_t)(2 * j) * W + 2 * i], b = xp[(size_t)(2 * j) * W + 2 * i + 1];
                const float c = xp[(size_t)(2 * j + 1) * W + 2 * i], d = xp[(size_t)(2 * j + 1) * W + 2 * i + 1];
                yp[(size_t)j * Wo + i] = fmaxf(fmaxf(a, b), fmaxf(c, d));
            }
    }
}

void orc_fc(const float *w, const float *b, int O, int K, int N, const float *x, int relu, float *y) { /* fcx :728 */
    gemm_cm(0, 0, O, N, K, w, O, x, K, 0.0f, y, O);
    for (int n = 0; n < N; ++n)
        for (int o = 0; o < O; ++o) {
            float v = CM(y, O, o, n) + b[o];
            CM(y, O, o, n) = (relu && v < 0.0f) ? 0.0f : v;
        }
}

void orc_vgg_forward(const orc_vgg *v, const float *x, int S, int N, float *feats) {
    int W = S, H = S, C = 3;
    float *cur = (float *)xmalloc(sizeof(float) * (size_t)W * H * C * N);
    memcpy(cur, x, sizeof(float) * (size_t)W * H * C * N);
    for (int l = 0; l < 13; ++l) { /* 13 x (conv, relu) + 5 pool   */
        const int Co = orc_vgg_cout[l];
        float *nxt = (float *)xmalloc(sizeof(float) * (size_t)W * H * Co * N);
        orc_conv3x3(cur, W, H, C, N, v->conv_w[l], v->conv_b[l], Co, 1, nxt);
        free(cur);
        cur = nxt;
        C = Co;
        if (orc_vgg_pool_after[l]) {
            float *p = (float *)xmalloc(sizeof(float) * (size_t)(W / 2) * (H / 2) * C * N);
            orc_pool2(cur, W, H, C, N, p);
            free(cur);
            cur = p;
            W /= 2;
            H /= 2;
        }
    }
    /* fc6 + relu6, fc7 (no relu7: the break at lrcn.jl:717 fires on fc7 itself); mat(x) flattens (w,h,c) col-major */
    const int K6 = W * H * C;
    float *f6 = (float *)xmalloc(sizeof(float) * (size_t)4096 * N);
    orc_fc(v->fc6_w, v->fc6_b, 4096, K6, N, cur, 1, f6);
    for (size_t i = 0; i < (size_t)4096 * N; ++i) f6[i] = rb(f6[i]); /* [bf16] relu6's output is stored bf16; fc7's stays f32 */
    float *f7 = (float *)xmalloc(sizeof(float) * (size_t)4096 * N);
    orc_fc(v->fc7_w, v->fc7_b, 4096, 4096, N, f6, 0, f7);
    /* return transpose(xs)  :746  -> N x 4096 col-major */
    for (int n = 0; n < N; ++n)
        for (int o = 0; o < 4096; ++o) CM(feats, N, n, o) = CM(f7, 4096, o, n);
    free(cur); free(f6); free(f7);
}

/* read_image_data tail (lrcn.jl:766-772) on a decoded S x S RGB uint8 crop img[n][row][col][c]:
 *   c1(w,h,c) = pixel(row=h, col=w, c)      permutedims(channelview(b1),(3,2,1))      :766
 *   f1 = 255*e1 .- averageImage                                                       :770
 *   g1(i,j,c) = f1(j,i,c)                   permutedims(f1,[2,1,3,4])                 :771
 * so out(i,j,c,n) = pixel(row=i, col=j, c) - mean[c]: dim 1 (fastest) runs down image ROWS. */
void orc_preprocess_u8(const uint8_t *img, int S, int N, const float mean[3], float *out) {
    for (int n = 0; n < N; ++n)
        for (int c = 0; c < 3; ++c)
            for (int j = 0; j < S; ++j)
                for (int i = 0; i < S; ++i)
                    out[(size_t)i + (size_t)S * (j + (size_t)S * (c + 3 * (size_t)n))] =
                        (float)img[(((size_t)n * S + i) * S + j) * 3 + c] - mean[c];
}
