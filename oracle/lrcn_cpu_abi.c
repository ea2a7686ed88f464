/*
 * lrcn_cpu_abi.c -- liblrcn_cpu.so: the C ABI of include/lrcn.h implemented on the HOST by the CPU oracle (SURVEY 8b "Library"
 * row: "a second build liblrcn_cpu.so exports the identical symbols from the oracle").
 *
 * TEST INFRASTRUCTURE ONLY, like everything under oracle/: the product (liblrcn_hip.so and the package around it) never loads it.
 * It exists so that (1) the ABI's argument conventions can be exercised without a GPU (tests/test_cpu_abi.py runs the golden
 * vectors THROUGH the ABI on the host), and (2) bench.py's cpu_baseline leg times the same step through the same entry points.
 * Every array pointer is a HOST pointer here; layouts, shapes, token conventions and error codes are those of include/lrcn.h.
 * What has no CPU meaning (streams, gradient-group events, grid caps, profiling, the fp8 stack, multi-rank communicators)
 * is accepted as a no-op or refused with LRCN_ESTATE, as noted per function.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../include/lrcn.h"
#include "lrcn_oracle.h"

struct lrcn_ctx {
    lrcn_config cfg;
    int nl, h;
    char err[512];
    double last_loss;
    /* VGG weights (copied by lrcn_vgg_load) */
    int vgg_loaded;
    float *cw[13], *cb[13], *fc6w, *fc6b, *fc7w, *fc7b;
    float *avg; /* (224,224,3) or NULL */
};
static char g_create_err[256] = "";

#define FAIL(c, code, ...)                               \
    do {                                                 \
        snprintf((c)->err, sizeof((c)->err), __VA_ARGS__); \
        return (code);                                   \
    } while (0)

static void *xcopy(const void *src, size_t bytes) {
    void *p = malloc(bytes ? bytes : 1);
    if (p && src) memcpy(p, src, bytes);
    return p;
}
static orc_model model_of(const lrcn_ctx *c, float *const p[9]) {
    orc_model m;
    m.E = c->cfg.E; m.H1 = c->cfg.H1; m.H2 = c->cfg.H2; m.V = c->cfg.V;
    m.W1 = p[0]; m.b1 = p[1]; m.W2 = p[2]; m.b2 = p[3]; m.Wproj = p[4]; m.Wcnn = p[5]; m.Wembed = p[6]; m.Wout = p[7]; m.bout = p[8];
    return m;
}
static void sizes_of(const lrcn_ctx *c, int64_t s[9]) { lrcn_param_sizes_n(c->nl, c->cfg.E, c->cfg.H1, c->cfg.H2, c->cfg.V, s); }

/* ---- lifetime / plumbing ---- */
const char *lrcn_version(void) { return "lrcn-cpu 0.4 (oracle)"; }
int lrcn_abi_version(void) { return LRCN_ABI_VERSION; }
/* options: the host twin always makes its weights afresh and sums in a fixed order -- every valid option is accepted as a no-op */
int lrcn_set_option(lrcn_ctx *c, int option, int64_t value) {
    if (!c) return LRCN_EINVAL;
    if (option == LRCN_OPT_FUSED_UPDATE || option == LRCN_OPT_DETERMINISTIC) return (value == 0 || value == 1) ? LRCN_OK : LRCN_EINVAL;
    if (option == LRCN_OPT_CONV_CHUNK_BYTES) return value >= 0 ? LRCN_OK : LRCN_EINVAL;
    return LRCN_EINVAL;
}
int lrcn_params_touched(lrcn_ctx *c) { return c ? LRCN_OK : LRCN_EINVAL; }
const char *lrcn_last_error(const lrcn_ctx *c) { return c ? c->err : g_create_err; }
int lrcn_param_sizes_n(int nl, int E, int H1, int H2, int V, int64_t s[9]) {
    if (E < 1 || H1 < 1 || H2 < 2 || (H2 & 1) || V < 3 || !s || (nl != 0 && nl != 1 && nl != 2)) return LRCN_EINVAL;
    if (nl == 1) {
        if (H1 != H2) return LRCN_EINVAL;
        orc1_param_sizes(E, H1, V, s);
    } else {
        orc_param_sizes(E, H1, H2, V, s);
    }
    return LRCN_OK;
}
int lrcn_param_sizes(int E, int H1, int H2, int V, int64_t s[9]) { return lrcn_param_sizes_n(2, E, H1, H2, V, s); }
int lrcn_create(const lrcn_config *cfg, lrcn_ctx **out) {
    int64_t s[9];
    if (!cfg || !out) return LRCN_EINVAL;
    *out = NULL;
    if (lrcn_param_sizes_n(cfg->n_layers, cfg->E, cfg->H1, cfg->H2, cfg->V, s) != LRCN_OK || cfg->max_B < 1 || cfg->max_T < 0 ||
        cfg->max_T > LRCN_MAX_T || cfg->max_images < 0) {
        snprintf(g_create_err, sizeof(g_create_err), "invalid lrcn_config");
        return LRCN_EINVAL;
    }
    lrcn_ctx *c = (lrcn_ctx *)calloc(1, sizeof(lrcn_ctx));
    if (!c) return LRCN_ENOMEM;
    c->cfg = *cfg;
    c->nl = cfg->n_layers == 1 ? 1 : 2;
    c->h = cfg->H2 / 2;
    *out = c;
    return LRCN_OK;
}
void lrcn_destroy(lrcn_ctx *c) {
    if (!c) return;
    for (int l = 0; l < 13; ++l) { free(c->cw[l]); free(c->cb[l]); }
    free(c->fc6w); free(c->fc6b); free(c->fc7w); free(c->fc7b); free(c->avg);
    free(c);
}
int lrcn_set_stream(lrcn_ctx *c, void *s) { (void)s; return c ? LRCN_OK : LRCN_EINVAL; }      /* no streams on the host */
int lrcn_set_wg_stream(lrcn_ctx *c, void *s) { (void)s; return c ? LRCN_OK : LRCN_EINVAL; }
int lrcn_vgg_set_wg_cap(lrcn_ctx *c, int cap) { (void)cap; return c ? LRCN_OK : LRCN_EINVAL; }
int lrcn_sync(lrcn_ctx *c) { return c ? LRCN_OK : LRCN_EINVAL; }
int lrcn_malloc(void **p, size_t bytes) { *p = malloc(bytes ? bytes : 16); return *p ? LRCN_OK : LRCN_ENOMEM; }
int lrcn_free(void *p) { free(p); return LRCN_OK; }
int lrcn_memcpy_h2d(void *d, const void *s, size_t n) { memcpy(d, s, n); return LRCN_OK; }
int lrcn_memcpy_d2h(void *d, const void *s, size_t n) { memcpy(d, s, n); return LRCN_OK; }
const char *lrcn_debug_route(lrcn_ctx *c, int which) { (void)c; (void)which; return "cpu-oracle"; }

/* ---- model ---- */
int lrcn_init_weights(lrcn_ctx *c, float *const p[9], uint64_t seed) {
    if (!c || !p) return LRCN_EINVAL;
    orc_model m = model_of(c, p);
    if (c->nl == 1) orc1_init_weights(&m, seed); else orc_init_weights(&m, seed);
    return LRCN_OK;
}
int lrcn_lstm(lrcn_ctx *c, const float *W, const float *b, int X, int H, int B, const float *x, const float *h, const float *cc,
              float *h_out, float *c_out) {
    if (!c || !W || !b || !x || !h || !cc || !h_out || !c_out) return LRCN_EINVAL;
    if (B < 1 || B > c->cfg.max_B) FAIL(c, LRCN_EINVAL, "B=%d outside [1,%d]", B, c->cfg.max_B);
    orc_lstm(W, b, X, H, B, x, h, cc, h_out, c_out, NULL);
    return LRCN_OK;
}
int lrcn_step(lrcn_ctx *c, const float *const p[9], float *const st[4], int B, const float *x_cnn, const float *x_lstm,
              const float *mask1, const float *mask2, float *logits) {
    if (!c || !p || !st || !x_cnn || !x_lstm || !logits) return LRCN_EINVAL;
    if (B < 1 || B > c->cfg.max_B) FAIL(c, LRCN_EINVAL, "B=%d outside [1,%d]", B, c->cfg.max_B);
    orc_model m = model_of(c, (float *const *)p);
    if (c->nl == 1) orc1_step(&m, B, st[0], st[1], x_cnn, x_lstm, mask1, logits);
    else orc_lrcn_step(&m, B, st[0], st[1], st[2], st[3], x_cnn, x_lstm, mask1, mask2, logits);
    return LRCN_OK;
}

/* dropout: explicit masks as given; pdrop > 0 without masks -> multipliers from a counter hash of (seed, which, index).  The HIP
 * library's generator is a different hash: only the distribution (Bernoulli keep 1-p, scaled 1/(1-p)) is common, as with Knet's. */
static uint64_t mix64(uint64_t z) {
    z += 0x9E3779B97F4A7C15ull; z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
static float *gen_mask(const lrcn_dropout *d, int which, size_t n) {
    float *m = (float *)malloc(sizeof(float) * (n ? n : 1));
    const float keep = 1.0f / (1.0f - d->pdrop);
    for (size_t i = 0; i < n; ++i) {
        const double u = (double)(mix64(d->seed * 0x100000001B3ull + (uint64_t)which * 0x9E3779B9ull + i) >> 11) * (1.0 / 9007199254740992.0);
        m[i] = u > d->pdrop ? keep : 0.0f;
    }
    return m;
}
static int loss_common(lrcn_ctx *c, const float *const p[9], const float *feats, const int32_t *tokens, int T, int B, int norm_B,
                       const lrcn_dropout *drop, float *const grads[9], double *out) {
    if (T < 0 || T > c->cfg.max_T) FAIL(c, LRCN_EINVAL, "T=%d outside [0,%d]", T, c->cfg.max_T);
    if (B < 1 || B > c->cfg.max_B) FAIL(c, LRCN_EINVAL, "B=%d outside [1,%d]", B, c->cfg.max_B);
    if (norm_B < 1) FAIL(c, LRCN_EINVAL, "norm_B=%d must be >= 1", norm_B);
    if (drop && (drop->pdrop < 0.0f || drop->pdrop >= 1.0f)) FAIL(c, LRCN_EINVAL, "pdrop=%g outside [0,1)", drop->pdrop);
    for (long i = 0; i < (long)T * B; ++i)
        if (tokens[i] < 0 || tokens[i] >= c->cfg.V) FAIL(c, LRCN_EINVAL, "a token id was outside [0, V=%d)", c->cfg.V);
    const int S = T + 1, E = c->cfg.E, H2 = c->cfg.H2, X1 = c->nl == 1 ? E + c->h : E;
    const float *m1 = drop ? drop->mask1 : NULL, *m2 = drop ? drop->mask2 : NULL;
    float *own1 = NULL, *own2 = NULL;
    if (drop && drop->pdrop > 0.0f && !m1) {
        m1 = own1 = gen_mask(drop, 1, (size_t)S * B * X1);
        if (c->nl == 2) m2 = own2 = gen_mask(drop, 2, (size_t)S * B * H2);
    }
    orc_model m = model_of(c, (float *const *)p), g;
    if (grads) g = model_of(c, grads);
    const double v = c->nl == 1 ? orc1_loss(&m, feats, tokens, T, B, norm_B, m1, grads ? &g : NULL)
                                : orc_loss(&m, feats, tokens, T, B, norm_B, m1, m2, grads ? &g : NULL);
    free(own1); free(own2);
    c->last_loss = v;
    if (out) *out = v;
    return LRCN_OK;
}
int lrcn_loss(lrcn_ctx *c, const float *const p[9], const float *feats, const int32_t *tokens, int T, int B, int norm_B,
              const lrcn_dropout *drop, double *loss_host) {
    if (!c || !p || !feats || (!tokens && T > 0)) return LRCN_EINVAL;
    return loss_common(c, p, feats, tokens, T, B, norm_B, drop, NULL, loss_host);
}
int lrcn_loss_grad(lrcn_ctx *c, const float *const p[9], const float *feats, const int32_t *tokens, int T, int B, int norm_B,
                   const lrcn_dropout *drop, float *const grads[9], double *loss_host) {
    if (!c || !p || !feats || (!tokens && T > 0) || !grads) return LRCN_EINVAL;
    return loss_common(c, p, feats, tokens, T, B, norm_B, drop, grads, loss_host);
}
int lrcn_refresh_shadows_group(lrcn_ctx *c, const float *const p[9], int group, void *stream) {
    (void)p; (void)stream;
    return (c && group >= 0 && group < LRCN_GRAD_GROUPS) ? LRCN_OK : LRCN_EINVAL; /* the host twin reads the f32 parameters directly: no shadows */
}
int lrcn_avg_loss_batch(lrcn_ctx *c, const float *const p[9], const float *feats, const int32_t *tokens, int T, int B, double *loss_host) {
    if (!c || !p || !feats || (!tokens && T > 0)) return LRCN_EINVAL;
    return loss_common(c, p, feats, tokens, T, B, B, NULL, NULL, loss_host); /* average_loss's body: pdrop 0, the batch's own size (lrcn.jl:412, 452-475) */
}
int lrcn_grad_group_wait(lrcn_ctx *c, int group, void *stream) {
    (void)stream;
    return (c && group >= 0 && group < LRCN_GRAD_GROUPS) ? LRCN_OK : LRCN_EINVAL; /* synchronous on the host: always ready */
}
int lrcn_last_loss(lrcn_ctx *c, double *out) {
    if (!c || !out) return LRCN_EINVAL;
    *out = c->last_loss;
    return LRCN_OK;
}
int lrcn_forward_logits(lrcn_ctx *c, const float *const p[9], const float *feats, const int32_t *tokens, int T, int B, float *out) {
    if (!c || !p || !feats || (!tokens && T > 0) || !out) return LRCN_EINVAL;
    orc_model m = model_of(c, (float *const *)p);
    if (c->nl == 1) orc1_forward_logits(&m, feats, tokens, T, B, out); else orc_forward_logits(&m, feats, tokens, T, B, out);
    return LRCN_OK;
}
static const int kGroup[LRCN_GRAD_GROUPS][2] = {{7, 8}, {2, 3}, {4, 5}, {0, 1}, {6, 6}};
int lrcn_adam_update_group(lrcn_ctx *c, float *const p[9], const float *const g[9], float *const m[9], float *const v[9], int group,
                           int step, float lr, float b1, float b2, float eps, void *stream) {
    (void)stream;
    if (!c || !p || !g || !m || !v || step < 1 || group < 0 || group >= LRCN_GRAD_GROUPS) return LRCN_EINVAL;
    int64_t sz[9];
    sizes_of(c, sz);
    for (int k = 0; k < 9; ++k)
        if ((k == kGroup[group][0] || k == kGroup[group][1]) && sz[k] > 0) orc_adam(p[k], g[k], m[k], v[k], sz[k], step, lr, b1, b2, eps);
    return LRCN_OK;
}
int lrcn_adam_update(lrcn_ctx *c, float *const p[9], const float *const g[9], float *const m[9], float *const v[9], int step, float lr,
                     float b1, float b2, float eps) {
    if (!c || !p || !g || !m || !v || step < 1) return LRCN_EINVAL;
    int64_t sz[9];
    sizes_of(c, sz);
    for (int k = 0; k < 9; ++k)
        if (sz[k] > 0) orc_adam(p[k], g[k], m[k], v[k], sz[k], step, lr, b1, b2, eps);
    return LRCN_OK;
}
int lrcn_adam_update_flat(lrcn_ctx *c, float *w, const float *g, float *m, float *v, int64_t n, int step, float lr, float b1, float b2,
                          float eps, void *stream) {
    (void)stream;
    if (!c || step < 1 || n < 0) return LRCN_EINVAL;
    if (n == 0) return LRCN_OK;
    if (!w || !g || !m || !v) return LRCN_EINVAL;
    orc_adam(w, g, m, v, n, step, lr, b1, b2, eps);
    return LRCN_OK;
}
int lrcn_train_step(lrcn_ctx *c, float *const p[9], float *const g[9], float *const m[9], float *const v[9], const float *feats,
                    const int32_t *tokens, int T, int B, int norm_B, const lrcn_dropout *drop, int step, float lr, float b1, float b2,
                    float eps, double *loss_host) {
    int r = lrcn_loss_grad(c, (const float *const *)p, feats, tokens, T, B, norm_B, drop, g, loss_host);
    return r ? r : lrcn_adam_update(c, p, (const float *const *)g, m, v, step, lr, b1, b2, eps);
}

/* ---- data parallelism: one host "rank" only ---- */
int lrcn_comm_unique_id(void *id) { if (!id) return LRCN_EINVAL; memset(id, 0, LRCN_UNIQUE_ID_BYTES); return LRCN_OK; }
int lrcn_comm_probe(lrcn_ctx *c) { return c ? LRCN_OK : LRCN_EINVAL; }
int lrcn_comm_init(lrcn_ctx *c, int world, int rank, const void *id) {
    if (!c || !id) return LRCN_EINVAL;
    if (world != 1 || rank != 0) FAIL(c, LRCN_ESTATE, "liblrcn_cpu has no transport: world must be 1");
    return LRCN_OK;
}
int lrcn_comm_destroy(lrcn_ctx *c) { return c ? LRCN_OK : LRCN_EINVAL; }
/* sparse exchange of the embedding gradient: the export side lives inside the HIP lossgradient (the oracle's loss has no such hook), so the
 * host twin refuses to turn it on; the import side is the plain ordered sum it stands for */
int lrcn_set_embed_rows_buffer(lrcn_ctx *c, float *rows, int32_t *tok, int capacity_rows) {
    if (!c) return LRCN_EINVAL;
    (void)capacity_rows;
    if (rows || tok) FAIL(c, LRCN_ESTATE, "the host twin has no sparse embedding-gradient export");
    return LRCN_OK;
}
int lrcn_embed_grad_from_rows(lrcn_ctx *c, const float *rows, const int32_t *tok, int n_rows, float *grad_wembed, void *stream) {
    (void)stream;
    if (!c || !rows || !tok || !grad_wembed) return LRCN_EINVAL;
    if (n_rows < 1 || n_rows > 8192) FAIL(c, LRCN_EINVAL, "n_rows=%d outside [1, 8192]", n_rows);
    const int V = c->cfg.V, E = c->cfg.E;
    memset(grad_wembed, 0, sizeof(float) * (size_t)V * E);
    for (int r = 0; r < n_rows; ++r) { /* rows in order: the same sums the device takes per token */
        if (tok[r] < 0 || tok[r] >= V) FAIL(c, LRCN_EINVAL, "token id %d outside [0, %d)", tok[r], V);
        for (int e = 0; e < E; ++e) grad_wembed[(size_t)tok[r] + (size_t)e * V] += rows[(size_t)r * E + e];
    }
    return LRCN_OK;
}
int lrcn_comm_set_stream(lrcn_ctx *c, void *s) { (void)s; return c ? LRCN_OK : LRCN_EINVAL; } /* no streams on the host */
int lrcn_allreduce_grads(lrcn_ctx *c, float *const grads[9], int group) {
    return (c && grads && group >= -1 && group < LRCN_GRAD_GROUPS) ? LRCN_OK : LRCN_EINVAL; /* sum over one rank */
}
int lrcn_comm_join(lrcn_ctx *c) { return c ? LRCN_OK : LRCN_EINVAL; }

/* ---- decode ---- */
int lrcn_beam_search(lrcn_ctx *c, const float *const p[9], const float *feat, int K, int nword, int32_t *out_tokens, int *out_len,
                     float *out_prob) {
    if (!c || !p || !feat || !out_tokens || !out_len) return LRCN_EINVAL;
    if (K < 1 || K > 32 || K > c->cfg.max_B || K > c->cfg.V) FAIL(c, LRCN_EINVAL, "beam width K=%d", K);
    if (nword < 1 || nword > 256) FAIL(c, LRCN_EINVAL, "nword=%d outside [1,256]", nword);
    orc_model m = model_of(c, (float *const *)p);
    float pr = 0.f;
    *out_len = c->nl == 1 ? orc1_beam_search(&m, feat, K, nword, out_tokens, &pr) : orc_beam_search(&m, feat, K, nword, out_tokens, &pr);
    if (out_prob) *out_prob = pr;
    return LRCN_OK;
}
int lrcn_beam_search_batch(lrcn_ctx *c, const float *const p[9], const float *feats, int N, int K, int nword, int32_t *out_tokens,
                           int *out_len, float *out_prob) {
    if (!c || !p || !feats || !out_tokens || !out_len) return LRCN_EINVAL;
    if (N < 1 || (long)N * K > c->cfg.max_B) FAIL(c, LRCN_EINVAL, "N*K exceeds max_B");
    float *row = (float *)malloc(sizeof(float) * LRCN_CNNOUT);
    int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * (nword + 3));
    int r = LRCN_OK;
    for (int n = 0; n < N && !r; ++n) {
        for (int j = 0; j < LRCN_CNNOUT; ++j) row[j] = feats[(size_t)n + (size_t)N * j]; /* row n of N x 4096 column-major */
        int len = 0;
        float pr = 0.f;
        r = lrcn_beam_search(c, p, row, K, nword, tmp, &len, &pr);
        if (!r) {
            memset(out_tokens + (size_t)n * (nword + 2), 0, sizeof(int32_t) * (nword + 2));
            memcpy(out_tokens + (size_t)n * (nword + 2), tmp, sizeof(int32_t) * len);
            out_len[n] = len;
            if (out_prob) out_prob[n] = pr;
        }
    }
    free(row); free(tmp);
    return r;
}

/* ---- VGG-16 and the image front end ---- */
static const int kCout[13] = {64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512};
int lrcn_vgg_load(lrcn_ctx *c, const float *const cw[13], const float *const cb[13], const float *fc6_w, const float *fc6_b,
                  const float *fc7_w, const float *fc7_b) {
    if (!c || !cw || !cb || !fc6_w || !fc6_b || !fc7_w || !fc7_b) return LRCN_EINVAL;
    if (c->cfg.max_images < 1) FAIL(c, LRCN_ESTATE, "context was created with max_images = 0");
    if (c->vgg_loaded) FAIL(c, LRCN_ESTATE, "VGG weights already loaded");
    int cin = 3;
    for (int l = 0; l < 13; ++l) {
        c->cw[l] = (float *)xcopy(cw[l], sizeof(float) * 9 * (size_t)cin * kCout[l]);
        c->cb[l] = (float *)xcopy(cb[l], sizeof(float) * kCout[l]);
        cin = kCout[l];
    }
    c->fc6w = (float *)xcopy(fc6_w, sizeof(float) * 4096ull * 25088ull);
    c->fc6b = (float *)xcopy(fc6_b, sizeof(float) * 4096);
    c->fc7w = (float *)xcopy(fc7_w, sizeof(float) * 4096ull * 4096ull);
    c->fc7b = (float *)xcopy(fc7_b, sizeof(float) * 4096);
    c->vgg_loaded = 1;
    return LRCN_OK;
}
int lrcn_vgg_forward(lrcn_ctx *c, const float *x, int N, float *feats) {
    if (!c || !x || !feats) return LRCN_EINVAL;
    if (!c->vgg_loaded) FAIL(c, LRCN_ESTATE, "lrcn_vgg_load has not been called");
    if (N < 1 || N > c->cfg.max_images) FAIL(c, LRCN_EINVAL, "N=%d outside [1,%d]", N, c->cfg.max_images);
    orc_vgg v;
    for (int l = 0; l < 13; ++l) { v.conv_w[l] = c->cw[l]; v.conv_b[l] = c->cb[l]; }
    v.fc6_w = c->fc6w; v.fc6_b = c->fc6b; v.fc7_w = c->fc7w; v.fc7_b = c->fc7b;
    orc_vgg_forward(&v, x, 224, N, feats);
    return LRCN_OK;
}
int lrcn_set_average_image(lrcn_ctx *c, const float *avg) {
    if (!c) return LRCN_EINVAL;
    free(c->avg);
    c->avg = avg ? (float *)xcopy(avg, sizeof(float) * 224 * 224 * 3) : NULL;
    return LRCN_OK;
}
int lrcn_preprocess_u8(lrcn_ctx *c, const uint8_t *img, int N, const float mean[3], float *out) {
    if (!c || !img || !out || (!mean && !c->avg) || N < 1) return LRCN_EINVAL;
    const int S = 224;
    if (!c->avg) {
        orc_preprocess_u8(img, S, N, mean, out);
        return LRCN_OK;
    }
    for (int n = 0; n < N; ++n) /* pixel (row i, col j, ch) - averageImage(j, i, ch): lrcn.jl:770-771 */
        for (int ch = 0; ch < 3; ++ch)
            for (int j = 0; j < S; ++j)
                for (int i = 0; i < S; ++i)
                    out[(size_t)i + (size_t)S * (j + (size_t)S * (ch + 3 * (size_t)n))] =
                        (float)img[(((size_t)n * S + i) * S + j) * 3 + ch] - c->avg[(size_t)j + (size_t)S * i + (size_t)S * S * ch];
    return LRCN_OK;
}
int lrcn_vgg_forward_u8(lrcn_ctx *c, const uint8_t *img, int N, const float mean[3], float *feats) {
    if (!c || !img || !feats || (!mean && !c->avg)) return LRCN_EINVAL;
    float *x = (float *)malloc(sizeof(float) * (size_t)(N > 0 ? N : 1) * 224 * 224 * 3);
    int r = lrcn_preprocess_u8(c, img, N, mean, x);
    if (!r) r = lrcn_vgg_forward(c, x, N, feats);
    free(x);
    return r;
}
int lrcn_resize_crop_u8(lrcn_ctx *c, const uint8_t *src, const int64_t *offsets, const int *heights, const int *widths, const int *channels,
                        int N, uint8_t *out) {
    if (!c || !src || !offsets || !heights || !widths || !channels || !out || N < 1) return LRCN_EINVAL;
    const int64_t S = 224;
    for (int n = 0; n < N; ++n) {
        const int64_t h = heights[n], w = widths[n], ch = channels[n];
        if (h < 1 || w < 1 || (ch != 1 && ch != 3 && ch != 4)) FAIL(c, LRCN_EINVAL, "image %d: bad geometry", n);
        const int64_t sm = h < w ? h : w, nh = h * S / sm, nw = w * S / sm; /* lrcn.jl:756 */
        const uint8_t *im = src + offsets[n];
        for (int64_t r = 0; r < S; ++r)
            for (int64_t q = 0; q < S; ++q) {
                const int64_t R = r + (nh - S) / 2, Q = q + (nw - S) / 2; /* :758-760 */
                int64_t ny = (2 * R + 1) * h - nh, nx = (2 * Q + 1) * w - nw;
                if (ny < 0) ny = 0;
                if (nx < 0) nx = 0;
                const int64_t y0 = ny / (2 * nh), fy = ny - y0 * 2 * nh, x0 = nx / (2 * nw), fx = nx - x0 * 2 * nw;
                const int64_t y1 = y0 + 1 < h ? y0 + 1 : h - 1, x1 = x0 + 1 < w ? x0 + 1 : w - 1;
                for (int k = 0; k < 3; ++k) {
                    const int64_t cs = ch >= 3 ? k : 0; /* grey -> three channels :762-764 */
                    const int64_t p00 = im[(y0 * w + x0) * ch + cs], p01 = im[(y0 * w + x1) * ch + cs], p10 = im[(y1 * w + x0) * ch + cs],
                                  p11 = im[(y1 * w + x1) * ch + cs];
                    const int64_t top = (2 * nw - fx) * p00 + fx * p01, bot = (2 * nw - fx) * p10 + fx * p11;
                    out[(((size_t)n * S + r) * S + q) * 3 + k] = (uint8_t)(((2 * nh - fy) * top + fy * bot + 2 * nh * nw) / (4 * nh * nw));
                }
            }
    }
    return LRCN_OK;
}
int lrcn_vgg_forward_u8_blocks(lrcn_ctx *c, const uint8_t *img, int N, const float mean[3], int block_rows, int normalize, float *feats) {
    if (!c || !img || !feats) return LRCN_EINVAL;
    if (block_rows < 1 || N % block_rows) FAIL(c, LRCN_EINVAL, "block_rows=%d must divide N=%d", block_rows, N);
    for (int b = 0; b < N / block_rows; ++b) { /* images are independent: block by block through the one-batch entry point */
        int r = lrcn_vgg_forward_u8(c, img + (size_t)b * block_rows * 224 * 224 * 3, block_rows, mean, feats + (size_t)b * block_rows * LRCN_CNNOUT);
        if (r) return r;
        if (normalize) lrcn_normalize_features(c, feats + (size_t)b * block_rows * LRCN_CNNOUT, block_rows);
    }
    return LRCN_OK;
}
/* input feed (rev 4): "device" memory IS host memory here -- the crops are used where they lie, nothing is copied or queued */
int lrcn_host_alloc(void **p, size_t bytes) { if (!p) return LRCN_EINVAL; *p = malloc(bytes ? bytes : 16); return *p ? LRCN_OK : LRCN_ENOMEM; }
int lrcn_host_free(void *p) { free(p); return LRCN_OK; }
int lrcn_upload_crops(lrcn_ctx *c, const uint8_t *host_u8, int N, const uint8_t **dev_out) {
    if (!c || !host_u8 || !dev_out) return LRCN_EINVAL;
    if (c->cfg.max_images < 1) FAIL(c, LRCN_ESTATE, "context was created with max_images = 0");
    if (N < 1 || N > c->cfg.max_images) FAIL(c, LRCN_EINVAL, "N=%d outside [1,%d]", N, c->cfg.max_images);
    *dev_out = host_u8;
    return LRCN_OK;
}
int lrcn_upload_wait(lrcn_ctx *c) { return c ? LRCN_OK : LRCN_EINVAL; }
int lrcn_normalize_features(lrcn_ctx *c, float *feats, int N) {
    if (!c || !feats || N < 1) return LRCN_EINVAL;
    for (int n = 0; n < N; ++n) { /* input / sum(input)  lrcn.jl:595-597 (float32 sum) */
        float s = 0.0f;
        for (int j = 0; j < LRCN_CNNOUT; ++j) s += feats[(size_t)n + (size_t)N * j];
        for (int j = 0; j < LRCN_CNNOUT; ++j) feats[(size_t)n + (size_t)N * j] /= s;
    }
    return LRCN_OK;
}
int lrcn_conv3x3(lrcn_ctx *c, const float *x, int W, int H, int Cin, int N, const float *w, const float *b, int Cout, int relu, int pool,
                 float *y) {
    if (!c || !x || !w || !b || !y) return LRCN_EINVAL;
    if (W < 2 || H < 2 || (W & 1) || (H & 1) || Cin < 1 || Cout < 1 || N < 1) FAIL(c, LRCN_EINVAL, "conv3x3: W,H must be even, sizes positive");
    if (!pool) {
        orc_conv3x3(x, W, H, Cin, N, w, b, Cout, relu, y);
        return LRCN_OK;
    }
    float *t = (float *)malloc(sizeof(float) * (size_t)W * H * Cout * N);
    orc_conv3x3(x, W, H, Cin, N, w, b, Cout, relu, t);
    orc_pool2(t, W, H, Cout, N, y);
    free(t);
    return LRCN_OK;
}
int lrcn_conv1_fused(lrcn_ctx *c, const uint8_t *img, int N, int S, const float mean[3], const float *w11, const float *b11, const float *w12,
                     const float *b12, float *y) {
    /* read_image_data's arithmetic, conv1_1 + ReLU, conv1_2 + ReLU, pool (lrcn.jl:770, 724-726) -- with orc_set_emulate_bf16(1) rounded where
     * the bf16 stack rounds (orc_conv3x3 rounds its operands and its result itself) */
    if (!c || !img || !mean || !w11 || !b11 || !w12 || !b12 || !y) return LRCN_EINVAL;
    if (N < 1 || S < 16 || (S % 16)) FAIL(c, LRCN_EINVAL, "conv1_fused: S must be a multiple of 16, N >= 1");
    const size_t px = (size_t)S * S * N;
    float *x = (float *)malloc(sizeof(float) * px * 3), *t1 = (float *)malloc(sizeof(float) * px * 64), *t2 = (float *)malloc(sizeof(float) * px * 64);
    if (!x || !t1 || !t2) { free(x); free(t1); free(t2); FAIL(c, LRCN_ENOMEM, "conv1_fused scratch"); }
    orc_preprocess_u8(img, S, N, mean, x);
    orc_conv3x3(x, S, S, 3, N, w11, b11, 64, 1, t1);
    orc_conv3x3(t1, S, S, 64, N, w12, b12, 64, 1, t2);
    orc_pool2(t2, S, S, 64, N, y);
    free(x); free(t1); free(t2);
    return LRCN_OK;
}
int lrcn_train_step_dp(lrcn_ctx *c, float *const p[9], float *const g[9], float *const m[9], float *const v[9], const uint8_t *img_u8,
                       const float mean[3], int normalize, float *feats, const int32_t *tokens, int T, int B, int norm_B,
                       const lrcn_dropout *drop, int step, float lr, float b1, float b2, float eps, double *loss_host) {
    if (!c || !feats) return LRCN_EINVAL;
    if (img_u8) {
        int r = lrcn_vgg_forward_u8(c, img_u8, B, mean, feats);
        if (r) return r;
        if (normalize) lrcn_normalize_features(c, feats, B);
    }
    return lrcn_train_step(c, p, g, m, v, feats, tokens, T, B, norm_B, drop, step, lr, b1, b2, eps, loss_host);
}

/* ---- GPU-only pieces ---- */
int lrcn_vgg_calibrate(lrcn_ctx *c, const uint8_t *img, int N, const float mean[3], float margin) {
    (void)img; (void)N; (void)mean; (void)margin;
    if (!c) return LRCN_EINVAL;
    FAIL(c, LRCN_ESTATE, "the e4m3 convolution stack exists on the GPU only");
}
int lrcn_conv3x3_fp8(lrcn_ctx *c, const float *x, int W, int H, int Cin, int N, const float *w, const float *b, int Cout, int relu, int pool,
                     float sa_in, float sa_out, float *y, float *sw_out) {
    (void)x; (void)W; (void)H; (void)Cin; (void)N; (void)w; (void)b; (void)Cout; (void)relu; (void)pool; (void)sa_in; (void)sa_out; (void)y; (void)sw_out;
    if (!c) return LRCN_EINVAL;
    FAIL(c, LRCN_ESTATE, "the e4m3 convolution stack exists on the GPU only");
}
int lrcn_profile(lrcn_ctx *c, int enable) { (void)enable; return c ? LRCN_OK : LRCN_EINVAL; }
int lrcn_profile_segment(lrcn_ctx *c, int segment, double *ms, int64_t *n, double *bytes) {
    if (!c || !ms || !n || !bytes || segment < 0 || segment >= LRCN_SEG_COUNT) return LRCN_EINVAL;
    *ms = 0.0; *n = 0; *bytes = 0.0; /* the host twin is synchronous and unprofiled */
    return LRCN_OK;
}
int lrcn_profile_get(lrcn_ctx *c, double *ms, int64_t *n) {
    if (!c || !ms || !n) return LRCN_EINVAL;
    *ms = 0.0; *n = 0;
    return LRCN_OK;
}
int lrcn_bench_conv(lrcn_ctx *c, int N, int S, int Cin, int Cout, int pool, int iters, double *ms) {
    (void)N; (void)S; (void)Cin; (void)Cout; (void)pool; (void)iters; (void)ms;
    if (!c) return LRCN_EINVAL;
    FAIL(c, LRCN_ESTATE, "kernel-development aid of the GPU library");
}
int lrcn_debug_stamps(lrcn_ctx *c, unsigned long long *host_out, int64_t n) {
    (void)host_out; (void)n;
    if (!c) return LRCN_EINVAL;
    FAIL(c, LRCN_ESTATE, "kernel-development aid of the GPU library");
}
int lrcn_bench_gemm(lrcn_ctx *c, int M, int N, int K, int iters, double *ms) {
    (void)M; (void)N; (void)K; (void)iters; (void)ms;
    if (!c) return LRCN_EINVAL;
    FAIL(c, LRCN_ESTATE, "kernel-development aid of the GPU library");
}
