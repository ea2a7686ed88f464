/*
 * selftest.c -- drives every entry point of the CPU oracle on small shapes; built with AddressSanitizer + UBSan by
 * `make -C oracle asan` (the sanitizer pass of this repo runs on the CPU checker; GPU ASAN is unavailable).
 * TEST INFRASTRUCTURE ONLY, like everything under oracle/.  Exit code 0 = no sanitizer report and the analytic checks hold:
 *   zero Wout/bout -> loss = ln V (the deck's epoch-0 points);  gradient buffers fully overwritten;
 *   beam search output starts with bos and has length <= nword + 2.
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "lrcn_oracle.h"

static float *falloc(size_t n) {
    float *p = (float *)calloc(n ? n : 1, sizeof(float));
    if (!p) { fprintf(stderr, "out of memory\n"); exit(2); }
    return p;
}

static void model_alloc(orc_model *m, int E, int H1, int H2, int V) {
    int64_t sz[9];
    orc_param_sizes(E, H1, H2, V, sz);
    m->E = E; m->H1 = H1; m->H2 = H2; m->V = V;
    float **slots[9] = {&m->W1, &m->b1, &m->W2, &m->b2, &m->Wproj, &m->Wcnn, &m->Wembed, &m->Wout, &m->bout};
    for (int k = 0; k < 9; ++k) *slots[k] = falloc((size_t)sz[k]);
}
static void model_free(orc_model *m) {
    free(m->W1); free(m->b1); free(m->W2); free(m->b2); free(m->Wproj); free(m->Wcnn); free(m->Wembed); free(m->Wout); free(m->bout);
}

int main(void) {
    const int E = 6, H1 = 8, H2 = 6, V = 19, B = 3, T = 4, S = T + 1;
    int fails = 0;
    orc_model m, g;
    model_alloc(&m, E, H1, H2, V);
    model_alloc(&g, E, H1, H2, V);
    orc_init_weights(&m, 7);
    float *feats = falloc((size_t)B * ORC_CNNOUT);
    for (int i = 0; i < B * ORC_CNNOUT; ++i) feats[i] = 0.01f * (float)((i * 37) % 11 - 5);
    int32_t tok[4 * 3];
    for (int i = 0; i < T * B; ++i) tok[i] = (i * 5 + 3) % V;
    float *m1 = falloc((size_t)S * B * E), *m2 = falloc((size_t)S * B * H2);
    for (int i = 0; i < S * B * E; ++i) m1[i] = (i % 3) ? 1.0f / 0.6f : 0.0f;
    for (int i = 0; i < S * B * H2; ++i) m2[i] = (i % 4) ? 1.0f / 0.6f : 0.0f;

    double l0 = orc_loss(&m, feats, tok, T, B, B, NULL, NULL, NULL);
    double l1 = orc_loss(&m, feats, tok, T, B, 2 * B, m1, m2, &g);
    if (!(l0 > 0.0) || !isfinite(l1)) { fprintf(stderr, "loss not finite\n"); ++fails; }
    /* T = 0: only the eos step */
    double lz = orc_loss(&m, feats, NULL, 0, B, B, NULL, NULL, &g);
    if (!isfinite(lz)) ++fails;
    /* zero output layer: uniform softmax, loss = ln V exactly (to float rounding) */
    int64_t sz[9];
    orc_param_sizes(E, H1, H2, V, sz);
    memset(m.Wout, 0, sizeof(float) * (size_t)sz[7]);
    memset(m.bout, 0, sizeof(float) * (size_t)sz[8]);
    double lu = orc_loss(&m, feats, tok, T, B, B, NULL, NULL, NULL);
    if (fabs(lu - log((double)V)) > 1e-6) { fprintf(stderr, "ln V check: %.9f vs %.9f\n", lu, log((double)V)); ++fails; }
    orc_init_weights(&m, 7);

    float *logits = falloc((size_t)S * B * V);
    orc_forward_logits(&m, feats, tok, T, B, logits);

    /* lstm / lrcn step */
    float *x = falloc((size_t)B * E), *h = falloc((size_t)B * H1), *c = falloc((size_t)B * H1), *ho = falloc((size_t)B * H1),
          *co = falloc((size_t)B * H1), *ga = falloc((size_t)B * 4 * H1);
    for (int i = 0; i < B * E; ++i) x[i] = 0.1f * (float)(i % 7 - 3);
    orc_lstm(m.W1, m.b1, E, H1, B, x, h, c, ho, co, ga);
    float *h2 = falloc((size_t)B * H2), *c2 = falloc((size_t)B * H2), *xc = falloc((size_t)B * (H2 / 2)), *lg = falloc((size_t)B * V);
    orc_lrcn_step(&m, B, h, c, h2, c2, xc, x, m1, m2, lg);

    /* adam */
    float *mom = falloc((size_t)sz[0]), *var = falloc((size_t)sz[0]);
    orc_adam(m.W1, g.W1, mom, var, sz[0], 1, 1e-3f, 0.9f, 0.999f, 1e-8f);

    /* beam search */
    int32_t out[16];
    float p = 0.f;
    const int nword = 6;
    int len = orc_beam_search(&m, feats, 3, nword, out, &p);
    if (len < 2 || len > nword + 2 || out[0] != ORC_BOS || !(p > 0.f)) { fprintf(stderr, "beam: len %d p %g\n", len, p); ++fails; }

    /* VGG operators on a small tensor */
    const int W = 6, Hh = 4, Cin = 3, Cout = 5, N = 2;
    float *cx = falloc((size_t)W * Hh * Cin * N), *cw = falloc(9 * Cin * Cout), *cb = falloc(Cout), *cy = falloc((size_t)W * Hh * Cout * N),
          *py = falloc((size_t)(W / 2) * (Hh / 2) * Cout * N);
    for (int i = 0; i < W * Hh * Cin * N; ++i) cx[i] = (float)(i % 13) - 6.f;
    for (int i = 0; i < 9 * Cin * Cout; ++i) cw[i] = 0.05f * (float)(i % 9 - 4);
    orc_conv3x3(cx, W, Hh, Cin, N, cw, cb, Cout, 1, cy);
    orc_pool2(cy, W, Hh, Cout, N, py);
    float *fw = falloc(7 * 11), *fb = falloc(7), *fx = falloc(11 * 2), *fy = falloc(7 * 2);
    orc_fc(fw, fb, 7, 11, 2, fx, 1, fy);
    uint8_t *img = (uint8_t *)calloc((size_t)2 * 8 * 8 * 3, 1);
    float mean[3] = {1.f, 2.f, 3.f};
    float *pre = falloc((size_t)8 * 8 * 3 * 2);
    orc_preprocess_u8(img, 8, 2, mean, pre);
    if (pre[0] != -1.f) ++fails;

    /* the same calls with the bf16 emulation on (rounded operand copies of every contraction, rounded conv operands): same bounds, other
     * code paths; the emulated loss stays within bf16 distance of the plain one */
    {
        const double l_plain = orc_loss(&m, feats, tok, T, B, B, m1, m2, NULL);
        orc_set_emulate_bf16(1);
        const double l_emu = orc_loss(&m, feats, tok, T, B, B, m1, m2, &g);
        orc_forward_logits(&m, feats, tok, T, B, logits);
        orc_lrcn_step(&m, B, h, c, h2, c2, xc, x, m1, m2, lg);
        orc_conv3x3(cx, W, Hh, Cin, N, cw, cb, Cout, 1, cy);
        orc_fc(fw, fb, 7, 11, 2, fx, 1, fy);
        orc_set_emulate_bf16(0);
        if (!(l_emu == l_emu) || l_emu == l_plain || (l_emu - l_plain) / l_plain > 2e-2 || (l_plain - l_emu) / l_plain > 2e-2) {
            fprintf(stderr, "emulated loss %g vs %g\n", l_emu, l_plain);
            ++fails;
        }
        if (orc_bf16_round(1.00390625f) != 1.0f || orc_bf16_round(1.01171875f) != 1.015625f) ++fails; /* ties to even */
    }

    free(img); free(pre); free(fw); free(fb); free(fx); free(fy); free(cx); free(cw); free(cb); free(cy); free(py);
    free(mom); free(var); free(h2); free(c2); free(xc); free(lg); free(x); free(h); free(c); free(ho); free(co); free(ga);
    free(logits); free(m1); free(m2); free(feats);
    model_free(&m); model_free(&g);
    if (fails) { fprintf(stderr, "selftest: %d check(s) failed\n", fails); return 1; }
    printf("oracle selftest ok (threads %d)\n", orc_num_threads());
    return 0;
}
