#!/usr/bin/env python3
"""BASELINE.json configs[4] (C5) END TO END on a fixture with an answer: synthetic scenes -> VGG-16 (fp8 / bf16 / f32 stack) -> fc7 ->
bf16 LRCN-2f beam-search-5, on a model TRAINED here (a few hundred Adam steps of this library's own train step) to caption those scenes.

BASELINE.md section 3 states C5's tolerance as "beam-search top caption identical on >= 95 % of fixture images (else BLEU within +-0.5 on
the fixture set)".  Random LRCN weights cannot test that: an untrained decoder's top-2 margins are exponentially distributed around
sigma / sqrt(2 ln V), so ANY perturbation of the features flips a word somewhere in 30 steps on most images (measured: bf16 vs f32 VGG
features, 1 % apart, already disagree on 45 % of captions).  A trained decoder is decisive where the data are, which is the regime the
tolerance was written for.  The fixture: NS scene classes (a fixed layout of coloured discs per class; an instance = the layout with
jittered positions / colours + pixel noise), one caption per class (T words drawn once), training instances through the bf16 VGG,
held-out instances through each VGG precision.

    python tools/c5_fixture.py            -> one JSON line: agreement between the precisions, BLEU of each against the class captions

Used by tests/test_gpu_config5.py (asserted) and tools/caption_bench.py (reported in the C5 bench line)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

# fc7 of the He-normal synthetic VGG is O(100).  Round 5 scaled it by 0.01 (features O(1)): the image embedding x_cnn = feats * Wcnn then
# starts at ~0.8 per element and LSTM-2's gates near saturation, and the loss sits on a plateau at ln(classes) that the optimiser leaves after
# 600..1100 steps, at a step that moves with the last bits of the gradients.  At 0.003 (measured in round 6, tools/r06/c5_spread2.sh,
# profiles/r06_c5_fixture_spread.txt: scales 0.0003..0.1 x 5 row-sampling seeds) there is no plateau at all: 9.27 -> 3.4 -> 2.1 -> 1.1 ->
# 0.5 at steps 0 / 100 / 200 / 300 / 400 for every seed.  (The reference's own features are divided by their sum: O(1e-4).)
FEAT_SCALE = float(os.environ.get("C5_FEAT_SCALE", "0.003"))
TARGET_LOSS = float(os.environ.get("C5_TARGET", "0.05"))     # training stops on this loss (two consecutive checks), not on a step count
MAX_STEPS = int(os.environ.get("C5_MAX_STEPS", "8000"))      # budget: > 5x the slowest escape from the ln(classes) plateau ever observed


def scene_images(classes, per_class, seed, jitter=6, noise=6.0):
    """-> (uint8 crops [classes * per_class, 224, 224, 3], class id per crop).  Class c = a background colour + 3..6 discs (position,
    radius, colour) from a generator seeded by c alone; an instance jitters every disc by <= `jitter` pixels and adds N(0, noise) per pixel."""
    yy, xx = np.mgrid[0:224, 0:224]
    out = np.zeros((classes * per_class, 224, 224, 3), np.uint8)
    ids = np.zeros(classes * per_class, np.int64)
    inst = np.random.default_rng(seed)
    for c in range(classes):
        g = np.random.default_rng(1000 + c)
        bg = g.integers(0, 256, size=3).astype(np.float32)
        discs = [(g.integers(20, 204), g.integers(20, 204), g.integers(18, 70), g.integers(0, 256, size=3).astype(np.float32)) for _ in range(g.integers(3, 7))]
        for k in range(per_class):
            img = np.ones((224, 224, 3), np.float32) * bg
            for cx, cy, r, col in discs:
                dx, dy = inst.integers(-jitter, jitter + 1, size=2)
                img[(xx - cx - dx) ** 2 + (yy - cy - dy) ** 2 < r * r] = col
            img += inst.standard_normal(img.shape).astype(np.float32) * noise
            out[c * per_class + k] = np.clip(img, 0, 255).astype(np.uint8)
            ids[c * per_class + k] = c
    return out, ids


def class_captions(classes, T, V, seed=11):
    """One caption of T word ids (>= 3) per class; the first word differs between classes as often as V allows."""
    return np.random.default_rng(seed).integers(3, V, size=(classes, T)).astype(np.int32)


def vgg_features(L, lrcn_amd, dtype, w, imgs_u8, calib=None, V=17):
    import torch
    n = imgs_u8.shape[0]
    ctx = L.Context(8, 8, 8, V, max_B=2, max_T=1, vgg_dtype=dtype, max_images=n)
    L.vgg_load(ctx, *w)
    dev = torch.as_tensor(imgs_u8).cuda()
    if dtype == lrcn_amd.LRCN_FP8:
        L.vgg_calibrate(ctx, torch.as_tensor(calib if calib is not None else imgs_u8[:32]).cuda())
    f = L.from_jl(L.convnet_u8(ctx, dev)).copy()
    ctx.close()
    return f * np.float32(FEAT_SCALE)


def train_decoder(L, lrcn_amd, ctx, feats, caps_of_rows, steps=400, B=32, seed=3, target=None, max_steps=None, deterministic=True, every=100):
    """Adam (Knet defaults) on rows sampled from (feats, captions): this library's lrcn_train_step, no dropout.  -> (param, loss trace,
    steps taken).

    The decode-agreement assertions downstream must not depend on luck in an optimiser, so the training is made to
    converge BY CONSTRUCTION: (1) LRCN_OPT_DETERMINISTIC -- every sum on the lossgradient route in a fixed order, so the trajectory is a
    function of the inputs alone (the default route's float atomics move the step at which the plateau at ln(classes) is left by hundreds
    of steps between same-seed runs); (2) with `target`, training stops on a loss threshold (two consecutive checks below it, at least
    `steps` steps) under a budget of `max_steps`, instead of after a fixed count."""
    import torch
    from lrcn_amd import _lib
    if deterministic:
        ctx.set_option(_lib.LRCN_OPT_DETERMINISTIC, 1)
    rng = np.random.default_rng(seed)
    param = L.initweights(ctx, seed=42)
    optim = L.initparams(param)
    grads = [L.jl_empty(*t.shape) for t in param]
    trace, below, k = [], 0, 0
    budget = max(steps, max_steps or steps)
    while k < budget:
        rows = rng.integers(0, feats.shape[0], size=B)
        check = k % every == 0 or k == budget - 1
        val = L.train_step(ctx, param, optim, grads, L.to_jl(feats[rows]), np.ascontiguousarray(caps_of_rows[rows].T), pdrop=0.0, seed=k,
                           want_loss=check)
        k += 1
        if val is not None:
            trace.append(float(val))
            below = below + 1 if (target is not None and val < target) else 0
            if target is not None and below >= 2 and k >= steps:
                break
    torch.cuda.synchronize()
    if deterministic:
        ctx.set_option(_lib.LRCN_OPT_DETERMINISTIC, 0)
    return param, trace, k


def run_fixture(classes=16, train_per_class=8, test_per_class=4, T=8, steps=400, K=5, nword=30, V=10640, precisions=("f32", "bf16", "fp8"),
                target=TARGET_LOSS, max_steps=MAX_STEPS, deterministic=os.environ.get("C5_DET", "1") != "0"):
    """-> dict: agreement of the decoded captions between VGG precisions on the held-out scenes, BLEU-1..4 of each precision against the
    class captions (this repo's port of eval/multi-bleu.perl), and the evidence that the fixture is not trivial."""
    import lrcn_amd
    from lrcn_amd import bleu
    from lrcn_amd import lrcn as L
    dts = {"f32": lrcn_amd.LRCN_F32, "bf16": lrcn_amd.LRCN_BF16, "fp8": lrcn_amd.LRCN_FP8}
    w = L.synthetic_vgg_weights(seed=1)
    tr_img, tr_id = scene_images(classes, train_per_class, seed=1)
    te_img, te_id = scene_images(classes, test_per_class, seed=2)
    caps = class_captions(classes, T, V)
    tr_feat = vgg_features(L, lrcn_amd, lrcn_amd.LRCN_BF16, w, tr_img)
    N = te_img.shape[0]
    ctx = L.Context(1000, 1000, 1000, V, max_B=max(32, N * K), max_T=T, lstm_dtype=lrcn_amd.LRCN_BF16)
    param, trace, taken = train_decoder(L, lrcn_amd, ctx, tr_feat, caps[tr_id], steps=steps, target=target if (target is None or target > 0) else None, max_steps=max_steps,
                                        deterministic=deterministic, seed=int(os.environ.get("C5_SEED", "3")))
    dec, feats = {}, {}
    for p in precisions:
        feats[p] = vgg_features(L, lrcn_amd, dts[p], w, te_img, calib=tr_img[:32])
        dec[p] = L.beam_search_batch(ctx, param, L.to_jl(feats[p]), K, nword)
    ctx.close()

    def words(tokens):   # the caption as generate prints it: after bos, up to the first eos (lrcn.jl:634-640)
        out = []
        for t in tokens[1:]:
            if t == L.EOS:
                break
            out.append(str(int(t)))
        return " ".join(out)

    refs = [[" ".join(str(int(t)) for t in caps[c])] for c in te_id]
    res = {"fixture": "%d scene classes, %d training / %d held-out instances each, captions of %d words, decoder trained %d Adam steps here "
                      "(loss %.3f -> %.3f; stop rule: loss < %g twice, >= %d steps, budget %d; %s sums)"
                      % (classes, train_per_class, test_per_class, T, taken, trace[0], trace[-1], target if target is not None else 0.0, steps,
                         max(steps, max_steps or steps), "fixed-order (LRCN_OPT_DETERMINISTIC)" if deterministic else "atomic"),
           "n_images": int(N), "train_steps": int(taken), "train_deterministic": bool(deterministic), "feat_scale": FEAT_SCALE,
           "train_loss_trace": [round(x, 3) for x in trace]}
    for p in precisions:
        hyp = [words(t) for t, _ in dec[p]]
        res["bleu_" + p] = [round(x, 2) for x in bleu.multi_bleu(hyp, [[r[0] for r in refs]])["bleu"]]
        res["correct_" + p] = float(np.mean([h == r[0] for h, r in zip(hyp, refs)]))
        res["distinct_captions_" + p] = len(set(hyp))
    base = precisions[0]
    for p in precisions[1:]:
        f, r = feats[p], feats[base]
        res["features_%s_vs_%s" % (p, base)] = {"cos_min": float(min((f[n] * r[n]).sum() / (np.linalg.norm(f[n]) * np.linalg.norm(r[n])) for n in range(N))),
                                              "rel_l2_max": float(max(np.linalg.norm(f[n] - r[n]) / np.linalg.norm(r[n]) for n in range(N)))}
    for a, b in [(x, y) for i, x in enumerate(precisions) for y in precisions[:i]]:
        same = [dec[a][n][0] == dec[b][n][0] for n in range(N)]
        gaps = [abs(np.log(dec[a][n][1] + 1e-300) - np.log(dec[b][n][1] + 1e-300)) for n in range(N) if not same[n]]
        res["%s_vs_%s" % (a, b)] = {"top_caption_identical": float(np.mean(same)), "max_logp_gap_of_mismatches": float(max(gaps)) if gaps else 0.0,
                                    "bleu4_diff": round(abs(res["bleu_" + a][3] - res["bleu_" + b][3]), 2)}
    return res


if __name__ == "__main__":
    print(json.dumps(run_fixture(steps=int(os.environ.get("C5_STEPS", "400")), classes=int(os.environ.get("C5_CLASSES", "16")))))
