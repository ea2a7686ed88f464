"""GPU: BASELINE.json configs[4] ("C5": fp8 MFMA conv stack + bf16 LSTM, beam-search-5) END TO END against its stated tolerance
("beam-search top caption identical on >= 95 % of fixture images (else BLEU within +-0.5 on the fixture set)";
VERDICT r4 weak 1d: the composition fp8 VGG -> fc7 -> beam-5 had never been compared with the f32 / bf16 decode of the same images).

The fixture (tools/c5_fixture.py): 16 classes of synthetic scenes, each with its own 8-word caption; the decoder is TRAINED here -- this
library's own lrcn_train_step on bf16-VGG features of the training instances, under LRCN_OPT_DETERMINISTIC (the trajectory is a function of
the inputs: bit-identical across runs, tools/r06) and UNTIL the loss is below 0.05 on two consecutive checks (>= 600 steps, budget 8000;
round 5 trained a fixed 1500 steps through float atomics across a plateau and failed on the driver's box) -- so that its word distributions are decisive
where the data are (an untrained decoder flips a word somewhere in 30 steps under ANY feature perturbation; measured in round 5: bf16 vs
f32 features, 1 % apart, disagreed on 45 % of captions with random weights).  64 held-out instances then pass through the f32
(exact-fp32 MFMA), bf16 and fp8 (e4m3 conv2_2..conv5_3, calibrated on training images) stacks and the same bf16 batched beam search
(generate / beam_search, lrcn.jl:585-678; K = 5, nword = 30, E = H = 1000, V = 10640)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.gpu


def test_config5_fp8_vgg_to_beam5_captions_match_the_f32_and_bf16_decodes():
    import c5_fixture
    r = c5_fixture.run_fixture(steps=600)
    print(r)
    assert r["n_images"] == 64 and r["train_deterministic"]
    # the fixture is non-trivial and solved: training converged (stop rule: loss < 0.05 twice), 16 different captions come out, and they
    # are the classes' captions
    assert r["train_loss_trace"][0] > 9.0 and r["train_loss_trace"][-1] < 0.05 and r["train_steps"] < 8000, (r["train_steps"], r["train_loss_trace"])
    assert r["distinct_captions_f32"] == 16 and r["correct_f32"] >= 0.95, r
    # fp8 features: the tolerance test_gpu_vgg_parity.py states for the stack (cosine >= 0.99, relative L2 <= 0.15 vs f32)
    assert r["features_fp8_vs_f32"]["cos_min"] >= 0.99 and r["features_fp8_vs_f32"]["rel_l2_max"] <= 0.15, r["features_fp8_vs_f32"]
    # BASELINE.md section 3, for fp8 and for bf16, against the f32 decode and against each other
    for pair in ("fp8_vs_f32", "fp8_vs_bf16", "bf16_vs_f32"):
        p = r[pair]
        assert p["top_caption_identical"] >= 0.95 or p["bleu4_diff"] <= 0.5, (pair, p)
        assert p["max_logp_gap_of_mismatches"] < 1.0, (pair, p)   # a caption may only differ where the two decodes were a near-tie
    for prec in ("bf16", "fp8"):
        assert all(abs(a - b) <= 0.5 for a, b in zip(r["bleu_" + prec], r["bleu_f32"])), (prec, r["bleu_" + prec], r["bleu_f32"])
