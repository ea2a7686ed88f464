"""GPU: the lrcn.jl-shaped driver (tools/lrcn.py) end to end on a tiny synthetic COCO-style dataset: tokenise -> minibatch
-> train two epochs through liblrcn_hip (loss must fall) -> checkpoint with Adam state -> reload -> beam-search captions
-> BLEU of the generated file against the captions' own references."""
import importlib
import json
import os
import sys

import numpy as np
import pytest

from lrcn_amd import bleu, formats as fmt

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_train_checkpoint_generate_round_trip(tmp_path, capsys):
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    cli = importlib.import_module("lrcn")
    rng = np.random.default_rng(0)
    # 12 "scenes": the image feature determines the caption, so a working trainer drives the loss well below ln V
    nouns, verbs = ["dog", "cat", "man", "bird"], ["runs", "sleeps", "jumps"]
    anns, feats = [], {}
    for img in range(120):
        n, v = nouns[img % 4], verbs[(img // 4) % 3]
        f = np.zeros(4096, np.float32)
        f[(img % 4) * 100:(img % 4) * 100 + 50] = 1.0
        f[1000 + ((img // 4) % 3) * 100:1000 + ((img // 4) % 3) * 100 + 50] = 1.0
        feats[img] = f / f.sum()
        for _ in range(2):
            anns.append({"image_id": img, "caption": "A %s %s ." % (n, v)})
    tr, va = str(tmp_path / "captions_train.json"), str(tmp_path / "captions_val.json")
    for p in (tr, va):
        with open(p, "w") as fh:
            json.dump({"annotations": anns}, fh)
    fp = str(tmp_path / "feats.npz")
    fmt.save_features(fp, feats)
    ck = str(tmp_path / "m.npz")
    common = ["--coco", "--datafiles", tr, va, "--features", fp, fp, "--hidden", "64", "64", "--embed", "64", "--batchsize", "10",
              "--atype", "f32", "--seed", "3"]
    assert cli.main(common + ["--train", "--epochs", "12", "--lr", "0.01", "--savefile", ck, "--dropout", "0.0"]) == 0
    out = capsys.readouterr().out
    losses = [float(ln.split(":loss,")[1].split(")")[0].split(",")[0]) for ln in out.splitlines() if ln.startswith("(:epoch")]
    assert len(losses) == 12 and losses[-1] < losses[0] and losses[-1] < 1.0, losses  # ln(10) = 2.3 at initialisation
    model, vocab, adam, _ = fmt.load_checkpoint(ck)
    assert len(vocab) == 3 + 8 and adam is not None and adam["step"] > 0 and model[6].shape == (len(vocab), 64)
    outdir = str(tmp_path / "eval")
    assert cli.main(common + ["--loadfile", ck, "--generate", "8", "--capnumber", "12", "--beam_width", "3", "--out", outdir]) == 0
    cands = open(os.path.join(outdir, "candidates.txt")).read().splitlines()
    ids = [int(x) for x in open(os.path.join(outdir, "candidate_ids.txt")).read().split()]
    assert len(cands) == len(ids) == 12 and all(c.endswith(" .") for c in cands)
    refs = bleu.coco_reference_lines(anns, ids, nrefs=2)
    r = bleu.multi_bleu(cands, refs)
    assert r["bleu"][0] > 80.0, (cands, r)  # the trained model reproduces "a <noun> <verb> ." for its image
