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


def _scene_dataset(tmp_path, n_img=40, with_images=False):
    """Captions determined by the image 'scene'; optionally the images themselves (colour = noun, stripe direction = verb)."""
    rng = np.random.default_rng(1)
    nouns, verbs = ["dog", "cat", "man", "bird"], ["runs", "sleeps", "jumps"]
    anns, feats = [], {}
    imgdir = str(tmp_path / "img")
    if with_images:
        from PIL import Image
        os.makedirs(imgdir, exist_ok=True)
    for img in range(n_img):
        a, b = img % 4, (img // 4) % 3
        f = np.zeros(4096, np.float32)
        f[a * 100:a * 100 + 50] = 1.0
        f[1000 + b * 100:1000 + b * 100 + 50] = 1.0
        feats[img] = f / f.sum()
        for _ in range(2):
            anns.append({"image_id": img, "caption": "A %s %s ." % (nouns[a], verbs[b])})
        if with_images:
            h, w = 60 + 4 * (img % 5), 80 + 3 * (img % 7)
            yy, xx = np.mgrid[0:h, 0:w]
            base = np.array([[220, 40, 40], [40, 200, 60], [50, 60, 230], [210, 200, 40]][a], np.float32)
            stripe = [np.sin(xx / 3.0), np.sin(yy / 3.0), np.sin((xx + yy) / 4.0)][b]
            arr = np.clip(base[None, None, :] * (0.6 + 0.4 * stripe[:, :, None]) + rng.normal(0, 4, (h, w, 3)), 0, 255).astype(np.uint8)
            Image.fromarray(arr).save(os.path.join(imgdir, "%d.jpg" % img), quality=95)
    tr = str(tmp_path / "captions_train.json")
    with open(tr, "w") as fh:
        json.dump({"annotations": anns}, fh)
    fp = str(tmp_path / "feats.npz")
    fmt.save_features(fp, feats)
    return tr, fp, imgdir


def _epoch_losses(out):
    return [[float(v) for v in ln.split(":loss,")[1].split(")")[0].split(",")] for ln in out.splitlines() if ln.startswith("(:epoch")]


def test_training_from_images_equals_training_on_extracted_features(tmp_path, capsys):
    # `--cnn --train --imagedir`: ids -> decoded images -> resize / crop on the device -> VGG beside the LSTM step (the trainer's side
    # stream) -> features / sum -> lossgradient + update!.  Same trajectory as the reference's two-stage way: --extfeatures first, then
    # --train on the (sum-normalised) feature table.  fp32 throughout so that the VGG's batch composition does not matter.
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    cli = importlib.import_module("lrcn")
    tr, _, imgdir = _scene_dataset(tmp_path, with_images=True)
    base = ["--coco", "--datafiles", tr, tr, "--hidden", "32", "32", "--embed", "32", "--batchsize", "10", "--atype", "f32", "--seed", "3"]
    common = base + ["--cnn", "--model", "synthetic:1", "--imagedir", imgdir]
    raw = str(tmp_path / "raw.npz")
    assert cli.main(common + ["--extfeatures", "--savefile", raw]) == 0
    table = fmt.load_features(raw)
    assert len(table) == 40
    fn = str(tmp_path / "featsn.npz")
    fmt.save_features(fn, {k: (v / v.sum()).astype(np.float32) for k, v in table.items()})   # the reference's "featsn" 
    capsys.readouterr()
    assert cli.main(base + ["--train", "--epochs", "3", "--lr", "0.01", "--dropout", "0.0", "--features", fn, fn]) == 0
    a = _epoch_losses(capsys.readouterr().out)
    ck = str(tmp_path / "e2e.npz")
    assert cli.main(common + ["--train", "--epochs", "3", "--lr", "0.01", "--dropout", "0.0", "--savefile", ck]) == 0
    b = _epoch_losses(capsys.readouterr().out)
    assert len(a) == len(b) == 3 and len(a[0]) == 2
    np.testing.assert_allclose(a, b, rtol=2e-3)
    assert b[-1][0] < b[0][0]
    model, vocab, adam, _ = fmt.load_checkpoint(ck)
    assert adam["step"] > 0 and adam["step"] % 3 == 0 and model[6].shape == (len(vocab), 32)   # 3 epochs of whole batches (the tail rule may drop one)


def test_cli_two_ranks_on_one_gpu_equals_one_rank(tmp_path):
    # `tools/lrcn.py --train --gpus 2`: the launcher (a parent that never touches the GPU), two ranks, every batch split by rows, global
    # normaliser, gradients summed, identical Adam, rank 0 writes the checkpoint.  A box has ONE GPU and RCCL refuses two ranks per device:
    # LRCN_CLI_FAKE_MULTI=1 puts both ranks on device 0 over gloo -- the real control flow on the real kernels, only the transport swapped.
    import subprocess
    tr, fp, _ = _scene_dataset(tmp_path)
    cli = os.path.join(os.path.dirname(HERE), "tools", "lrcn.py")
    common = ["--coco", "--datafiles", tr, tr, "--features", fp, fp, "--hidden", "64", "64", "--embed", "64", "--batchsize", "10", "--atype", "f32",
              "--seed", "3", "--train", "--epochs", "2", "--lr", "0.01", "--dropout", "0.0"]
    outs = {}
    for gpus in (1, 2):
        ck = str(tmp_path / ("m%d.npz" % gpus))
        env = dict(os.environ, LRCN_CLI_FAKE_MULTI="1", LRCN_CLI_WATCHDOG_S="600")
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            env.pop(k, None)
        r = subprocess.run([sys.executable, cli] + common + ["--gpus", str(gpus), "--savefile", ck], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stderr[-3000:]
        outs[gpus] = (_epoch_losses(r.stdout), fmt.load_checkpoint(ck))
    l1, (m1, v1, a1, _) = outs[1]
    l2, (m2, v2, a2, _) = outs[2]
    assert len(l1) == len(l2) == 2 and v1 == v2 and a1["step"] == a2["step"]
    np.testing.assert_allclose(l1, l2, rtol=1e-5)
    for a, b in zip(m1, m2):
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-5)
