"""CPU: the host-side "next" rows of SURVEY.md 8(f): BLEU port against the reference script's outputs, tokenizer /
vocabulary / equal-length minibatcher properties, checkpoint and feature formats, crop geometry, CLI flag surface."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

import lrcn_amd  # noqa: F401
from lrcn_amd import bleu, captions as cap, formats as fmt

HERE = os.path.dirname(os.path.abspath(__file__))
BLEU_DIR = os.path.join(HERE, "golden", "bleu")
REF_EVAL = "/root/reference/eval"


def lines(path):
    with open(path) as f:
        return [ln.rstrip("\n") for ln in f]


def test_bleu_matches_reference_script_on_committed_fixture():
    hyp = lines(os.path.join(BLEU_DIR, "cand200.txt"))
    refs = [lines(os.path.join(BLEU_DIR, "ref200_%d" % k)) for k in range(5)]
    got = bleu.format_bleu(bleu.multi_bleu(hyp, refs))
    assert got == lines(os.path.join(BLEU_DIR, "expected200.txt"))[0]


@pytest.mark.skipif(not os.path.isdir(REF_EVAL), reason="full known-answer files live in the reference tree (build container only)")
def test_bleu_four_known_answers_of_the_survey():
    exp = lines(os.path.join(BLEU_DIR, "expected_full.txt"))
    cases = [("candidates.txt", "coco_refs/ref"), ("caps_flickr_bm3", "flickr_refs/f_ref"), ("caps_flickr_bm5", "flickr_refs/f_ref"),
             ("caps_flickr_bm10", "flickr_refs/f_ref")]
    for (hyp, stem), want in zip(cases, exp):
        got = bleu.format_bleu(bleu.multi_bleu(lines(os.path.join(REF_EVAL, hyp)), bleu.read_refs(os.path.join(REF_EVAL, stem))))
        assert got == want
    assert exp[0].startswith("BLEU = 68.2/47.1/33.0/23.5") and exp[2].startswith("BLEU = 60.9/39.4/26.0/17.6")


def test_bleu_edge_cases():
    r = bleu.multi_bleu(["a b c d", ""], [["a b c d", "x"], ["a b", "y z"]])
    assert r["hyp_len"] == 4 and r["ref_len"] == 5  # closest length: 4 for line 1; tie-free 1 for the empty line
    assert abs(r["bleu"][0] - 100.0) < 1e-9 and abs(r["bleu"][3] - 100.0) < 1e-9
    assert bleu.multi_bleu([], [[]])["ref_len"] == 0
    # clipping: hypothesis repeats a word more often than any reference
    r = bleu.multi_bleu(["the the the the"], [["the cat"], ["the the dog"]])
    assert abs(r["bleu"][0] - 50.0) < 1e-9


FLICKR = ["%d.jpg#%d\\t%s" % (1000 + i // 5, i % 5, s) for i, s in enumerate(
    ["A man , in a Blue shirt.", "Two dogs run !", "a man", "The man's dog (big) runs .", "dogs run",
     "A cat sits on a mat .", "a cat", "the cat sits", "A CAT !", "cat on mat",
     "Two men play ball .", "men play", "two men", "a ball", "Men play ball outside today"])]
FLICKR = [s.replace("\\t", "\t") for s in FLICKR]


def test_flickr_tokenizer_and_sort():
    i, w = cap.tokenize_flickr_line(FLICKR[0])
    assert i == 1000 and w == ["a", "man", "in", "a", "blue", "shirt"]
    assert cap.tokenize_flickr_line(FLICKR[3])[1] == ["the", "man's", "dog", "big", "runs"]  # inner apostrophe survives the strip
    caps = cap.tokenize_flickr(FLICKR)
    assert [c[1] for c in caps] == sorted(c[1] for c in caps)
    same = [c[0][0] for c in caps if c[1] == 2]
    assert same == [1000, 1000, 1001, 1001, 1002, 1002, 1002]  # stable: file order inside a length group


def test_coco_tokenizer_vocab_and_specials():
    js = json.dumps({"annotations": [{"image_id": 7, "caption": "A  dog runs."}, {"image_id": 8, "caption": "a dog ( sleeps ) !"},
                                     {"image_id": 9, "caption": "A dog"}]})
    caps = cap.tokenize_coco(js)
    assert [c[0] for c in caps] == [(9, ["a", "dog"]), (7, ["a", "dog", "runs"]), (8, ["a", "dog", "sleeps"])]
    v = cap.build_vocab([caps], threshold=2)
    assert v == {"~~": 1, "``": 2, "##": 3, "a": 4, "dog": 5}
    assert cap.index_to_word(v) == ["~~", "``", "##", "a", "dog"]
    assert cap.build_vocab([caps, caps], threshold=2)["runs"] == 6  # counts accumulate over lists (val/test included)


def test_minibatcher_contract():
    rng = np.random.default_rng(0)
    caps = sorted([((i, ["w%d" % rng.integers(0, 30) for _ in range(rng.integers(1, 7))]), 0) for i in range(500)], key=lambda t: len(t[0][1]))
    caps = [((i, w), len(w)) for (i, w), _ in caps]
    vocab = cap.build_vocab([caps], threshold=1)
    kept = cap.delete_unbatchable_captions(caps, 25)
    for n in set(c[1] for c in kept):
        assert sum(1 for c in kept if c[1] == n) % 25 == 0
    kept = cap.delete_unbatchable_captions(caps, 10)  # minibatch() below forces batch 10 on a split this small
    seq, ids, lengths, bs = cap.minibatch(caps, vocab, 25)
    assert bs == 10  # <= 30000 captions: forced to 10 (lrcn.jl:260-270)
    assert len(lengths) % 10 == 0 and len(ids) == len(lengths) // 10 and len(seq) == sum(lengths[::10])
    total = 0
    for b, (bid, toks) in enumerate(cap.batches(seq, ids, lengths, bs)):
        T, B = toks.shape
        assert B == 10 and toks.dtype == np.int32 and toks.min() >= 0 and toks.max() < len(vocab)
        words = {i: w for (i, w), _ in kept}
        assert [vocab[w] - 1 for w in words[bid[3]]] == list(toks[:, 3])  # column b of the block = that caption, 0-based
        total += T
    assert total == sum(lengths[::10])
    # unknown words map to unk; long captions are skipped by batches()
    long_caps = [((1, ["zz"] * 30), 30)] * 10
    s2, i2, l2, b2 = cap.minibatch(long_caps, vocab, 10)
    assert all(t == cap.UNK for row in s2 for t in row) and list(cap.batches(s2, i2, l2, b2)) == []
    assert cap.caption_text([1, 5, 6, 0, 9], ["~~", "``", "##", "x", "y", "a", "b"]) == "a b ."


def _caps_of(lengths):
    return [((i, ["w"] * n), n) for i, n in enumerate(lengths)]


def _julia_trace(lengths, B):
    """delete_unbatchable_captions! (lrcn.jl:299-327) written out with Julia's own 1-based variables, as a second, independent
    transcription for the test to compare against (kept deliberately literal; `findfirst` returns 0 when absent in Julia 0.5/0.6)."""
    L = [None] + list(lengths)          # L[1..n]
    n = len(lengths)
    limit = n - B + 1
    max_length = max(lengths)
    current_length, current_index = L[1], 1
    ranges = []
    guard = 0
    while current_index < limit:
        guard += 1
        assert guard < 10 * n + 10, "the reference would spin here"
        if L[current_index + B - 1] == current_length:
            current_index += B
        else:
            old_index = current_index
            current_index = 0
            while current_index == 0:
                current_length += 1
                if current_length > max_length:
                    break
                current_index = next((i for i in range(1, n + 1) if L[i] == current_length), 0)
            ranges += list(range(old_index, current_index))       # old_index:current_index-1
        if current_index >= limit:
            ranges += list(range(current_index, n + 1))           # current_index:length(lengths)
            break
    return [i for i in range(1, n + 1) if i not in set(ranges)]   # surviving 1-based positions


def test_delete_unbatchable_is_the_references_function():
    """f1: integer work, so the bar is the reference's exact result.  Expectations traced by hand through lrcn.jl:299-327
    (cursor / limit / ranges written in the comments), then cross-checked against a literal transcription on random inputs."""
    B = 10
    cases = [
        # limit = 11; cursor 1 -> 11 >= limit: delete 11:20 -- the final FULL batch goes (tail rule :320-323)
        ([3] * 20, 10, list(range(0, 10))),
        # limit = 31; 1 -> 11 -> 21 -> 31 >= limit at the boundary of the 3-group: delete 31:40, the whole 4-group
        ([3] * 30 + [4] * 10, 30, list(range(0, 30))),
        # limit = 33; 1 -> 11 -> 21, window 21..30 straddles: delete 21:25, cursor 26; 26 -> 36 >= 33: delete 36:42
        ([3] * 25 + [4] * 17, 30, list(range(0, 20)) + list(range(25, 35))),
        # limit = 52; 2-group (7) has no full window: delete 1:7, cursor 8 (first 3); 8 -> 18 -> 28, window 28..37 straddles:
        # delete 28:30, current_length 4 is ABSENT (findfirst = 0, loop again), 5 found at 31; 31 -> 41 -> 51 -> 61 >= 52: delete 61:61
        ([2] * 7 + [3] * 23 + [5] * 31, 50, list(range(7, 27)) + list(range(30, 60))),
        # a short FIRST group and a skipped length: delete 1:3, cursor 4 (length 6 after skipping 5); 4 -> 14 -> 24 >= limit 16... traced:
        # n = 25, limit = 16; cursor 1: L[10] = 6 != 4 -> delete 1:3, length 5 absent, 6 at 4; 4 -> 14; L[23] = 6 -> 24 >= 16: delete 24:25
        ([4] * 3 + [6] * 22, 20, list(range(3, 23))),
        # the group after a boundary is shorter than a batch and the last group is exactly one batch: it is deleted by the tail rule
        # n = 33, limit = 24; 1 -> 11; L[20] = 5 != 3 -> 4 found at 11 (empty range); L[20] = 5 != 4 -> delete 11:13, cursor 14;
        # 14 -> 24 >= 24: delete 24:33
        ([3] * 10 + [4] * 3 + [5] * 20, 20, list(range(0, 10)) + list(range(13, 23))),
        # exactly one batch: limit = 1, the loop body never runs, nothing is deleted
        ([7] * 10, 10, list(range(0, 10))),
    ]
    for lengths, n_keep, keep_idx in cases:
        got = cap.delete_unbatchable_captions(_caps_of(lengths), B)
        assert [c[0][0] for c in got] == keep_idx, (lengths, [c[0][0] for c in got])
        assert len(got) == n_keep
        assert [i - 1 for i in _julia_trace(lengths, B)] == keep_idx
    # the repo's non-reference variant keeps the final full batch
    assert len(cap.delete_unbatchable_captions(_caps_of([3] * 20), B, reference_tail=False)) == 20
    assert len(cap.delete_unbatchable_captions(_caps_of([3] * 30 + [4] * 10), B, reference_tail=False)) == 40
    # fewer captions than a batch: nothing can be batched (the reference deletes nothing and then indexes out of bounds, :283)
    assert cap.delete_unbatchable_captions(_caps_of([3] * 4), B) == []
    assert cap.delete_unbatchable_captions([], B) == []
    # random sorted inputs, several batch sizes: identical survivors to the literal transcription, all groups whole batches
    rng = np.random.default_rng(11)
    for trial in range(200):
        bs = int(rng.choice([2, 3, 5, 10, 25]))
        n = int(rng.integers(bs, 400))
        lengths = sorted(int(x) for x in rng.choice([1, 2, 3, 4, 6, 7, 9, 15, 28, 31], size=n))
        got = [c[0][0] + 1 for c in cap.delete_unbatchable_captions(_caps_of(lengths), bs)]
        assert got == _julia_trace(lengths, bs), (lengths, bs)
        if n > bs:  # (n == bs: the reference deletes nothing whatever the lengths are)
            kept_len = [lengths[i - 1] for i in got]
            assert all(kept_len.count(v) % bs == 0 for v in set(kept_len))
            # and it never keeps MORE than the per-length multiple; it differs from it only in the tail
            sane = [c[0][0] + 1 for c in cap.delete_unbatchable_captions(_caps_of(lengths), bs, reference_tail=False)]
            assert set(got) <= set(sane) and len(sane) - len(got) in (0, bs)
    with pytest.raises(ValueError):   # unsorted: the reference's scan would never end
        cap.delete_unbatchable_captions(_caps_of([5] * 12 + [3] * 12), B)


def test_minibatch_sizing_matches_reference():
    """lrcn.jl:276-281: nbatch = div(sum(lengths), batch_size) word vectors, one id vector per batch start."""
    caps = _caps_of([2] * 7 + [3] * 23 + [5] * 31)
    vocab = {"~~": 1, "``": 2, "##": 3, "w": 4}
    seq, ids, lengths, bs = cap.minibatch(caps, vocab, 25)
    assert bs == 10 and len(lengths) == 50
    assert len(seq) == sum(lengths) // bs == 2 * 3 + 3 * 5 and len(ids) == len(range(0, len(lengths), bs)) == 5
    assert [lengths[i] for i in range(0, 50, 10)] == [3, 3, 5, 5, 5]
    assert ids[0] == list(range(7, 17)) and ids[2] == list(range(30, 40))
    assert all(row == [4] * 10 for row in seq)
    seq2, ids2, lengths2, _ = cap.minibatch(caps, vocab, 25, reference_tail=False)
    assert len(lengths2) == 50  # here the tail the reference deletes is the single leftover caption either way


def test_flickr_split_by_ids_and_seed():
    train, val, test = cap.split_flickr(FLICKR, test_ids=[1002], val_ids=[1001])
    assert {c[0][0] for c in test} == {1002} and {c[0][0] for c in val} == {1001} and {c[0][0] for c in train} == {1000}
    a = cap.split_flickr(FLICKR, val_size=1, test_size=1, seed=5)
    b = cap.split_flickr(FLICKR, val_size=1, test_size=1, seed=5)
    assert a == b and len(a[0]) == len(a[1]) == len(a[2]) == 5


def test_checkpoint_and_feature_round_trip(tmp_path):
    rng = np.random.default_rng(1)
    shapes = [(6, 8), (1, 8), (8, 8), (1, 8), (2, 2), (4096, 2), (11, 4), (2, 11), (1, 11)]
    model = [rng.standard_normal(s).astype(np.float32) for s in shapes]
    vocab = {"~~": 1, "``": 2, "##": 3, "dog": 4}
    adam = {"m": [m * 0.1 for m in model], "v": [m * m for m in model], "step": 17}
    p = str(tmp_path / "m.npz")
    fmt.save_checkpoint(p, model, vocab, adam=adam, meta={"epoch": 3})
    m2, v2, a2, meta = fmt.load_checkpoint(p)
    assert v2 == vocab and meta == {"epoch": 3} and a2["step"] == 17
    for x, y in zip(model + adam["m"] + adam["v"], m2 + a2["m"] + a2["v"]):
        np.testing.assert_array_equal(x, y)
    fmt.save_checkpoint(p, model, vocab)
    assert fmt.load_checkpoint(p)[2] is None  # the reference's payload: no optimizer state
    # a checkpoint is plain data: no object arrays inside (np.load(allow_pickle=False) reads every member), unicode words survive
    z = np.load(p, allow_pickle=False)
    assert all(z[k].dtype != object for k in z.files)
    uni = {"~~": 1, "caf\u00e9": 2, "\u72ac": 3}
    fmt.save_checkpoint(p, model, uni)
    assert fmt.load_checkpoint(p)[1] == uni
    # and a file that does carry a pickled object array (the round-1 layout) is refused, never unpickled
    legacy = str(tmp_path / "legacy.npz")
    d = {"param_%d_%s" % (i, n): a for i, (n, a) in enumerate(zip(fmt.PARAM_NAMES, model))}
    np.savez(legacy, vocab_words=np.array(list(vocab), dtype=object), vocab_ids=np.arange(4), meta=np.array("{}"), **d)
    with pytest.raises(ValueError):
        fmt.load_checkpoint(legacy)
    feats = {42: np.arange(4096, dtype=np.float32) + 1, 7: np.ones(4096, np.float32)}
    fp = str(tmp_path / "f.npz")
    fmt.save_features(fp, feats)
    back = fmt.load_features(fp)
    np.testing.assert_array_equal(back[42], feats[42])
    assert abs(fmt.load_features(fp, normalize=True)[7].sum() - 1.0) < 1e-5


def test_karpathy_and_matconvnet_readers(tmp_path):
    from scipy.io import savemat
    rng = np.random.default_rng(2)
    f = rng.standard_normal((4096, 3)).astype(np.float32)
    savemat(str(tmp_path / "vgg_feats.mat"), {"feats": f})
    with open(tmp_path / "dataset.json", "w") as fh:
        json.dump({"images": [{"imgid": 2, "filename": "77.jpg"}, {"imgid": 0, "filename": "5.jpg"}]}, fh)
    d = fmt.karpathy_features(str(tmp_path / "dataset.json"), str(tmp_path / "vgg_feats.mat"))
    np.testing.assert_array_equal(d[77], f[:, 2])
    np.testing.assert_array_equal(d[5], f[:, 0])
    # a MatConvNet-shaped file with tiny layers: 13 conv (+relu/pool) + fc6 + relu6 + fc7 + relu7 + fc8
    names = ["conv1_1", "relu1_1", "conv1_2", "relu1_2", "pool1", "conv2_1", "relu2_1", "conv2_2", "relu2_2", "pool2", "conv3_1", "relu3_1",
             "conv3_2", "relu3_2", "conv3_3", "relu3_3", "pool3", "conv4_1", "relu4_1", "conv4_2", "relu4_2", "conv4_3", "relu4_3", "pool4",
             "conv5_1", "relu5_1", "conv5_2", "relu5_2", "conv5_3", "relu5_3", "pool5", "fc6", "relu6", "fc7", "relu7", "fc8"]
    layers = np.empty((1, len(names)), dtype=object)
    for k, n in enumerate(names):
        L = {"name": n, "type": "x"}
        if n.startswith("conv"):
            L["weights"] = np.array([rng.standard_normal((3, 3, 2, 4)).astype(np.float32), rng.standard_normal((1, 4)).astype(np.float32)], dtype=object)
        elif n.startswith("fc"):
            L["weights"] = np.array([rng.standard_normal((2, 2, 3, 5)).astype(np.float32), rng.standard_normal((1, 5)).astype(np.float32)], dtype=object)
        layers[0, k] = L
    savemat(str(tmp_path / "vgg.mat"), {"layers": layers})
    cw, cb, fc6, fc7, mean = fmt.load_vgg_mat(str(tmp_path / "vgg.mat"))
    assert len(cw) == 13 and cw[0].shape == (3, 3, 2, 4) and cb[0].shape == (4,)
    assert fc6[0].shape == (5, 12) and fc7[0].shape == (5, 12)  # transpose(mat(w)): out x (w*h*c), stops at fc7 inclusive
    w6 = layers[0, names.index("fc6")]["weights"][0]
    assert fc6[0][3, 1 + 2 * (0 + 2 * 2)] == w6[1, 0, 2, 3]  # column-major flatten over (w, h, c)


def test_center_crop_geometry():
    img = np.zeros((300, 500, 3), np.uint8)
    img[:, 250:] = 200
    c = fmt.center_crop_224(img)
    assert c.shape == (224, 224, 3) and c.dtype == np.uint8
    assert c[:, :100].max() < 50 and c[:, 124:].min() > 150  # resized to 224 x 373, centre crop keeps the edge near the middle
    g = fmt.center_crop_224(np.full((224, 224), 7, np.uint8))
    assert g.shape == (224, 224, 3) and (g == 7).all()


def test_cli_flag_surface():
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    import importlib
    cli = importlib.import_module("lrcn")
    o = cli.build_parser().parse_args([])
    assert (o.hidden, o.embed, o.batchsize, o.epochs, o.beam_width, o.capnumber, o.generate) == ([1000, 1000], 1000, 25, 10, 3, 1000, 0)
    o = cli.build_parser().parse_args("--coco --train --datafiles a.json b.json --savefile m.npz --hidden 512 512 --seed 3 img.jpg".split())
    assert o.coco and o.train and o.datafiles == ["a.json", "b.json"] and o.hidden == [512, 512] and o.image == "img.jpg"
    out = subprocess.run([sys.executable, os.path.join(os.path.dirname(HERE), "tools", "lrcn.py"), "--help"], capture_output=True, text=True)
    for flag in ("--model", "--datafiles", "--loadfile", "--savefile", "--generate", "--hidden", "--embed", "--epochs", "--capnumber",
                 "--batchsize", "--lr", "--gclip", "--seed", "--atype", "--train", "--cnn", "--extfeatures", "--flickr", "--coco",
                 "--beam_width", "--bestfile"):
        assert flag in out.stdout


def _write_npy_like_the_julia_exporter(path, a):
    """Byte for byte what julia/export_jld_to_npy.jl's write_npy emits (that script cannot run here: no Julia): magic, version 1.0, uint16
    header length, the dict with fortran_order True and the shape as Julia prints it, space padding to a multiple of 64, newline, then the
    array's COLUMN-MAJOR memory image."""
    a = np.asarray(a)
    descr = {"float32": "<f4", "int64": "<i8"}[a.dtype.name]
    shape = ", ".join(str(n) for n in a.shape) + ("," if a.ndim == 1 else "")
    d = "{'descr': '%s', 'fortran_order': True, 'shape': (%s), }" % (descr, shape)
    pad = (64 - (10 + len(d) + 1) % 64) % 64
    header = d + " " * pad + "\n"
    with open(path, "wb") as f:
        f.write(b"\x93NUMPY\x01\x00")
        f.write(np.uint16(len(header)).tobytes())
        f.write(header.encode("latin1"))
        f.write(np.asfortranarray(a).tobytes(order="F"))


def test_reference_jld_export_directory_loads(tmp_path):
    # a trained reference model exported by julia/export_jld_to_npy.jl (JLD/HDF5 cannot be read in this image): the directory format
    # formats.load_npy_dir / load_feature_npy_dir read, written here exactly as the Julia script writes it
    from lrcn_amd import formats as fmt
    from lrcn_amd import lrcn as L
    rng = np.random.default_rng(0)
    E, H, V = 12, 8, 23
    model = [rng.standard_normal(tuple(sh)).astype(np.float32) for sh in L.param_shapes(E, H, H, V)]
    d = tmp_path / "model_npy"
    d.mkdir()
    for k, (n, a) in enumerate(zip(fmt.PARAM_NAMES, model)):
        _write_npy_like_the_julia_exporter(str(d / ("param_%d_%s.npy" % (k, n))), a)
    words = ["~~", "``", "##"] + ["w%d" % i for i in range(V - 3)]
    (d / "vocab.tsv").write_text("".join("%s\t%d\n" % (w, i + 1) for i, w in enumerate(words)), encoding="utf-8")
    got, vocab, adam, meta = fmt.load_checkpoint(str(d))       # a directory goes to load_npy_dir
    assert adam is None and "jld" in meta["source"]
    for a, b in zip(got, model):
        assert a.shape == b.shape and a.dtype == np.float32
        np.testing.assert_array_equal(a, b)
    assert vocab["~~"] == 1 and vocab["``"] == 2 and vocab["##"] == 3 and len(vocab) == V
    (d / "vocab.tsv").write_text("only\t1\n", encoding="utf-8")
    with pytest.raises(ValueError):
        fmt.load_npy_dir(str(d))
    # this repository's one-layer model (LRCN-1f: W1 takes [embedding | x_cnn | h], W2 / b2 / Wproj are empty) passes the shape check too
    # (the check demanded Wembed columns == rows(W1) - H1, the two-layer relation)
    d1 = tmp_path / "model_1f_npy"
    d1.mkdir()
    model1 = [rng.standard_normal(tuple(sh)).astype(np.float32) for sh in L.param_shapes(E, H, H, V, n_layers=1)]
    for k, (n, a) in enumerate(zip(fmt.PARAM_NAMES, model1)):
        _write_npy_like_the_julia_exporter(str(d1 / ("param_%d_%s.npy" % (k, n))), a)
    (d1 / "vocab.tsv").write_text("".join("%s\t%d\n" % (w, i + 1) for i, w in enumerate(words)), encoding="utf-8")
    got1 = fmt.load_npy_dir(str(d1))[0]
    assert [a.shape for a in got1] == [tuple(sh) for sh in L.param_shapes(E, H, H, V, n_layers=1)]
    # feature dictionary: ids + a 4096 x N matrix
    f = tmp_path / "feats_npy"
    f.mkdir()
    ids = np.array([7, 42, 100003], np.int64)
    m = rng.random((4096, 3)).astype(np.float32)
    _write_npy_like_the_julia_exporter(str(f / "feature_ids.npy"), ids)
    _write_npy_like_the_julia_exporter(str(f / "features.npy"), m)
    feats = fmt.load_features(str(f))
    assert sorted(feats) == [7, 42, 100003]
    np.testing.assert_array_equal(feats[42], m[:, 1])
    np.testing.assert_allclose(fmt.load_features(str(f), normalize=True)[7].sum(), 1.0, rtol=1e-5)
