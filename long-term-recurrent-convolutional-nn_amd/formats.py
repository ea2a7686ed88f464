"""On-disk formats around the hot path -- host-only code.

  * model checkpoints: the reference saves `model` + `vocab` as JLD/HDF5 after every epoch (`lrcn.jl:228-231, 183-186`)
    and **never saves the optimizer** (Adam restarts cold on `--loadfile`, `:88-95`).  No HDF5 reader exists in this image,
    so the checkpoint here is a NumPy `.npz` with the same payload in the reference's array shapes (column-major
    semantics preserved by storing the `(rows, cols)` arrays as-is) plus what the reference forgot: Adam moments, step.
  * MatConvNet VGG-16 (`imagenet-vgg-verydeep-16.mat`, `lrcn.jl:110-113`): `load_vgg_mat` walks `layers` like
    `get_params_cnn` (`:697-721`) up to and including `fc7` and returns the arrays `lrcn_vgg_load` takes.
  * Karpathy features (`feature_extractor.jl:13-50`): `dataset.json` + `vgg_feats.mat` (`feats` 4096 x N, column
    `imgid + 1`) -> {integer filename stem: float32[4096]}; the feature dictionary itself is stored as `.npz`.
  * a trained REFERENCE model / feature dictionary: `julia/export_jld_to_npy.jl` (run where Julia + JLD exist) writes the `.jld`
    payloads as plain `.npy` files by hand; `load_npy_dir` / `load_feature_npy_dir` (and load_checkpoint / load_features given a
    directory) read them.
  * crops: `center_crop_224` restates `read_image_data`'s geometry (`lrcn.jl:755-765`): resize so that the shorter side is
    224 (integer `div`), centre crop, grey -> 3 channels; returns the uint8 `[row][col][3]` crop that
    `lrcn_vgg_forward_u8` consumes (the float arithmetic `255 x - averageImage` and the H/W swap happen on the GPU).
"""
import json
import os

import numpy as np

PARAM_NAMES = ["W1", "b1", "W2", "b2", "Wproj", "Wcnn", "Wembed", "Wout", "bout"]  # initweights order (lrcn.jl:489-510)


def save_checkpoint(path, model, vocab, adam=None, meta=None):
    """model: 9 arrays in reference shapes (host, float32); vocab: word -> 1-based id; adam: optional
    dict(m=[9 arrays], v=[9 arrays], step=int)."""
    d = {"param_%d_%s" % (i, n): np.asarray(a, dtype=np.float32) for i, (n, a) in enumerate(zip(PARAM_NAMES, model))}
    # the words as ONE JSON string (like `meta`): no dtype=object array, so loading never needs pickle
    d["vocab_json"] = np.array(json.dumps(sorted(vocab, key=vocab.get), ensure_ascii=False))
    d["vocab_ids"] = np.array([vocab[w] for w in sorted(vocab, key=vocab.get)], dtype=np.int64)
    if adam is not None:
        for i, (m, v) in enumerate(zip(adam["m"], adam["v"])):
            d["adam_m_%d" % i] = np.asarray(m, dtype=np.float32)
            d["adam_v_%d" % i] = np.asarray(v, dtype=np.float32)
        d["adam_step"] = np.int64(adam["step"])
    d["meta"] = np.array(json.dumps(meta or {}))
    tmp = path + ".tmp.npz"
    np.savez(tmp, **d)
    os.replace(tmp, path)


def load_checkpoint(path):
    """-> (model list of 9 float32 arrays, vocab dict, adam dict or None, meta dict)."""
    if os.path.isdir(path):   # a reference .jld exported by julia/export_jld_to_npy.jl
        return load_npy_dir(path)
    z = np.load(path, allow_pickle=False)  # a checkpoint is data: nothing in it may execute on load
    model = [z["param_%d_%s" % (i, n)] for i, n in enumerate(PARAM_NAMES)]
    if "vocab_json" in z.files:
        words = json.loads(str(z["vocab_json"]))
    else:  # round-1 files stored the words as a pickled object array
        raise ValueError("%s holds a pickled vocabulary (pre-round-2 checkpoint); re-save it with save_checkpoint -- "
                         "pickled checkpoints are not loaded" % path)
    vocab = {str(w): int(i) for w, i in zip(words, z["vocab_ids"])}
    adam = None
    if "adam_step" in z.files:
        adam = {"m": [z["adam_m_%d" % i] for i in range(9)], "v": [z["adam_v_%d" % i] for i in range(9)],
                "step": int(z["adam_step"])}
    return model, vocab, adam, json.loads(str(z["meta"]))


def load_npy_dir(path):
    """A trained REFERENCE model exported from its `.jld` file by `julia/export_jld_to_npy.jl` (the reference saves `model` + `vocab` as
    JLD/HDF5, lrcn.jl:183-186, 228-231, and this image has no HDF5 reader): a directory with `param_<k>_<name>.npy` -- the nine arrays
    of `initweights` (lrcn.jl:489-510) in Julia's column-major order, `fortran_order: True`, so NumPy sees the reference's shapes --
    and `vocab.tsv` (word TAB 1-based id, lrcn.jl:248-255).  -> (model list of 9 float32 arrays, vocab dict, None, meta dict), the
    tuple load_checkpoint returns (no optimizer state: the reference never saved any)."""
    model = []
    for i, n in enumerate(PARAM_NAMES):
        a = np.load(os.path.join(path, "param_%d_%s.npy" % (i, n)), allow_pickle=False)
        if a.dtype != np.float32 or a.ndim != 2:
            raise ValueError("%s: expected a 2-D float32 array, got %s %s" % (n, a.dtype, a.shape))
        model.append(a)
    E_H1, H1x4 = model[0].shape
    x1 = E_H1 - H1x4 // 4   # LSTM-1's input width: E for the reference's two-layer model, E + h for LRCN-1f (the embedding and x_cnn side by side)
    one_layer = model[2].size == 0 and model[4].size == 0
    if model[1].shape != (1, H1x4) or model[6].shape[1] != (x1 - model[5].shape[1] if one_layer else x1) or model[7].shape[1] != model[6].shape[0] or \
            model[8].shape != (1, model[6].shape[0]):
        raise ValueError("the nine arrays do not have initweights' shapes (lrcn.jl:489-510): %s" % [m.shape for m in model])
    vocab = {}
    with open(os.path.join(path, "vocab.tsv"), encoding="utf-8") as f:
        for line in f:
            line = line.rstrip("\n")
            if not line:
                continue
            w, i = line.rsplit("\t", 1)
            vocab[w] = int(i)
    if len(vocab) != model[6].shape[0]:
        raise ValueError("vocab.tsv holds %d words, Wembed has %d rows" % (len(vocab), model[6].shape[0]))
    return model, vocab, None, {"source": "exported from a reference .jld by julia/export_jld_to_npy.jl"}


def load_feature_npy_dir(path, normalize=False):
    """A reference feature dictionary (`Dict{Int,Array{Float32}}`, lrcn.jl:206-207, 220; feature_extractor.jl:50) exported by
    `julia/export_jld_to_npy.jl feats`: `feature_ids.npy` (int64) + `features.npy` (4096 x N).  -> {image id: float32[4096]}."""
    ids = np.load(os.path.join(path, "feature_ids.npy"), allow_pickle=False)
    m = np.load(os.path.join(path, "features.npy"), allow_pickle=False)
    if m.ndim != 2 or m.shape[0] != 4096 or m.shape[1] != ids.shape[0]:
        raise ValueError("features.npy must be 4096 x %d, got %s" % (ids.shape[0], m.shape))
    out = {}
    for j, i in enumerate(ids):
        f = np.ascontiguousarray(m[:, j], dtype=np.float32)
        out[int(i)] = f / f.sum() if normalize else f
    return out


def save_features(path, feats):
    """feats: {image id: float32[4096]} (the reference's Dict{Int,Array{Float32}} saved as feats.jld)."""
    ids = np.array(sorted(feats), dtype=np.int64)
    np.savez(path, ids=ids, feats=np.stack([np.asarray(feats[i], dtype=np.float32).reshape(-1) for i in ids]))


def load_features(path, normalize=False):
    """-> {image id: float32[4096]}.  normalize=True divides each vector by its sum, which is what the reference's
    `featsn` files hold and what `generate` does for fresh images (`lrcn.jl:597`)."""
    if os.path.isdir(path):   # a reference feature dictionary exported by julia/export_jld_to_npy.jl
        return load_feature_npy_dir(path, normalize)
    z = np.load(path)
    out = {}
    for i, f in zip(z["ids"], z["feats"]):
        out[int(i)] = f / f.sum() if normalize else f
    return out


def karpathy_features(dataset_json_path, vgg_feats_mat_path):
    """feature_extractor.jl:13-34: image k of dataset.json -> column imgid of `feats`, keyed by the integer filename stem."""
    from scipy.io import loadmat
    with open(dataset_json_path) as f:
        images = json.load(f)["images"]
    feats = loadmat(vgg_feats_mat_path)["feats"]  # 4096 x N
    out = {}
    for im in images:
        key = int(im["filename"].split(".")[0])
        out.setdefault(key, np.ascontiguousarray(feats[:, im["imgid"]], dtype=np.float32))
    return out


def load_vgg_mat(path, last_layer="fc7"):
    """MatConvNet model -> (conv_w[13] (3,3,Cin,Cout), conv_b[13], (fc6_w (4096,25088), fc6_b), (fc7_w (4096,4096), fc7_b),
    average RGB).  `load_vgg_mat.average_image` holds, after the call, the full (224,224,3) averageImage when the file stores
    one (lrcn.jl:113 uses the array as it is; lrcn_set_average_image) and None when it stores only three channel means.  Follows get_params_cnn (lrcn.jl:697-721): conv weights as stored, fc weight = transpose(mat(w)), i.e.
    the (7,7,512,4096) array flattened column-major over (w,h,c) then transposed; stops after `last_layer` inclusive."""
    from scipy.io import loadmat
    m = loadmat(path, squeeze_me=False, struct_as_record=False)
    layers = m["layers"].ravel()
    conv_w, conv_b, fcs = [], [], []
    for cell in layers:
        L = cell[0, 0] if isinstance(cell, np.ndarray) else cell
        name = str(np.ravel(L.name)[0])
        w = getattr(L, "weights", None)
        if w is not None and np.size(w):
            w0, w1 = w.ravel()[0], w.ravel()[1]
            if name.startswith("conv"):
                conv_w.append(np.asarray(w0, dtype=np.float32))
                conv_b.append(np.asarray(w1, dtype=np.float32).reshape(-1))
            elif name.startswith("fc"):
                w0 = np.asarray(w0, dtype=np.float32)
                flat = w0.reshape((-1, w0.shape[-1]), order="F")  # mat(w): (w*h*c) x out, column-major flatten
                fcs.append((np.ascontiguousarray(flat.T), np.asarray(w1, dtype=np.float32).reshape(-1)))
        if name.startswith(last_layer):
            break
    mean = None
    load_vgg_mat.average_image = None
    avg = None
    try:
        avg = np.asarray(m["meta"][0, 0].normalization[0, 0].averageImage, dtype=np.float32)
    except Exception:  # older files keep it under "normalization"
        try:
            avg = np.asarray(m["normalization"][0, 0].averageImage, dtype=np.float32)
        except Exception:
            pass
    if avg is not None:
        if avg.size == 3:
            mean = avg.reshape(-1)
        elif avg.ndim == 3 and avg.shape[2] == 3:
            mean = avg.mean(axis=(0, 1))
            if avg.shape[:2] == (224, 224):
                load_vgg_mat.average_image = avg
    if len(conv_w) != 13 or len(fcs) != 2:
        raise ValueError("expected 13 conv + 2 fc weighted layers up to %s, found %d + %d" % (last_layer, len(conv_w), len(fcs)))
    return conv_w, conv_b, fcs[0], fcs[1], mean


def center_crop_224(img):
    """read_image_data's geometry (lrcn.jl:755-765) on a decoded image (PIL.Image or HxW[x3] uint8 array): resize so the
    shorter side is 224 with the other side div(side * 224, shorter), centre crop with div offsets, grey -> 3 channels.
    -> uint8 [224][224][3] (row, col, channel).  Resampling is PIL's bilinear (Images.imresize's kernel is not pinned).
    Host-side utility only: the driver (tools/lrcn.py) crops on the GPU with lrcn_resize_crop_u8 (lrcn.resize_crop_u8)."""
    from PIL import Image
    if not isinstance(img, Image.Image):
        img = Image.fromarray(np.asarray(img))
    img = img.convert("RGB")
    w, h = img.size  # PIL: (width, height); the reference's size(a0) = (rows, cols) = (h, w)
    s = min(h, w)
    nh, nw = (h * 224) // s, (w * 224) // s
    img = img.resize((nw, nh), Image.BILINEAR)
    i1, j1 = (nh - 224) // 2, (nw - 224) // 2
    a = np.asarray(img, dtype=np.uint8)[i1:i1 + 224, j1:j1 + 224, :3]
    return np.ascontiguousarray(a)
