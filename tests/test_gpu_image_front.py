"""GPU: the image front end (lrcn.jl:190-221, 750-773) -- batched resize + centre crop + grey->RGB on the
device against a NumPy restatement (bit-exact: integer arithmetic), the full averageImage against the reference's own order
of operations restated in NumPy, feature normalisation, and the driver's --extfeatures / --cnn --generate paths on JPEGs."""
import importlib
import json
import os
import sys

import numpy as np
import pytest
import torch

import lrcn_amd
from lrcn_amd import formats as fmt
from lrcn_amd import lrcn as L
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def small_ctx(**kw):
    return L.Context(8, 8, 8, 17, max_B=2, max_T=1, **kw)


def test_resize_crop_kernel_bit_exact_vs_numpy():
    rng = np.random.default_rng(5)
    shapes = [(300, 451, 3), (451, 300, 3), (224, 224, 3), (224, 500, 1), (640, 480), (50, 70, 3), (97, 31, 4), (1000, 1500, 3), (225, 224, 3),
              (1, 1, 3), (2047, 223, 1)]
    ims = [rng.integers(0, 256, size=s, dtype=np.uint8) for s in shapes]
    ctx = small_ctx()
    got = L.resize_crop_u8(ctx, ims).cpu().numpy()
    ref = orc.resize_crop_u8(ims)
    assert got.shape == (len(ims), 224, 224, 3) and got.dtype == np.uint8
    np.testing.assert_array_equal(got, ref)
    np.testing.assert_array_equal(got[2], ims[2])                    # 224 x 224 passes through untouched
    assert (got[3][..., 0] == got[3][..., 1]).all() and (got[3][..., 1] == got[3][..., 2]).all()   # grey -> three equal channels (lrcn.jl:762-764)
    # geometry: a vertical edge at the centre column of a 300 x 500 image stays at the centre of the crop (div offsets, lrcn.jl:756-760)
    edge = np.zeros((300, 500, 3), np.uint8)
    edge[:, 250:] = 200
    c = L.resize_crop_u8(ctx, [edge]).cpu().numpy()[0]
    assert c[:, :110].max() == 0 and c[:, 114:].min() == 200
    # and the smooth-image agreement with the host utility (PIL bilinear; same geometry, different resampling kernel)
    yy, xx = np.mgrid[0:300, 0:451]
    sm = ((np.sin(yy / 40.0) + np.cos(xx / 55.0)) * 60 + 128).astype(np.uint8)
    a = L.resize_crop_u8(ctx, [sm]).cpu().numpy()[0, :, :, 0].astype(int)
    b = fmt.center_crop_224(sm)[:, :, 0].astype(int)
    assert np.abs(a - b).max() <= 2
    with pytest.raises(lrcn_amd.LrcnError):
        L.resize_crop_u8(ctx, [np.zeros((10, 10, 2), np.uint8)])   # 2 channels: not an image format the reference handles
    ctx.close()


def test_average_image_array_matches_reference_order_of_operations():
    # lrcn.jl:770-771: 255 e1 .- averageImage BEFORE the last permutedims -> pixel (row r, col q) meets averageImage(q, r)
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, size=(2, 224, 224, 3), dtype=np.uint8)
    avg = (rng.random((224, 224, 3)) * 40 + 100).astype(np.float32)
    ctx = small_ctx()
    L.set_average_image(ctx, avg)
    got = L.from_jl(L.read_image_data_u8(ctx, torch.as_tensor(img).cuda(), mean=None))
    ref = orc.preprocess_u8_avg(img, avg)
    np.testing.assert_array_equal(got, ref)
    assert got[5, 9, 1, 0] == np.float32(img[0, 5, 9, 1]) - avg[9, 5, 1]
    L.set_average_image(ctx, None)
    back = L.from_jl(L.read_image_data_u8(ctx, torch.as_tensor(img).cuda()))
    np.testing.assert_array_equal(back, orc.preprocess_u8(img, np.array(L.VGG_MEAN, np.float32)))
    with pytest.raises(lrcn_amd.LrcnError):
        L.read_image_data_u8(ctx, torch.as_tensor(img).cuda(), mean=None)   # neither means nor an averageImage
    ctx.close()


@pytest.mark.parametrize("dtype,tol", [(lrcn_amd.LRCN_F32, 1e-4), (lrcn_amd.LRCN_BF16, 3e-2)])
def test_full_vgg_with_average_image_vs_oracle(dtype, tol):
    w = L.synthetic_vgg_weights(seed=2, bias_std=0.05)
    host = ([L.from_jl(t) for t in w[0]], [t.cpu().numpy() for t in w[1]], (L.from_jl(w[2][0]), w[2][1].cpu().numpy()),
            (L.from_jl(w[3][0]), w[3][1].cpu().numpy()))
    rng = np.random.default_rng(81)
    img = rng.integers(0, 256, size=(2, 224, 224, 3), dtype=np.uint8)
    # far from constant and not symmetric: a ramp along dim 1 plus noise -- a transposed or per-channel mean gives other features
    avg = (40.0 + 0.8 * np.arange(224, dtype=np.float32)[:, None, None] + rng.random((224, 224, 3)).astype(np.float32) * 20).astype(np.float32)
    ref = orc.vgg_forward(host[0], host[1], host[2], host[3], orc.preprocess_u8_avg(img, avg))
    ctx = small_ctx(vgg_dtype=dtype, max_images=2)
    L.vgg_load(ctx, *w)
    L.set_average_image(ctx, avg)
    got = L.from_jl(L.convnet_u8(ctx, torch.as_tensor(img).cuda(), mean=None))
    assert np.abs(got - ref).max() <= tol * np.abs(ref).max()
    wrong = orc.vgg_forward(host[0], host[1], host[2], host[3], orc.preprocess_u8_avg(img, np.transpose(avg, (1, 0, 2))))
    assert np.abs(wrong - ref).max() > 3 * tol * np.abs(ref).max()  # the test can tell the two orientations apart
    # normalised features: input / sum(input)  (lrcn.jl:595-597)
    f = L.convnet_u8(ctx, torch.as_tensor(img).cuda(), mean=None, normalize=True)
    np.testing.assert_allclose(L.from_jl(f), got / got.sum(axis=1, keepdims=True), rtol=2e-5, atol=1e-7)
    ctx.close()


def _write_images(tmp_path, ids, rng):
    from PIL import Image
    d = tmp_path / "imgs"
    d.mkdir()
    sizes = [(240, 320), (320, 240), (224, 224), (500, 375), (230, 600)]
    for k, i in enumerate(ids):
        h, w = sizes[k % len(sizes)]
        yy, xx = np.mgrid[0:h, 0:w]
        base = (np.sin(yy / (7.0 + k)) * 50 + np.cos(xx / (5.0 + 2 * k)) * 50 + 128)
        arr = np.stack([base, base[::-1], base[:, ::-1]], axis=2).clip(0, 255).astype(np.uint8)
        im = Image.fromarray(arr if k % 3 else arr[:, :, 0])   # every third image is greyscale
        im.save(str(d / ("%d.jpg" % i)), quality=95)
    return str(d)


def test_cli_extfeatures_and_cnn_generate_on_jpegs(tmp_path, capsys):
    sys.path.insert(0, os.path.join(os.path.dirname(HERE), "tools"))
    cli = importlib.import_module("lrcn")
    rng = np.random.default_rng(0)
    ids = [11, 12, 13, 14, 15]
    imgdir = _write_images(tmp_path, ids, rng)
    anns = [{"image_id": i, "caption": "A dog runs ."} for i in ids for _ in range(5)]
    cf = str(tmp_path / "captions.json")
    with open(cf, "w") as fh:
        json.dump({"annotations": anns}, fh)
    out = str(tmp_path / "feats.npz")
    # --extfeatures (lrcn.jl:162-172, 190-221): decode -> GPU resize/crop -> VGG-16 -> fc7, batches of 2 and a last batch of 1
    assert cli.main(["--coco", "--cnn", "--model", "synthetic:4", "--extfeatures", "--datafiles", cf, "--imagedir", imgdir, "--savefile", out,
                     "--batchsize", "2", "--atype", "f32", "--hidden", "16", "16", "--embed", "16"]) == 0
    table = fmt.load_features(out)
    assert sorted(table) == ids and all(v.shape == (4096,) and np.isfinite(v).all() for v in table.values())
    # the same features from the pieces: PIL decode, NumPy restatement of the crop, CPU oracle VGG
    from PIL import Image
    w = L.synthetic_vgg_weights(seed=4, bias_std=0.05)
    host = ([L.from_jl(t) for t in w[0]], [t.cpu().numpy() for t in w[1]], (L.from_jl(w[2][0]), w[2][1].cpu().numpy()),
            (L.from_jl(w[3][0]), w[3][1].cpu().numpy()))
    pick = [ids[0], ids[3]]
    crops = orc.resize_crop_u8([np.asarray(Image.open(os.path.join(imgdir, "%d.jpg" % i))) for i in pick])
    ref = orc.vgg_forward(host[0], host[1], host[2], host[3], orc.preprocess_u8(crops, np.array(L.VGG_MEAN, np.float32)))
    for k, i in enumerate(pick):
        assert np.abs(table[i] - ref[k]).max() <= 1e-4 * np.abs(ref[k]).max(), i
    # --cnn --generate <image> (lrcn.jl:127-137, 585-642): one caption line "w1 w2 ... ."
    ck = str(tmp_path / "m.npz")
    assert cli.main(["--coco", "--datafiles", cf, "--savefile", ck, "--hidden", "16", "16", "--embed", "16", "--seed", "3"]) == 0
    capsys.readouterr()
    assert cli.main(["--cnn", "--model", "synthetic:4", "--loadfile", ck, "--generate", "6", "--beam_width", "3", "--atype", "f32",
                     "--hidden", "16", "16", "--embed", "16", os.path.join(imgdir, "12.jpg")]) == 0
    lines = [ln for ln in capsys.readouterr().out.splitlines() if ln.endswith(" .") or ln == "."]
    assert len(lines) == 1 and len(lines[0].split()) <= 8, lines
