"""CPU: liblrcn_cpu.so -- the C ABI of include/lrcn.h implemented on the host by the oracle (oracle/lrcn_cpu_abi.c,
test infrastructure).  The ABI's conventions (column-major arrays, 0-based [T][B] tokens, 9-slot models incl. LRCN-1f, error codes)
are exercised end to end against the golden vectors without a GPU, through the SAME ctypes signatures the product binding uses."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from lrcn_amd import _lib
from oracle import oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CPU_LIB = os.path.join(ROOT, "oracle", "liblrcn_cpu.so")


@pytest.fixture(scope="module")
def lib():
    orc.build()
    L = C.CDLL(CPU_LIB)
    for name, (res, args) in _lib.SIGNATURES.items():
        fn = getattr(L, name)   # AttributeError = a declared symbol the twin does not export
        fn.restype, fn.argtypes = res, args
    return L


def fptr(a):
    return a.ctypes.data_as(C.c_void_p)


def p9(arrs):
    return _lib.P9(*[a.ctypes.data_as(C.c_void_p).value if a.size else None for a in arrs])


def make_ctx(lib, E, H1, H2, V, max_B, max_T, n_layers=2, max_images=0):
    cfg = _lib.Config(0, E, H1, H2, V, max_B, max_T, _lib.LRCN_F32, _lib.LRCN_F32, max_images, n_layers)
    h = C.c_void_p()
    assert lib.lrcn_create(C.byref(cfg), C.byref(h)) == 0
    return h


def test_twin_exports_the_whole_header(lib):
    txt = re.sub(r"/\*.*?\*/", "", open(_lib.HEADER).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(lrcn_[a-z0-9_]+)\s*\(", txt)))
    assert sorted(_lib.SIGNATURES) == names
    assert lib.lrcn_version().startswith(b"lrcn-cpu")


@pytest.mark.parametrize("name", ["lstm_tiny", "lstm_tiny_drop", "lstm_ragged", "lstm1_tiny", "lstm1_drop"])
def test_golden_loss_grads_and_adam_through_the_abi(lib, golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    E, H1, H2, V = (int(z[k]) for k in ("E", "H1", "H2", "V"))
    nl = int(z["n_layers"]) if "n_layers" in z else 2
    T, B = z["tokens"].shape
    ctx = make_ctx(lib, E, H1, H2, V, B, T, nl)
    param = [orc.fa(z["p_" + n]) for n in orc.PARAM_NAMES]
    grads = [np.zeros_like(a) for a in param]
    feats, tokens = orc.fa(z["feats"]), np.ascontiguousarray(z["tokens"], np.int32)
    d = None
    keep = []
    if "mask1" in z:
        m1 = np.ascontiguousarray(np.stack([np.asfortranarray(b).ravel(order="F") for b in z["mask1"]]), np.float32)
        keep.append(m1)
        d = _lib.Dropout(0.0, 0, m1.ctypes.data, None)
        if "mask2" in z:
            m2 = np.ascontiguousarray(np.stack([np.asfortranarray(b).ravel(order="F") for b in z["mask2"]]), np.float32)
            keep.append(m2)
            d.mask2 = m2.ctypes.data
    out = C.c_double()
    assert lib.lrcn_loss_grad(ctx, p9(param), fptr(feats), fptr(tokens), T, B, int(z["norm_B"]), C.byref(d) if d else None, p9(grads),
                              C.byref(out)) == 0
    assert abs(out.value - float(z["loss"])) <= 1e-6 * abs(float(z["loss"]))
    for n, g in zip(orc.PARAM_NAMES, grads):
        if g.size:
            np.testing.assert_allclose(g, z["g_" + n], rtol=1e-4, atol=1e-7, err_msg=n)
    last = C.c_double()
    assert lib.lrcn_last_loss(ctx, C.byref(last)) == 0 and last.value == out.value
    if d is None:
        # the body of average_loss's batch loop under its own name (rev 5): pdrop 0, divided by the batch's OWN size (lrcn.jl:412, 452-475)
        avg, own = C.c_double(), C.c_double()
        assert lib.lrcn_avg_loss_batch(ctx, p9(param), fptr(feats), fptr(tokens), T, B, C.byref(avg)) == 0
        assert lib.lrcn_loss(ctx, p9(param), fptr(feats), fptr(tokens), T, B, B, None, C.byref(own)) == 0
        assert avg.value == own.value and abs(avg.value * B - float(z["loss"]) * int(z["norm_B"])) <= 1e-6 * abs(avg.value * B)
    # train1's loop body as one call, twice (the golden Adam trajectory)
    mom, var = [np.zeros_like(a) for a in param], [np.zeros_like(a) for a in param]
    for step, ref_loss in enumerate(z["adam_losses"], start=1):
        assert lib.lrcn_train_step(ctx, p9(param), p9(grads), p9(mom), p9(var), fptr(feats), fptr(tokens), T, B, int(z["norm_B"]),
                                   C.byref(d) if d else None, step, 1e-3, 0.9, 0.999, 1e-8, C.byref(out)) == 0
        assert abs(out.value - ref_loss) <= 2e-6 * abs(ref_loss)
    for n, a in zip(orc.PARAM_NAMES, param):
        if a.size:
            np.testing.assert_allclose(a, z["a_" + n], rtol=0, atol=2e-6, err_msg=n)
    lib.lrcn_destroy(ctx)


@pytest.mark.parametrize("name", ["lstm_tiny", "lstm1_tiny"])
def test_golden_beam_search_and_logits_through_the_abi(lib, golden_dir, name):
    z = np.load(os.path.join(golden_dir, name + ".npz"))
    E, H1, H2, V = (int(z[k]) for k in ("E", "H1", "H2", "V"))
    nl = int(z["n_layers"]) if "n_layers" in z else 2
    T, B = z["tokens"].shape
    K, nword = int(z["beam_K"]), int(z["beam_nword"])
    ctx = make_ctx(lib, E, H1, H2, V, max(B, 4 * K), T, nl)
    param = [orc.fa(z["p_" + n]) for n in orc.PARAM_NAMES]
    logits = np.zeros((T + 1, V, B), np.float32)
    feats, tokens = orc.fa(z["feats"]), np.ascontiguousarray(z["tokens"], np.int32)
    assert lib.lrcn_forward_logits(ctx, p9(param), fptr(feats), fptr(tokens), T, B, fptr(logits)) == 0
    np.testing.assert_allclose(np.transpose(logits, (0, 2, 1)), z["logits"], rtol=1e-5, atol=1e-6)
    n = len(z["beam_tokens"])
    out = (C.c_int32 * (n * (nword + 2)))()
    lens, probs = (C.c_int * n)(), (C.c_float * n)()
    assert lib.lrcn_beam_search_batch(ctx, p9(param), fptr(orc.fa(z["feats"][:n])), n, K, nword, out, lens, probs) == 0
    for i, (ref, rp) in enumerate(zip(z["beam_tokens"], z["beam_prob"])):
        assert list(out[i * (nword + 2):i * (nword + 2) + lens[i]]) == list(ref[ref >= 0])
        assert abs(probs[i] - rp) <= 1e-5 * abs(rp)
    lib.lrcn_destroy(ctx)


def test_image_front_end_and_operators_through_the_abi(lib, golden_dir):
    ctx = make_ctx(lib, 8, 8, 8, 17, 2, 1)
    rng = np.random.default_rng(2)
    ims = [rng.integers(0, 256, size=s, dtype=np.uint8) for s in [(300, 451, 3), (224, 500, 1), (97, 31, 4)]]
    flat = np.concatenate([a.reshape(-1) for a in ims])
    offs = np.cumsum([0] + [a.size for a in ims[:-1]]).astype(np.int64)
    N = len(ims)
    hs, ws, cs = ((C.c_int * N)(*[a.shape[k] for a in ims]) for k in range(3))
    out = np.zeros((N, 224, 224, 3), np.uint8)
    assert lib.lrcn_resize_crop_u8(ctx, fptr(flat), offs.ctypes.data_as(C.POINTER(C.c_int64)), hs, ws, cs, N, fptr(out)) == 0
    np.testing.assert_array_equal(out, orc.resize_crop_u8(ims))
    avg = (rng.random((224, 224, 3)) * 50 + 90).astype(np.float32)
    assert lib.lrcn_set_average_image(ctx, fptr(orc.fa(avg))) == 0
    pre = np.zeros((224, 224, 3, N), np.float32, order="F")
    assert lib.lrcn_preprocess_u8(ctx, fptr(out), N, None, fptr(pre)) == 0
    np.testing.assert_array_equal(pre, orc.preprocess_u8_avg(out, avg))
    z = np.load(os.path.join(golden_dir, "cnn_small.npz"))
    x, w, b = orc.fa(z["x"]), orc.fa(z["w"]), orc.fa(z["b"])
    yp = np.zeros(z["yp"].shape, np.float32, order="F")
    assert lib.lrcn_conv3x3(ctx, fptr(x), x.shape[0], x.shape[1], x.shape[2], x.shape[3], fptr(w), fptr(b), w.shape[3], 1, 1, fptr(yp)) == 0
    np.testing.assert_allclose(yp, z["yp"], rtol=1e-5, atol=1e-5)
    # the bf16 stack's first launch as a probe (lrcn_conv1_fused): on the twin = the oracle's operators composed
    img = rng.integers(0, 256, size=(2, 16, 16, 3), dtype=np.uint8)
    w11 = orc.fa((rng.standard_normal((3, 3, 3, 64)) * 0.3).astype(np.float32))
    w12 = orc.fa((rng.standard_normal((3, 3, 64, 64)) * 0.06).astype(np.float32))
    b11, b12 = (rng.standard_normal(64).astype(np.float32) for _ in range(2))
    mean = (C.c_float * 3)(104.0, 117.0, 123.0)
    y1 = np.zeros((8, 8, 64, 2), np.float32, order="F")
    assert lib.lrcn_conv1_fused(ctx, fptr(img), 2, 16, mean, fptr(w11), fptr(b11), fptr(w12), fptr(b12), fptr(y1)) == 0
    x1 = orc.preprocess_u8(img, np.array([104.0, 117.0, 123.0], np.float32))
    np.testing.assert_allclose(y1, orc.pool2(orc.conv3x3(orc.conv3x3(x1, w11, b11, relu=True), w12, b12, relu=True)), rtol=1e-5, atol=1e-4)
    assert lib.lrcn_conv1_fused(ctx, fptr(img), 2, 20, mean, fptr(w11), fptr(b11), fptr(w12), fptr(b12), fptr(y1)) != 0  # S % 16
    f = orc.fa(np.abs(rng.standard_normal((3, 4096))).astype(np.float32))
    ref = f / f.sum(axis=1, keepdims=True)
    assert lib.lrcn_normalize_features(ctx, fptr(f), 3) == 0
    np.testing.assert_allclose(f, ref, rtol=1e-5)
    lib.lrcn_destroy(ctx)


def test_error_codes_match_the_header(lib):
    ctx = make_ctx(lib, 8, 8, 8, 17, 4, 3)
    sz = (C.c_int64 * 9)()
    assert lib.lrcn_param_sizes_n(2, 8, 8, 8, 17, sz) == 0
    param = [np.zeros(int(n), np.float32) for n in sz]
    assert lib.lrcn_init_weights(ctx, p9(param), 5) == 0 and (param[1][:8] == 1).all() and (param[1][8:] == 0).all()
    feats = np.zeros((4, 4096), np.float32, order="F")
    out = C.c_double()
    bad = np.full((2, 4), 17, np.int32)   # id V: the forgotten 1-based -> 0-based shift
    assert lib.lrcn_loss(ctx, p9(param), fptr(feats), fptr(bad), 2, 4, 4, None, C.byref(out)) == -1   # LRCN_EINVAL
    assert b"token id" in lib.lrcn_last_error(ctx)
    ok = np.full((2, 4), 5, np.int32)
    assert lib.lrcn_loss(ctx, p9(param), fptr(feats), fptr(ok), 2, 4, 4, None, C.byref(out)) == 0 and np.isfinite(out.value)
    assert lib.lrcn_loss(ctx, p9(param), fptr(feats), fptr(ok), 2, 9, 9, None, C.byref(out)) == -1     # B > max_B
    assert lib.lrcn_vgg_calibrate(ctx, fptr(feats), 1, None, 1.25) != 0                                 # GPU-only piece refuses
    assert lib.lrcn_comm_init(ctx, 2, 0, fptr(feats)) == -4                                             # LRCN_ESTATE: no transport on the host
    lib.lrcn_destroy(ctx)
    cfg = _lib.Config(0, 8, 8, 7, 17, 4, 3, 0, 0, 0, 2)   # odd H2
    h = C.c_void_p()
    assert lib.lrcn_create(C.byref(cfg), C.byref(h)) == -1
