"""GPU: the HIP VGG-16 path (implicit-GEMM 3x3 conv + bias + ReLU + fused 2x2 max-pool, fc6/fc7, preprocessing) through
the C ABI against the golden vectors and the CPU oracle.  fp32: exact-fp32 MFMA, tolerance 1e-4 relative to the
tensor's max; bf16: 3e-2 relative to the tensor's max (operands rounded to 8 bits of mantissa per layer)."""
import os

import numpy as np
import pytest
import torch

import lrcn_amd
from lrcn_amd import lrcn as L
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def rel_max_err(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def small_ctx(vgg_dtype=lrcn_amd.LRCN_F32, max_images=0):
    return L.Context(8, 8, 8, 17, max_B=2, max_T=1, vgg_dtype=vgg_dtype, max_images=max_images)


def test_conv_pool_vs_golden_fp32(golden_dir):
    z = np.load(os.path.join(golden_dir, "cnn_small.npz"))
    ctx = small_ctx()
    x, w, b = L.to_jl(z["x"]), L.to_jl(z["w"]), torch.as_tensor(z["b"]).cuda()
    y = L.from_jl(L.conv3x3(ctx, x, w, b, relu=True, pool=False))
    np.testing.assert_allclose(y, z["y"], rtol=1e-5, atol=1e-5)
    yp = L.from_jl(L.conv3x3(ctx, x, w, b, relu=True, pool=True))
    np.testing.assert_allclose(yp, z["yp"], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("glds", ["0", "force"])
@pytest.mark.parametrize("dtype,tol", [(lrcn_amd.LRCN_F32, 1e-5), (lrcn_amd.LRCN_BF16, 2e-2)])
@pytest.mark.parametrize("shape", [(28, 64, 128, 3), (14, 128, 64, 5), (8, 96, 40, 2), (2, 32, 32, 1), (16, 64, 320, 3),
                                   (12, 192, 64, 2)])
def test_conv_layers_vs_oracle(dtype, tol, shape, glds, monkeypatch):
    # glds: the bf16 direct-to-LDS kernel (gemm_glds.hip) forced on / off; both must agree with the oracle
    monkeypatch.setenv("LRCN_GLDS", glds)
    S, Cin, Cout, N = shape
    rng = np.random.default_rng(S * 1000 + Cin)
    x = rng.standard_normal((S, S, Cin, N)).astype(np.float32)
    w = (rng.standard_normal((3, 3, Cin, Cout)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    ctx = small_ctx(dtype)
    ref = orc.conv3x3(x, w, b, relu=True)
    got = L.from_jl(L.conv3x3(ctx, L.to_jl(x), L.to_jl(w), torch.as_tensor(b).cuda(), relu=True, pool=False))
    assert rel_max_err(got, ref) <= tol
    refp = orc.pool2(ref)
    gotp = L.from_jl(L.conv3x3(ctx, L.to_jl(x), L.to_jl(w), torch.as_tensor(b).cuda(), relu=True, pool=True))
    assert rel_max_err(gotp, refp) <= tol
    # no-ReLU path keeps negatives (fc7-style)
    ref_lin = orc.conv3x3(x, w, b, relu=False)
    got_lin = L.from_jl(L.conv3x3(ctx, L.to_jl(x), L.to_jl(w), torch.as_tensor(b).cuda(), relu=False, pool=False))
    assert (got_lin < 0).any() and rel_max_err(got_lin, ref_lin) <= tol


@pytest.mark.parametrize("shape", [(16, 64, 256, 2), (28, 64, 128, 3), (12, 128, 136, 3), (10, 64, 384, 5), (8, 192, 512, 9),
                                   (6, 64, 200, 8)])
def test_conv_phase_interleaved_kernel_vs_oracle(shape, monkeypatch):
    # gemm_8p.hip (256x256 / 256x128 tiles, 16x16x32 MFMA, two staggered wave groups) forced on even for small grids:
    # full / partial M tiles, N tails, both epilogues (swapped plain, fused pool); and the same layer with the kernel off.
    S, Cin, Cout, N = shape
    rng = np.random.default_rng(S * 77 + Cout)
    x = rng.standard_normal((S, S, Cin, N)).astype(np.float32)
    w = (rng.standard_normal((3, 3, Cin, Cout)) * np.sqrt(2.0 / (9 * Cin))).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    ref = orc.conv3x3(x, w, b, relu=True)
    refp = orc.pool2(ref)
    ref_lin = orc.conv3x3(x, w, b, relu=False)
    outs = {}
    for knob in ("force", "0"):
        monkeypatch.setenv("LRCN_8P", knob)
        ctx = small_ctx(lrcn_amd.LRCN_BF16)
        got = L.from_jl(L.conv3x3(ctx, L.to_jl(x), L.to_jl(w), torch.as_tensor(b).cuda(), relu=True, pool=False))
        gotp = L.from_jl(L.conv3x3(ctx, L.to_jl(x), L.to_jl(w), torch.as_tensor(b).cuda(), relu=True, pool=True))
        got_lin = L.from_jl(L.conv3x3(ctx, L.to_jl(x), L.to_jl(w), torch.as_tensor(b).cuda(), relu=False, pool=False))
        assert rel_max_err(got, ref) <= 2e-2 and rel_max_err(gotp, refp) <= 2e-2 and rel_max_err(got_lin, ref_lin) <= 2e-2
        outs[knob] = (got, gotp, got_lin)
        ctx.close()
    # same bf16 operands, f32 accumulation in a different order: the two kernels agree to bf16 output rounding
    for a, c in zip(outs["force"], outs["0"]):
        assert rel_max_err(a, c) <= 1e-2


@pytest.mark.parametrize("shape", [(16, 16, 64, 3), (32, 32, 128, 2), (48, 16, 64, 5), (16, 64, 192, 1), (32, 32, 64, 11)])
def test_conv64_halo_patch_kernel_vs_oracle(shape, monkeypatch):
    # conv64.hip (Cin = 64: 18x18 halo patch in LDS, weights register-resident, persistent tiles, 3 patch buffers):
    # image borders, tile borders inside an image, several 64-channel chunks, more tiles than one pass of the ring
    # (N = 11 -> 44 tiles), W != H; against the oracle and against the implicit-GEMM kernels on the same operands.
    W, H, Cout, N = shape
    rng = np.random.default_rng(W * 31 + H + Cout)
    x = rng.standard_normal((W, H, 64, N)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 64, Cout)) * np.sqrt(2.0 / (9 * 64))).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    ref = orc.conv3x3(x, w, b, relu=True)
    refp = orc.pool2(ref)
    ref_lin = orc.conv3x3(x, w, b, relu=False)
    outs = {}
    for knob in ("1", "0"):
        monkeypatch.setenv("LRCN_CONV64", knob)
        ctx = small_ctx(lrcn_amd.LRCN_BF16)
        got = L.from_jl(L.conv3x3(ctx, L.to_jl(x), L.to_jl(w), torch.as_tensor(b).cuda(), relu=True, pool=False))
        gotp = L.from_jl(L.conv3x3(ctx, L.to_jl(x), L.to_jl(w), torch.as_tensor(b).cuda(), relu=True, pool=True))
        got_lin = L.from_jl(L.conv3x3(ctx, L.to_jl(x), L.to_jl(w), torch.as_tensor(b).cuda(), relu=False, pool=False))
        assert rel_max_err(got, ref) <= 2e-2 and rel_max_err(gotp, refp) <= 2e-2 and rel_max_err(got_lin, ref_lin) <= 2e-2
        outs[knob] = (got, gotp, got_lin)
        ctx.close()
    for a, c in zip(outs["1"], outs["0"]):
        assert rel_max_err(a, c) <= 1e-2


def test_preprocess_u8_bit_exact():
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, size=(3, 224, 224, 3), dtype=np.uint8)
    ctx = small_ctx()
    got = L.from_jl(L.read_image_data_u8(ctx, torch.as_tensor(img).cuda()))
    ref = orc.preprocess_u8(img, np.array(L.VGG_MEAN, np.float32))
    np.testing.assert_array_equal(got, ref)


@pytest.fixture(scope="module")
def vgg_setup():
    w = L.synthetic_vgg_weights(seed=1)
    host = ([L.from_jl(t) for t in w[0]], [t.cpu().numpy() for t in w[1]], (L.from_jl(w[2][0]), w[2][1].cpu().numpy()),
            (L.from_jl(w[3][0]), w[3][1].cpu().numpy()))
    rng = np.random.default_rng(1234)
    img = rng.integers(0, 256, size=(2, 224, 224, 3), dtype=np.uint8)
    x = orc.preprocess_u8(img, np.array(L.VGG_MEAN, np.float32))
    ref = orc.vgg_forward(host[0], host[1], host[2], host[3], x)  # N x 4096
    return w, img, x, ref


def test_full_vgg_fp32_vs_oracle(vgg_setup):
    w, img, x, ref = vgg_setup
    ctx = small_ctx(lrcn_amd.LRCN_F32, max_images=2)
    L.vgg_load(ctx, *w)
    got = L.from_jl(L.convnet(ctx, L.to_jl(x)))
    assert got.shape == (2, 4096) and (got < 0).any()  # pre-ReLU fc7
    assert rel_max_err(got, ref) <= 1e-4
    got8 = L.from_jl(L.convnet_u8(ctx, torch.as_tensor(img).cuda()))
    np.testing.assert_allclose(got8, got, rtol=0, atol=1e-6 * np.abs(ref).max())
    # batch of one image gives the same row (M-edge tiles)
    got1 = L.from_jl(L.convnet_u8(ctx, torch.as_tensor(img[1:2]).cuda()))
    np.testing.assert_allclose(got1[0], got8[1], rtol=0, atol=1e-5 * np.abs(ref).max())


def test_full_vgg_bf16_vs_oracle(vgg_setup):
    w, img, x, ref = vgg_setup
    ctx = small_ctx(lrcn_amd.LRCN_BF16, max_images=2)
    L.vgg_load(ctx, *w)
    got = L.from_jl(L.convnet_u8(ctx, torch.as_tensor(img).cuda()))
    assert rel_max_err(got, ref) <= 3e-2
    cos = float((got * ref).sum() / (np.linalg.norm(got) * np.linalg.norm(ref)))
    assert cos > 0.999


def test_fused_conv1_1_conv1_2_matches_two_launch_path(vgg_setup, monkeypatch):
    # conv64.hip FUSE (conv1_1 computed inside the conv1_2 kernel from the raw uint8 window) against the two-launch path
    # (conv11.hip + conv64.hip) on the same crops: same bf16 operands, conv1_1 summed in a different K order.
    w, img, x, ref = vgg_setup
    rng = np.random.default_rng(99)
    img3 = np.concatenate([img, rng.integers(0, 256, size=(1, 224, 224, 3), dtype=np.uint8)])
    img3[2, :40] = 255  # saturated / dark bands: exercises ReLU cut-off and the image borders
    img3[2, -30:] = 0
    host = ([L.from_jl(t) for t in w[0]], [t.cpu().numpy() for t in w[1]], (L.from_jl(w[2][0]), w[2][1].cpu().numpy()),
            (L.from_jl(w[3][0]), w[3][1].cpu().numpy()))
    ref3 = orc.vgg_forward(host[0], host[1], host[2], host[3], orc.preprocess_u8(img3[2:3], np.array(L.VGG_MEAN, np.float32)))
    outs = {}
    for knob in ("1", "0"):
        monkeypatch.setenv("LRCN_FUSE11", knob)
        ctx = small_ctx(lrcn_amd.LRCN_BF16, max_images=3)
        L.vgg_load(ctx, *w)
        outs[knob] = L.from_jl(L.convnet_u8(ctx, torch.as_tensor(img3).cuda()))
        ctx.close()
        assert rel_max_err(outs[knob][:2], ref) <= 3e-2 and rel_max_err(outs[knob][2:], ref3) <= 3e-2
    assert rel_max_err(outs["1"], outs["0"]) <= 1.5e-2
