"""GPU: the kernel instantiations bench.py actually launches at B = 256 against the CPU oracle.

The default dispatch picks tiles by grid size, so the small-N parity tests elsewhere run other instantiations than the
benchmark does.  Here every VGG layer runs at its REAL geometry, at a batch large enough that the default route is the one
taken at 256 images (asserted through lrcn_debug_route), with non-zero biases; the oracle checks two sampled images per
layer (convolution is per-image).  Then the whole bf16 stack at N = 256 and N = 32 with random conv/fc biases, and the
register-resident softmax+top-K instantiations <8> and <12> (V = 7730 / 10640) through a beam-5 decode.
Tolerance bf16: 2e-2 of the tensor's max per layer, 3e-2 for the 15-layer stack (operands rounded to 8 mantissa bits)."""
import numpy as np
import pytest
import torch

import lrcn_amd
from lrcn_amd import lrcn as L
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


def rel_max_err(a, b):
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


# (name, S, Cin, Cout, pool, N, route at 256 images)
LAYERS = [
    ("conv1_2", 224, 64, 64, 1, 32, "conv64"),
    ("conv2_1", 112, 64, 128, 0, 32, "conv64"),
    ("conv2_2", 112, 128, 128, 1, 32, "8p:2"),   # the 512 x 128 tile (all 160 KiB of LDS)
    ("conv3_1", 56, 128, 256, 0, 32, "8p:0"),
    ("conv3_2", 56, 256, 256, 0, 32, "8p:0"),
    ("conv3_3", 56, 256, 256, 1, 32, "8p:0"),
    ("conv4_1", 28, 256, 512, 0, 64, "8p:0"),
    ("conv4_2", 28, 512, 512, 0, 64, "8p:0"),
    ("conv4_3", 28, 512, 512, 1, 64, "8p:0"),
    ("conv5_1", 14, 512, 512, 0, 256, "8p:0"),
    ("conv5_3", 14, 512, 512, 1, 256, "8p:0"),
    # small image batches (what 8-way data parallelism leaves of 256; the decode batches): too few tiles for the chip -> the K range is
    # split over workgroups, f32 slabs, and the reduce kernel applies bias / ReLU / pool and undoes the window-major row order
    ("conv5_1@32", 14, 512, 512, 0, 32, "8p-splitk:2"),
    ("conv5_3@32", 14, 512, 512, 1, 32, "8p-splitk:2"),
    ("conv4_3@4", 28, 512, 512, 1, 4, "8p-splitk:4"),
    ("conv3_2@2", 56, 256, 256, 0, 2, "8p-splitk:3"),
]


@pytest.mark.parametrize("name,S,Cin,Cout,pool,N,route", LAYERS, ids=[l[0] for l in LAYERS])
def test_vgg_layer_real_geometry_default_route_vs_oracle(name, S, Cin, Cout, pool, N, route):
    g = torch.Generator(device="cuda")
    g.manual_seed(S * 131 + Cin + Cout + pool)
    x = torch.randn((N, Cin, S, S), generator=g, device="cuda").abs_().permute(3, 2, 1, 0)  # (W,H,Cin,N) column-major, post-ReLU-like
    w = L.jl_empty(3, 3, Cin, Cout)
    w.copy_(torch.randn((3, 3, Cin, Cout), generator=g, device="cuda") * float(np.sqrt(2.0 / (9 * Cin))))
    b = torch.randn(Cout, generator=g, device="cuda") * 0.5  # non-zero bias, some outputs cut by the ReLU
    ctx = L.Context(8, 8, 8, 17, max_B=2, max_T=1, vgg_dtype=lrcn_amd.LRCN_BF16)
    y = L.conv3x3(ctx, x, w, b, relu=True, pool=bool(pool))
    assert L.debug_route(ctx) == route, (name, L.debug_route(ctx))
    pick = [0, N - 1]
    got = L.from_jl(y[..., pick])
    ref = orc.conv3x3(L.from_jl(x[..., pick]), L.from_jl(w), b.cpu().numpy(), relu=True)
    if pool:
        ref = orc.pool2(ref)
    assert (ref == 0).any() and (ref > 0).any()
    assert rel_max_err(got, ref) <= 2e-2, (name, rel_max_err(got, ref))
    ctx.close()


def _host_weights(w):
    return ([L.from_jl(t) for t in w[0]], [t.cpu().numpy() for t in w[1]], (L.from_jl(w[2][0]), w[2][1].cpu().numpy()),
            (L.from_jl(w[3][0]), w[3][1].cpu().numpy()))


@pytest.fixture(scope="module")
def biased_vgg():
    w = L.synthetic_vgg_weights(seed=3, bias_std=0.1)
    assert all(float(b.abs().max()) > 0 for b in w[1]) and float(w[2][1].abs().max()) > 0 and float(w[3][1].abs().max()) > 0
    return w, _host_weights(w)


BENCH_ROUTES = "conv64-fused11,conv64,8p:2," + ",".join(["8p:0"] * 9)


@pytest.mark.parametrize("N,cap", [(256, 224), (256, 0), (32, 224), (34, 64)])  # 34 images, cap 64: persistent walks that end on a partial row tile
def test_full_vgg_bf16_bench_batch_nonzero_biases_vs_oracle(biased_vgg, N, cap):
    # the bench's VGG forward: N = 256 crops, capped persistent grids (dp.py sets 224), conv1_1 bias as the fused kernel's accumulator
    # input, fc6 / fc7 biases through the split-K reduce; sampled images against the oracle's fp32 stack
    w, host = biased_vgg
    g = torch.Generator(device="cuda")
    g.manual_seed(1234)
    imgs = torch.randint(0, 256, (N, 224, 224, 3), generator=g, device="cuda", dtype=torch.uint8)
    ctx = L.Context(8, 8, 8, 17, max_B=2, max_T=1, vgg_dtype=lrcn_amd.LRCN_BF16, max_images=N)
    L.vgg_load(ctx, *w)
    L.vgg_set_wg_cap(ctx, cap)
    got = L.from_jl(L.convnet_u8(ctx, imgs))
    routes = L.debug_route(ctx, 1)
    if N == 256:
        assert routes.startswith(BENCH_ROUTES + ",8p-splitk:"), routes
    # N = 256 (round 5): EIGHT oracle-checked images, chosen where the persistent walks of the capped grids change hands.
    # At 256 images a layer with P output pixels per image has exactly P row tiles of 256 rows (x Cout / 256 column tiles), and workgroup w
    # of a grid of G walks tiles w, w + G, ...: conv3_x (P = 3136, G = 224) starts its second round in image 18, conv4_x (P = 784, two
    # column tiles) in image 36, conv5_x (P = 196, 392 tiles on 196 workgroups) ends its first round with image 127 and starts the second
    # with image 128; images 0 / 255 hold the first tile of the first walk and the last tile of the last one; 1 and 254 their neighbours.
    # fc6's split-K seams lie in K (the 25088 features), so every one of these rows crosses all of them.
    pick = [0, 1, 18, 36, 127, 128, 254, 255] if N == 256 else [0, N - 1]
    x = orc.preprocess_u8(imgs[pick].cpu().numpy(), np.array(L.VGG_MEAN, np.float32))
    ref = orc.vgg_forward(host[0], host[1], host[2], host[3], x)
    assert np.isfinite(got).all()
    worst, worst_cos = 0.0, 1.0
    for i, n in enumerate(pick):   # per image: a wrong image cannot hide behind seven right ones
        err = rel_max_err(got[n], ref[i])
        cos = float((got[n] * ref[i]).sum() / (np.linalg.norm(got[n]) * np.linalg.norm(ref[i])))
        worst, worst_cos = max(worst, err), min(worst_cos, cos)
        assert err <= 3e-2 and cos > 0.999, (n, err, cos)
    print("N=%d cap=%d routes=%s images %s: worst rel_max_err=%.4g min cos=%.6f" % (N, cap, routes, pick, worst, worst_cos))
    ctx.close()


@pytest.mark.parametrize("V", [7730, 10640, 4097, 16001])
def test_beam5_large_vocab_topk_instantiations_vs_oracle(V):
    # softmax_topk_rows_kernel<Q>: V = 7730 -> <8>, 10640 -> <12>, 16001 -> <16>, 4097 -> <8> (first float4 past 4096);
    # small E = H = 64 model so the oracle's K sequential decodes stay cheap; fp32 so candidates order identically
    E = H = 64
    K, nword = 5, 6   # 7 steps: the float32 product of probabilities stays far above the denormal range
    rng = np.random.default_rng(V)
    m = orc.init_weights(E, H, H, V, seed=V)
    m.p["Wout"] *= 8.0   # spread the distribution: top-5 gaps far above fp32 noise
    m.p["bout"][:] = (rng.standard_normal((1, V)) * 0.5).astype(np.float32)
    ctx = L.Context(E, H, H, V, max_B=3 * K, max_T=1, lstm_dtype=lrcn_amd.LRCN_F32)
    param = L.model_from_arrays(m.p)
    feats = (rng.standard_normal((3, 4096)) * 0.05).astype(np.float32)
    refs = [orc.beam_search(m, feats[i], K, nword) for i in range(3)]
    for i in range(3):
        seq, p = L.beam_search(ctx, param, L.to_jl(feats[i:i + 1]), K, nword)
        assert seq == list(refs[i][0]), (V, i)
        assert refs[i][1] > 1e-30 and abs(p - refs[i][1]) <= 1e-4 * refs[i][1]
    batch = L.beam_search_batch(ctx, param, L.to_jl(feats), K, nword)
    for i in range(3):
        assert batch[i][0] == list(refs[i][0]) and abs(batch[i][1] - refs[i][1]) <= 1e-4 * refs[i][1]
    ctx.close()


def test_topk_tie_group_at_the_K_boundary_follows_probabilities():
    # The reference ranks float32 PROBABILITIES with a stable sort (lrcn.jl:652-656).  Wout = 0 makes logits = bout exactly;
    # ids 300 < 400 < 500 get logits x, nextafter(x), nextafter(nextafter(x)) (distinct, increasing) whose probabilities round
    # to the same float: the reference's top-1 is the LOWEST index 300, a logit-ranked top-K would return 500.  K = 1 so
    # that the boundary tie group decides the emitted caption.
    E = H = 16
    V, K, nword = 600, 1, 4
    m = orc.init_weights(E, H, H, V, seed=1)
    m.p["Wout"][:] = 0.0
    bout = np.full(V, -4.0, np.float32)
    bout[0] = -9.0                          # eos unlikely: the decode runs all nword + 1 steps
    x = np.float32(1e-3)
    bout[300] = x
    bout[400] = np.nextafter(x, np.float32(1.0))
    bout[500] = np.nextafter(bout[400], np.float32(1.0))
    m.p["bout"][:] = bout[None, :]
    lse = np.log(np.exp(bout.astype(np.float64)).sum())
    pf = np.exp(bout.astype(np.float64) - lse).astype(np.float32)
    assert bout[300] < bout[400] < bout[500] and pf[300] == pf[400] == pf[500], "the engineered logits no longer tie in float32"
    feat = np.zeros((1, 4096), np.float32)
    ref_seq, ref_p = orc.beam_search(m, feat[0], K, nword)
    assert list(ref_seq) == [1] + [300] * (nword + 1)
    ctx = L.Context(E, H, H, V, max_B=4, max_T=1, lstm_dtype=lrcn_amd.LRCN_F32)
    param = L.model_from_arrays(m.p)
    seq, p = L.beam_search(ctx, param, L.to_jl(feat), K, nword)
    assert seq == list(ref_seq) and abs(p - ref_p) <= 1e-5 * ref_p
    got = L.beam_search_batch(ctx, param, L.to_jl(np.concatenate([feat, feat])), K, nword)
    assert got[0][0] == list(ref_seq) and got[1][0] == list(ref_seq)
    with pytest.raises(lrcn_amd.LrcnError):
        L.beam_search(ctx, param, L.to_jl(feat), 0, nword)
    ctx.close()


def test_beam_width_larger_than_vocabulary_is_rejected():
    ctx = L.Context(8, 8, 8, 5, max_B=16, max_T=1)
    param = L.initweights(ctx, seed=1)
    feat = L.to_jl(np.zeros((1, 4096), np.float32))
    with pytest.raises(lrcn_amd.LrcnError):
        L.beam_search(ctx, param, feat, 6, 3)
    with pytest.raises(lrcn_amd.LrcnError):
        L.beam_search_batch(ctx, param, feat, 6, 3)
    seq, _ = L.beam_search(ctx, param, feat, 5, 3)  # K == V is legal
    assert seq[0] == 1
    ctx.close()


def test_vgg_forward_with_layer_inputs_beyond_4GiB_equals_256_image_chunks(biased_vgg):
    # 1280 images: conv2_2's bf16 input is 4.1 GB, past what the direct-to-LDS kernels address with 32-bit offsets -- the layer is cut
    # into launches of whole images (lrcn_api.hip launch_conv_chunked) instead of falling through to the register-staged kernel.
    # Same kernels, same tiles per image => the features equal those of five 256-image forwards (fc6/fc7 split K differently: 1e-3).
    w, _ = biased_vgg
    N = 1280
    g = torch.Generator(device="cuda")
    g.manual_seed(77)
    imgs = torch.randint(0, 256, (N, 224, 224, 3), generator=g, device="cuda", dtype=torch.uint8)
    ctx = L.Context(8, 8, 8, 17, max_B=2, max_T=1, vgg_dtype=lrcn_amd.LRCN_BF16, max_images=N)
    L.vgg_load(ctx, *w)
    big = L.from_jl(L.convnet_u8(ctx, imgs)).copy()
    routes = L.debug_route(ctx, 1)
    ctx.close()
    assert routes.startswith("conv64-fused11,conv64,8p:2," + ",".join(["8p:0"] * 9)), routes
    ctx = L.Context(8, 8, 8, 17, max_B=2, max_T=1, vgg_dtype=lrcn_amd.LRCN_BF16, max_images=256)
    L.vgg_load(ctx, *w)
    for s0 in range(0, N, 256):
        part = L.from_jl(L.convnet_u8(ctx, imgs[s0:s0 + 256]))
        assert rel_max_err(big[s0:s0 + 256], part) <= 1e-3, s0
    ctx.close()


def test_fp8_vgg_forward_beyond_2GiB_of_e4m3_input_equals_chunks(biased_vgg):
    # Round 6: the e4m3 convolution kernel addresses its input with signed 32-bit element offsets, so a layer input of 2 GiB or more
    # (conv2_2 from 1338 images: 1.6 MB of e4m3 each) must be cut into launches of whole images like the bf16 layers are at 4 GiB; before,
    # 1536 and 2048 images per forward ended in "fp8 conv layer S=112 ...: invalid argument".  1536 images == three forwards of 512
    # (same calibration, same kernels and tiles per image; fc6 / fc7 split K differently: 1e-3).
    w, _ = biased_vgg
    N = 1536
    g = torch.Generator(device="cuda")
    g.manual_seed(78)
    imgs = torch.randint(0, 256, (N, 224, 224, 3), generator=g, device="cuda", dtype=torch.uint8)
    outs = {}
    for n in (N, 512):
        ctx = L.Context(8, 8, 8, 17, max_B=2, max_T=1, vgg_dtype=lrcn_amd.LRCN_FP8, max_images=n)
        L.vgg_load(ctx, *w)
        L.vgg_calibrate(ctx, imgs[:32])
        outs[n] = np.concatenate([L.from_jl(L.convnet_u8(ctx, imgs[s0:s0 + n])).copy() for s0 in range(0, N, n)])
        ctx.close()
    assert np.isfinite(outs[N]).all()
    assert rel_max_err(outs[N], outs[512]) <= 1e-3

