"""GPU: the > 4 GiB input cut of the implicit-GEMM convolutions (lrcn_api.hip launch_conv_chunked: per-chunk A / C pointer offsets,
pooled versus un-pooled output stride, per-chunk M) -- a product path for max_images above ~1171 that no ordinary test batch reaches.
LRCN_OPT_CONV_CHUNK_BYTES lowers the cut so that a handful of images already runs as several launches per layer; the features must be the
single-launch features.  Where a chunk is small enough to change a layer's kernel route (split-K below 64 / 16 images) the
summation order changes with it, so the comparison is bit-exact only when the per-layer routes are the same, and to summation-order
accuracy otherwise -- a wrong stride or offset is garbage, not a rounding difference."""
import numpy as np
import pytest
import torch

import lrcn_amd
from lrcn_amd import _lib
from lrcn_amd import lrcn as L

pytestmark = pytest.mark.gpu

# input bytes per image of the layers on the chunked route (bf16): conv2_2 3.2 MB, conv3_1 0.8 MB, conv3_2/3 1.6 MB, conv4_1 0.4 MB,
# conv4_2/3 0.8 MB, conv5_x 0.2 MB -> limits that cut (some of) them into 1..3 images per launch at N = 7
LIMITS = [7_000_000, 1_700_000, 450_000, 210_000]


@pytest.mark.parametrize("vgg_dtype,tol", [(lrcn_amd.LRCN_BF16, 8e-3), (lrcn_amd.LRCN_F32, 2e-5), (lrcn_amd.LRCN_FP8, 6e-2)],
                         ids=["bf16", "f32", "fp8"])
def test_chunked_convolutions_equal_the_single_launch(vgg_dtype, tol):
    N = 7   # odd: the last chunk is ragged
    ctx = L.Context(8, 8, 8, 17, max_B=2, max_T=1, vgg_dtype=vgg_dtype, max_images=N)
    L.vgg_load(ctx, *L.synthetic_vgg_weights(seed=1, bias_std=0.1))
    g = torch.Generator(device="cuda")
    g.manual_seed(11)
    img = torch.randint(0, 256, (N, 224, 224, 3), generator=g, device="cuda", dtype=torch.uint8)
    if vgg_dtype == lrcn_amd.LRCN_FP8:
        L.vgg_calibrate(ctx, img)
    ref = L.from_jl(L.convnet_u8(ctx, img)).copy()
    ref_routes = L.debug_route(ctx, 1)
    scale = np.abs(ref).max()
    assert scale > 0 and np.isfinite(ref).all()
    for limit in LIMITS:
        ctx.set_option(_lib.LRCN_OPT_CONV_CHUNK_BYTES, limit)
        got = L.from_jl(L.convnet_u8(ctx, img)).copy()
        routes = L.debug_route(ctx, 1)
        if routes == ref_routes:
            assert np.array_equal(got, ref), (limit, np.abs(got - ref).max())
        else:
            assert np.abs(got - ref).max() <= tol * scale, (limit, np.abs(got - ref).max() / scale, routes, ref_routes)
    ctx.set_option(_lib.LRCN_OPT_CONV_CHUNK_BYTES, 0)
    again = L.from_jl(L.convnet_u8(ctx, img))
    assert np.array_equal(again, ref)
    ctx.close()
