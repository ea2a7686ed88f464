"""Development probe: bf16 HIP lossgradient vs the bf16-EMULATING oracle (oracle/lrcn_oracle.h ORC_EMULATE_BF16), elementwise.
Prints, per tensor, max|d| / max|ref| and the worst |d| / (atol + rtol |ref|) for the tolerance the tests use."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import lrcn_amd  # noqa: E402
from lrcn_amd import lrcn as L  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def probe(E, H, V, B, T, nl=2, masks=True, scale=1.0, env=None, beside=False, rtol=5e-3, afrac=1e-3):
    for k, v in (env or {}).items():
        os.environ[k] = v
    rng = np.random.default_rng(B * 7 + T)
    m = orc.init_weights(E, H, H, V, seed=7, n_layers=nl)
    for n in ("W1", "W2", "Wout", "Wproj"):
        m.p[n] *= scale
    feats = (rng.standard_normal((B, 4096)) * 0.05).astype(np.float32)
    tokens = rng.integers(0, V, size=(T, B)).astype(np.int32)
    kw = {}
    if masks:
        X1 = E if nl == 2 else E + H // 2
        kw["mask1"] = ((rng.random((T + 1, B, X1)) > 0.3) / 0.7).astype(np.float32)
        if nl == 2:
            kw["mask2"] = ((rng.random((T + 1, B, H)) > 0.3) / 0.7).astype(np.float32)
    f32_loss, f32_g = orc.loss(m, feats, tokens, want_grad=True, **kw)
    with orc.emulate_bf16():
        ref_loss, ref_g = orc.loss(m, feats, tokens, want_grad=True, **kw)
    ctx = L.Context(E, H, H, V, max_B=B, max_T=T, lstm_dtype=lrcn_amd.LRCN_BF16, vgg_dtype=lrcn_amd.LRCN_BF16, max_images=1 if beside else 0,
                    n_layers=nl)
    if beside:
        L.vgg_load(ctx, *L.synthetic_vgg_weights(seed=1))
        L.vgg_set_wg_cap(ctx, 224)
    grads, val = L.lossgradient(ctx, L.model_from_arrays(m.p), L.to_jl(feats), tokens, **kw)
    print("E=%d H=%d V=%d B=%d T=%d nl=%d env=%s beside=%s: loss hip %.7f emu %.7f (rel %.1e) f32 %.7f (rel %.1e)" % (
        E, H, V, B, T, nl, env, beside, val, ref_loss, abs(val - ref_loss) / abs(ref_loss), f32_loss, abs(val - f32_loss) / abs(f32_loss)))
    for n, g in zip(orc.PARAM_NAMES, grads):
        r = ref_g.p[n]
        if r.size == 0:
            continue
        a = L.from_jl(g).astype(np.float64)
        d = np.abs(a - r)
        mx = np.abs(r).max()
        worst = (d / (afrac * mx + rtol * np.abs(r))).max()
        d32 = np.abs(a - f32_g.p[n])
        print("   %-7s max|ref| %.2e  max|d|/max|ref| %.2e  worst/(tol) %.2f   [vs f32 oracle: %.2e]" % (n, mx, d.max() / mx, worst, d32.max() / mx))
    ctx.close()
    for k in (env or {}):
        del os.environ[k]


if __name__ == "__main__":
    probe(64, 64, 301, 8, 6)
    probe(256, 256, 1000, 24, 5, scale=2.0)
    probe(256, 256, 1000, 64, 5, scale=2.0)
    probe(256, 256, 1000, 160, 5, scale=2.0)
    probe(256, 256, 1000, 256, 5, scale=2.0)
    probe(256, 256, 1000, 256, 5, scale=2.0, beside=True)
    probe(256, 256, 1000, 256, 5, scale=2.0, beside=True, env={"LRCN_LSTM_EPI": "1"})
    probe(320, 320, 777, 24, 5, env={"LRCN_SKINNY": "0", "LRCN_LSTM_FUSED": "0"})
    probe(256, 256, 1000, 64, 7, env={"LRCN_8P": "force"})
    probe(256, 256, 1000, 64, 7, env={"LRCN_8P": "0"})
    probe(96, 128, 501, 12, 5, nl=1)
    probe(1000, 1000, 10640, 32, 3)
