"""Shared parity checks of the GPU tests (tests/ only: imports the oracle, the checker).

bf16 kernels are compared ELEMENTWISE with the bf16-EMULATING oracle (oracle/lrcn_oracle.h, ORC_EMULATE_BF16: the same CPU
restatement of lrcn.jl rounded to bfloat16 exactly where liblrcn_hip.so stores or feeds bf16) -- what is left between the two is
summation order and the occasional one-ulp flip of a rounding whose argument differs in the last float32 bit:

    loss        |d| <= 1e-6 |loss|             (measured 3e-8 .. 5e-8; against the f32 oracle it is ~1e-7 .. 3e-7 only by luck of averaging)
    gradients   |d| <= 5e-3 |ref| + 2.5e-3 max|ref|   per element,  and  ||d|| <= 3e-3 ||ref||   per tensor
                (measured worst element 1.8e-3 max|ref| -- Wcnn, whose contraction is only B long, so one flipped bf16 of d x_cnn shows --
                 and 2e-4 .. 8e-4 for the others; the same kernels against the UN-emulated f32 oracle sit at 3e-3 .. 8e-3 max|ref|.)

A wrong 5 % term (a dropped bias column, a mis-scaled split-K slab, a stale K-padding column) moves ||d|| / ||ref|| to ~5e-2 and single
elements by 5e-2 |ref|: both bounds catch it; the cosine > 0.99 these tests used before did not.
"""
import numpy as np

from lrcn_amd import lrcn as L
from oracle import oracle as orc

BF16_LOSS_RTOL = 1e-6
BF16_GRAD_RTOL = 5e-3
BF16_GRAD_ATOL_FRAC = 2.5e-3
BF16_GRAD_NORM = 3e-3


def emulated_reference(model, feats, tokens, **kw):
    """loss and the nine gradients from the bf16-emulating oracle (kw: norm_B, mask1, mask2)."""
    with orc.emulate_bf16():
        return orc.loss(model, feats, tokens, want_grad=True, **kw)


def assert_bf16_matches_emulation(val, grads, ref_loss, ref_g, what=""):
    """val / grads: what the HIP library returned (grads: device tensors in reference layout, or numpy arrays)."""
    assert abs(val - ref_loss) <= BF16_LOSS_RTOL * abs(ref_loss), (what, val, ref_loss)
    for n, g in zip(orc.PARAM_NAMES, grads):
        r = ref_g.p[n].astype(np.float64)
        if r.size == 0:
            continue
        a = (g if isinstance(g, np.ndarray) else L.from_jl(g)).astype(np.float64)
        assert a.shape == r.shape, (what, n, a.shape, r.shape)
        d = np.abs(a - r)
        mx = np.abs(r).max()
        tol = BF16_GRAD_RTOL * np.abs(r) + BF16_GRAD_ATOL_FRAC * mx
        bad = d > tol
        assert not bad.any(), "%s %s: %d of %d elements outside rtol %g + %g max|ref|; worst |d| %.3e at |ref| %.3e (max|ref| %.3e)" % (
            what, n, int(bad.sum()), r.size, BF16_GRAD_RTOL, BF16_GRAD_ATOL_FRAC, d.max(), np.abs(r).ravel()[d.argmax()], mx)
        rel = np.linalg.norm(a - r) / (np.linalg.norm(r) + 1e-300)
        assert rel <= BF16_GRAD_NORM, "%s %s: ||d|| / ||ref|| = %.3e > %g" % (what, n, rel, BF16_GRAD_NORM)
