"""GPU: LRCN_OPT_FUSED_UPDATE (include/lrcn.h rev 3) -- update! and the NEXT step's shadow weights in one pass over the parameters.
The training trajectory must be the one the separate passes give, bit for bit (same Adam arithmetic, same shadow values: the shadow of
a parameter is a pure function of it), in every form the update is issued: one launch (lrcn_train_step), lrcn_loss_grad + lrcn_adam_update,
the per-group launches of the data-parallel pipeline, bf16 and f32, LRCN-2f and LRCN-1f; and the contract about foreign writes holds."""
import numpy as np
import pytest
import torch

import lrcn_amd
from lrcn_amd import _lib
from lrcn_amd import lrcn as L

pytestmark = pytest.mark.gpu


def trajectory(dtype, n_layers, fused, mode, nsteps=4, E=72, H=64, V=301, B=6, T=5, touch=None):
    ctx = L.Context(E, H, H, V, max_B=B, max_T=T, lstm_dtype=dtype, n_layers=n_layers)
    ctx.set_option(_lib.LRCN_OPT_FUSED_UPDATE, 1 if fused else 0)
    param = L.initweights(ctx, seed=11)
    optim = L.initparams(param)
    grads = L.zeros_like_model(param)
    rng = np.random.default_rng(5)
    losses = []
    for k in range(nsteps):
        feats = L.to_jl((rng.standard_normal((B, 4096)) * 0.05).astype(np.float32))
        toks = rng.integers(0, V, size=(T, B)).astype(np.int32)
        if touch is not None and k == 2:   # a foreign write to the parameters between two steps
            param[7].mul_(0.5)
            if touch == "declared":
                ctx.params_touched()
        if mode == "train_step":
            losses.append(L.train_step(ctx, param, optim, grads, feats, toks, pdrop=0.0, want_loss=True))
        elif mode == "grad_then_update":
            _, val = L.lossgradient(ctx, param, feats, toks, grads=grads)
            L.update(ctx, param, grads, optim)
            losses.append(val)
        else:  # the five per-group launches, in the order the data-parallel pipeline issues them
            _, val = L.lossgradient(ctx, param, feats, toks, grads=grads)
            optim.t += 1
            for g in range(5):
                L.update_group(ctx, param, grads, optim, g)
            losses.append(val)
    torch.cuda.synchronize()
    out = [L.from_jl(p).copy() for p in param] + [L.from_jl(m).copy() for m in optim.m] + [L.from_jl(v).copy() for v in optim.v]
    ctx.close()
    return np.array(losses), out


@pytest.mark.parametrize("dtype", [lrcn_amd.LRCN_BF16, lrcn_amd.LRCN_F32])
@pytest.mark.parametrize("n_layers", [2, 1])
@pytest.mark.parametrize("mode", ["train_step", "grad_then_update", "groups"])
def test_fused_update_reproduces_the_separate_passes_bit_for_bit(dtype, n_layers, mode):
    la, pa = trajectory(dtype, n_layers, False, mode)
    lb, pb = trajectory(dtype, n_layers, True, mode)
    assert np.all(np.isfinite(la)) and la[-1] != la[0]
    # Wembed's gradient is a float-atomic scatter (order-dependent in its last bits) -> everything downstream of step 1 may differ there;
    # with few rows per token the sums here have a single addend per (token, column) most of the time: compare tightly, not bitwise
    np.testing.assert_allclose(la, lb, rtol=1e-6)
    for a, b in zip(pa, pb):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-6)


@pytest.mark.parametrize("mode", ["train_step", "groups"])
def test_fused_update_keeps_the_gate_interleaved_copies_of_the_cell_epilogue_route(mode, monkeypatch):
    """Round 6: at 256..512 rows beside the capped VGG grids a training step may run its forward recurrence as GEMM + cell epilogue
    (LRCN_LSTM_EPI=f), which reads a THIRD copy of the recurrent weights with (unit, gate)-interleaved rows.  Under
    LRCN_OPT_FUSED_UPDATE that copy, too, is written by the Adam kernel into the second shadow set (before: the route forced a full
    shadow pass every step -- 0.26 ms at the benchmark's size, more than the epilogue saved).  Five steps with the option on must follow
    the trajectory of the option off (which remakes every copy from the parameters at every call)."""
    monkeypatch.setenv("LRCN_LSTM_EPI", "f")
    E, H, V, B, T = 72, 136, 301, 256, 4
    res = []
    for fused in (False, True):
        ctx = L.Context(E, H, H, V, max_B=B, max_T=T, lstm_dtype=lrcn_amd.LRCN_BF16, vgg_dtype=lrcn_amd.LRCN_BF16, max_images=1)
        L.vgg_load(ctx, *L.synthetic_vgg_weights(seed=1))
        L.vgg_set_wg_cap(ctx, 224)
        ctx.set_option(_lib.LRCN_OPT_FUSED_UPDATE, 1 if fused else 0)
        ctx.set_option(_lib.LRCN_OPT_DETERMINISTIC, 1)   # no float atomics: the two trajectories may be compared tightly
        param = L.initweights(ctx, seed=11)
        optim = L.initparams(param)
        grads = L.zeros_like_model(param)
        rng = np.random.default_rng(5)
        losses = []
        for k in range(5):
            feats = L.to_jl((rng.standard_normal((B, 4096)) * 0.05).astype(np.float32))
            toks = rng.integers(0, V, size=(T, B)).astype(np.int32)
            if mode == "train_step":
                losses.append(L.train_step(ctx, param, optim, grads, feats, toks, pdrop=0.0, want_loss=True))
            else:
                _, val = L.lossgradient(ctx, param, feats, toks, grads=grads)
                optim.t += 1
                for g in range(5):
                    L.update_group(ctx, param, grads, optim, g)
                losses.append(val)
        torch.cuda.synchronize()
        res.append((np.array(losses), [L.from_jl(p).copy() for p in param]))
        ctx.close()
    (la, pa), (lb, pb) = res
    assert la[-1] < la[0]
    # (a STALE interleaved copy -- the previous step's recurrent weights -- moves the loss in its 4th digit; the two update kernels' own
    # last-bit differences stay below 1e-8)
    np.testing.assert_allclose(la, lb, rtol=1e-7)
    for k, (a, b) in enumerate(zip(pa, pb)):
        np.testing.assert_allclose(a, b, rtol=0, atol=2e-5, err_msg=str(k))   # (Adam turns a last-bit gradient difference into a fraction of lr = 1e-3)


def test_first_fused_step_equals_unfused_exactly():
    """One step from identical state: parameters and moments after the fused kernel are bit-identical to adam_kernel's (Wembed excluded:
    its GRADIENT is an atomic sum)."""
    _, pa = trajectory(lrcn_amd.LRCN_BF16, 2, False, "train_step", nsteps=1)
    _, pb = trajectory(lrcn_amd.LRCN_BF16, 2, True, "train_step", nsteps=1)
    for k, (a, b) in enumerate(zip(pa, pb)):
        if k % 9 == 6:
            np.testing.assert_allclose(a, b, rtol=0, atol=1e-7)
        else:
            assert np.array_equal(a, b), k


def test_foreign_write_needs_params_touched():
    """The contract of LRCN_OPT_FUSED_UPDATE: a parameter array written by the caller between two steps is only seen by the next step if
    the caller says so.  Declared -> the trajectory of the unfused context (which re-reads the parameters every call); undeclared -> the
    step after the write still runs on the old shadows (its loss is the UNMODIFIED model's), which is exactly why the option is opt-in."""
    l_ref, _ = trajectory(lrcn_amd.LRCN_BF16, 2, False, "train_step", touch="declared")
    l_dec, _ = trajectory(lrcn_amd.LRCN_BF16, 2, True, "train_step", touch="declared")
    l_und, _ = trajectory(lrcn_amd.LRCN_BF16, 2, True, "train_step", touch="undeclared")
    l_none, _ = trajectory(lrcn_amd.LRCN_BF16, 2, True, "train_step", touch=None)
    np.testing.assert_allclose(l_dec, l_ref, rtol=1e-6)
    assert abs(l_ref[2] - l_none[2]) > 1e-4 * abs(l_none[2])          # the write matters ...
    np.testing.assert_allclose(l_und[2], l_none[2], rtol=1e-6)        # ... and goes unseen for one step when undeclared


def test_options_are_validated():
    ctx = L.Context(32, 32, 32, 50, max_B=2, max_T=2, lstm_dtype=lrcn_amd.LRCN_F32)
    with pytest.raises(L.LrcnError):
        ctx.set_option(99, 1)
    with pytest.raises(L.LrcnError):
        ctx.set_option(_lib.LRCN_OPT_FUSED_UPDATE, 2)
    with pytest.raises(L.LrcnError):
        ctx.set_option(_lib.LRCN_OPT_CONV_CHUNK_BYTES, -1)
    ctx.set_option(_lib.LRCN_OPT_DETERMINISTIC, 1)
    ctx.set_option(_lib.LRCN_OPT_DETERMINISTIC, 0)
    assert _lib.lib().lrcn_abi_version() == _lib.LRCN_ABI_VERSION
    ok, why = L.comm_probe(ctx)
    assert ok, why
    ctx.close()


@pytest.mark.parametrize("n_layers", [2, 1])
def test_shadows_refreshed_per_group_equal_the_shadow_pass(n_layers):
    """lrcn_refresh_shadows_group (rev 5): a host that updates the parameters itself (the sharded update: lrcn_adam_update_flat on slices)
    refreshes the next step's shadow weights group by group; once all five groups are in, the next lossgradient must skip its shadow
    pass and give BIT-identical gradients to the same call after lrcn_params_touched (which makes the shadows afresh); an unfinished
    sequence (four of five groups) must NOT be taken for a finished one, neither now nor after a later complete sequence."""
    E, H, V, B, T = 72, 64, 301, 6, 5
    rng = np.random.default_rng(9)
    ctx = L.Context(E, H, H, V, max_B=B, max_T=T, lstm_dtype=lrcn_amd.LRCN_BF16, n_layers=n_layers)
    ctx.set_option(_lib.LRCN_OPT_FUSED_UPDATE, 1)
    param = L.initweights(ctx, seed=3)
    feats = L.to_jl((rng.standard_normal((B, 4096)) * 0.05).astype(np.float32))
    toks = rng.integers(0, V, size=(T, B)).astype(np.int32)
    mask1 = ((rng.random((T + 1, B, E if n_layers == 2 else E + H // 2)) > 0.4) / 0.6).astype(np.float32)
    mask2 = ((rng.random((T + 1, B, H)) > 0.4) / 0.6).astype(np.float32) if n_layers == 2 else None

    def grads_now():
        g, val = L.lossgradient(ctx, param, feats, toks, mask1=mask1, mask2=mask2)
        torch.cuda.synchronize()
        return val, [L.from_jl(x).copy() for x in g]

    grads_now()                                     # makes the first shadow set from the initial parameters
    for k, p in enumerate(param):                   # the host changes every parameter itself ...
        if p.numel():
            p.mul_(1.0 + 0.01 * (k + 1))
    for g in (0, 1, 2, 3):                          # ... and refreshes four of the five groups only
        L.refresh_shadows_group(ctx, param, g)
    v_partial, g_partial = grads_now()              # must have made its shadows afresh (the sequence was unfinished)
    ctx.params_touched()
    v_ref, g_ref = grads_now()
    assert v_partial == v_ref
    for n, a, b in zip(L.PARAM_NAMES, g_partial, g_ref):
        if n != "Wembed":                           # (a float-atomic scatter: order-dependent in its last bits)
            np.testing.assert_array_equal(a, b, err_msg=n)
    for p in param:
        if p.numel():
            p.mul_(0.97)
    for g in (4, 2, 0, 1, 3):                       # all five, any order: the refreshed set becomes current
        L.refresh_shadows_group(ctx, param, g)
    v_new, g_new = grads_now()
    ctx.params_touched()
    v_ref2, g_ref2 = grads_now()
    assert v_new == v_ref2 and v_new != v_ref
    for n, a, b in zip(L.PARAM_NAMES, g_new, g_ref2):
        if n != "Wembed":
            np.testing.assert_array_equal(a, b, err_msg=n)
    ctx.close()
