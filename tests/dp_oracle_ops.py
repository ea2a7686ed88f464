"""CPU stand-ins for the device operations of dp.DataParallelTrainer (test infrastructure: the oracle plays the HIP calls).
Used by tests/test_dp_gloo.py and by bench.py's LRCN_BENCH_DRYRUN leg (tests/test_bench_launcher.py) -- never by the product path."""
import numpy as np
import torch

from oracle import oracle as orc


class OracleOps:
    def __init__(self, dims):
        self.dims = dims
        self._loss = 0.0

    def _model(self, param):
        E, H1, H2, V = self.dims
        return orc.Model(E, H1, H2, V, {n: p.numpy() for n, p in zip(orc.PARAM_NAMES, param)})

    def vgg(self, img):
        raise AssertionError("features are given in this test")

    def lossgradient(self, param, feats, tokens, norm_B, pdrop, seed, grads):
        val, g = orc.loss(self._model(param), feats.numpy(), tokens, norm_B=norm_B, want_grad=True)
        self._loss = val
        for n, t in zip(orc.PARAM_NAMES, grads):
            t.copy_(torch.as_tensor(g.p[n]))

    def update(self, param, grads, optim):
        optim.t += 1
        for p, g, m, v in zip(param, grads, optim.m, optim.v):
            w, mm, vv = (np.asfortranarray(a.numpy()) for a in (p, m, v))
            orc.adam(w, np.asfortranarray(g.numpy()), mm, vv, optim.t)
            p.copy_(torch.as_tensor(w)); m.copy_(torch.as_tensor(mm)); v.copy_(torch.as_tensor(vv))

    def last_loss(self):
        return self._loss

    def loss(self, param, feats, tokens):
        return orc.loss(self._model(param), feats.numpy(), tokens)


class OracleGroupOps(OracleOps):
    """The same, plus the CPU stand-ins of what the per-group [all-reduce -> Adam] pipeline needs (dp.py
    _reduce_and_update_groups / the bucketed _allreduce_async): "streams" are labels, the gradient-ready "events" are already
    complete (lossgradient is synchronous here), and every call is logged so the test can check the ORDER the trainer drives."""

    def __init__(self, dims):
        super().__init__(dims)
        self.log = []

    def make_streams(self, n):
        self.log.append(("make_streams", n))
        return ["bucket%d" % k for k in range(n)]

    def stream_ctx(self, stream):
        import contextlib
        return contextlib.nullcontext()

    def grad_group_wait(self, group, stream):
        self.log.append(("wait", group, stream))

    def update_group(self, param, grads, optim, group, stream):
        from lrcn_amd import dp
        self.log.append(("adam", group, stream, optim.t))
        for k in dp.GRAD_GROUPS[group]:
            w, mm, vv = (np.asfortranarray(a.numpy()) for a in (param[k], optim.m[k], optim.v[k]))
            orc.adam(w, np.asfortranarray(grads[k].numpy()), mm, vv, optim.t)
            param[k].copy_(torch.as_tensor(w)); optim.m[k].copy_(torch.as_tensor(mm)); optim.v[k].copy_(torch.as_tensor(vv))

    def join(self, streams):
        self.log.append(("join", len(streams)))

    def update_flat(self, w, g, m, v, optim, stream):
        """Adam on one flat run of floats (the sharded update): same arithmetic as orc.adam, on 1-D tensors."""
        self.log.append(("adam_flat", stream, optim.t, w.numel()))
        ww, mm, vv = (np.ascontiguousarray(a.numpy()) for a in (w, m, v))
        orc.adam(ww, np.ascontiguousarray(g.numpy()), mm, vv, optim.t)
        w.copy_(torch.as_tensor(ww)); m.copy_(torch.as_tensor(mm)); v.copy_(torch.as_tensor(vv))


class HostAdam:
    def __init__(self, param):
        self.t = 0
        self.m = [torch.zeros_like(p) for p in param]
        self.v = [torch.zeros_like(p) for p in param]




def make(E, H, V, seed=0):
    """-> (param list of torch tensors, HostAdam, OracleOps) for a tiny LRCN-2f model."""
    m = orc.init_weights(E, H, H, V, seed=seed)
    param = [torch.as_tensor(np.array(m.p[n])) for n in orc.PARAM_NAMES]
    return param, HostAdam(param), OracleOps((E, H, H, V))
