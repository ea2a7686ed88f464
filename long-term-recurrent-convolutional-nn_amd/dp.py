"""Data-parallel training step: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI).

The reference is single-device; what shards is the batch: each rank runs VGG forward + LSTM
forward/backward on its B_global/N rows, normalising by the GLOBAL batch (lrcn.jl:564-568 uses the global
`batchsize`), then ONE all-reduce(SUM) of the 9 gradient tensors (held in one flat buffer: one collective of
4 B/param) and an identical Adam step everywhere.  VGG is frozen (its output does not depend on the LSTM
parameters), so the all-reduce of step k is overlapped with the VGG forward of step k+1 while keeping exactly
synchronous-SGD semantics: RCCL runs on its own stream, the compute stream waits for it only before Adam.
On the GPU the VGG forward of step k+1 also runs on a SIDE HIP stream, concurrently with the LSTM step k (hundreds of small,
latency-bound launches that leave most CUs idle): the frozen extractor shares nothing with the LSTM but read-only weights.
"""
import collections
import ctypes as C
import os

import numpy as np
import torch
import torch.distributed as dist

from . import lrcn as L


def shard_rows(B_global, world, rank):
    """Contiguous row shard of a global batch (equal-length captions, so every rank sees the same T)."""
    if B_global % world:
        raise L.LrcnError("global batch %d is not divisible by world size %d" % (B_global, world))
    b = B_global // world
    return slice(rank * b, (rank + 1) * b)


def vgg_wg_cap_for(device, rows=256, chunk=1):
    """Convolution-grid cap for the side-stream VGG forward: 7/8 of the CUs (224 of 256).  The capped kernels walk their
    tiles persistently and leave 32 CUs on which the LSTM step's chain of small dependent launches never queues behind
    27-us convolution workgroups.  Measured on one MI355X (ms/step, cap 0 -> 224): B=32 1.86 -> 1.72, B=64 2.78 -> 2.49,
    B=128 4.60 -> 4.44, B=256 8.58 -> 8.31 (same box); 192 and 240 are slower than either (tile-count quantisation).
    Several batches per forward (chunk > 1, rows <= 64 per step): the forward of 256 images then runs beside 8 (4) whole LSTM steps,
    whose Adam and time-batched GEMMs want CUs too -- round 4, same box, ms/step at 32 rows with 8 steps per forward: cap 224 1.65,
    196 1.53, 176 1.49, 160 1.37, 147 1.38, 128 1.43 (one forward per step, cap 224: 1.60); at 64 rows with 4 steps per forward:
    224 2.47, 196 2.19, 176 2.17, 160 2.29 (one per step: 2.28)."""
    ncu = torch.cuda.get_device_properties(device).multi_processor_count
    if chunk > 1 and rows <= 32:
        return (ncu * 5 // 8) & ~7      # 160 of 256
    if chunk > 1 and rows <= 64:
        return (ncu * 11 // 16) & ~7    # 176 of 256
    return (ncu * 7 // 8) & ~7


def streams_share_a_queue(a, b, device):
    """HIP multiplexes its streams onto a few hardware queues (GPU_MAX_HW_QUEUES, default 4): two streams on one queue run IN ORDER, whatever
    the program says.  Measured, not assumed: a ~2 ms sleep kernel on `a`, then a tiny kernel on `b` -- if b's finishes only after a's, they
    share a queue.  (Round 4: a gradient-group stream that shared the VGG side stream's queue made its Adam wait for the whole forward --
    one step in eight took 9 ms instead of 1.1 with 8 steps per forward.)"""
    torch.cuda.synchronize(device)
    e0, e_long, e_short = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    y = torch.zeros(16, device=device)
    with torch.cuda.stream(a):
        e0.record()
        torch.cuda._sleep(4_000_000)
        e_long.record()
    with torch.cuda.stream(b):
        y.add_(1.0)
        e_short.record()
    torch.cuda.synchronize(device)
    return e0.elapsed_time(e_short) > 0.5 * e0.elapsed_time(e_long)


def independent_stream(device, avoid, tries=12):
    """A new stream that shares its hardware queue with none of `avoid` (streams whose work must never be serialised with it).  The
    rejected candidates stay alive so that the next candidate lands on another queue.  None if every try collides."""
    keep = []
    for _ in range(tries):
        s = torch.cuda.Stream(device=device)
        if not any(streams_share_a_queue(o, s, device) or streams_share_a_queue(s, o, device) for o in avoid if o is not None):
            independent_stream._rejected = getattr(independent_stream, "_rejected", []) + keep
            return s
        keep.append(s)
    independent_stream._rejected = getattr(independent_stream, "_rejected", []) + keep
    return None


# lossgradient finalises the gradients in this order of groups of (adjacent) parameters -- include/lrcn.h, LRCN_GRAD_GROUPS
GRAD_GROUPS = [(7, 8), (2, 3), (4, 5), (0, 1), (6,)]


def flat_model_like(shapes, device="cuda", group_align=0):
    """One flat float32 buffer + 9 column-major views into it (so the gradient all-reduce is one collective).
    group_align > 0: every gradient group (GRAD_GROUPS: adjacent tensors) is padded with zeros to a multiple of `group_align` elements, so
    that a reduce-scatter / all-gather can cut it into equal shards; returns (flat, views, ranges) with ranges[g] = the padded [a, b) of
    group g in the order of GRAD_GROUPS."""
    sizes = [int(np.prod(s)) for s in shapes]
    offs, off = [], 0
    group_end = {max(g): g for g in GRAD_GROUPS}
    starts = {}
    for k, n in enumerate(sizes):
        if group_align and any(k == min(g) for g in GRAD_GROUPS):
            starts[[g for g in GRAD_GROUPS if k == min(g)][0]] = off
        offs.append(off)
        off += n
        if group_align and k in group_end:
            off = -(-off // group_align) * group_align
            starts[("end", group_end[k])] = off
    flat = torch.zeros(off, device=device, dtype=torch.float32)
    views = []
    for s, n, o in zip(shapes, sizes, offs):
        views.append(flat[o:o + n].view(*reversed(s)).permute(*reversed(range(len(s)))))
    if not group_align:
        return flat, views
    return flat, views, [(starts[g], starts[("end", g)]) for g in GRAD_GROUPS]


class HipOps:
    """The device operations a trainer step is made of -- all of them liblrcn_hip calls."""

    def __init__(self, ctx, mean=L.VGG_MEAN):
        self.ctx = ctx
        self.mean = mean   # channel means of read_image_data (lrcn.jl:770); None = the averageImage registered with the context

    def vgg(self, img_u8, feats=None, normalize=False):
        # host crops are uploaded through the staging buffers first
        return L.convnet_u8(self.ctx, img_u8, mean=self.mean, feats=feats, normalize=normalize)

    def loss(self, param, feats, tokens):
        """Forward-only loss of one batch (average_loss's body, lrcn.jl:452-475): pdrop 0, normalised by the batch's own size."""
        return L.avg_loss_batch(self.ctx, param, feats, tokens)

    def vgg_blocks(self, img_u8, rows, feats=None, normalize=False):
        """One forward for the crops of several batches -> list of rows x 4096 feature blocks (lrcn_vgg_forward_u8_blocks)."""
        return L.convnet_u8_blocks(self.ctx, img_u8, rows, mean=self.mean, feats=feats, normalize=normalize)

    def upload(self, host_u8):
        """Start the host -> device copy of a batch of crops on the library's copy stream (returns at once)."""
        return L.upload_crops(self.ctx, host_u8)

    def params_touched(self):
        self.ctx.params_touched()

    def side_stream(self):
        """A second HIP stream for the VGG forward (None = run everything in order on the caller's stream)."""
        if os.environ.get("LRCN_OVERLAP_VGG", "1")[:1] == "0":
            return None
        prio = int(os.environ.get("LRCN_VGG_STREAM_PRIO", "0"))  # -1 = high priority queue for the convolutions
        return torch.cuda.Stream(device=self.ctx.device, priority=prio)

    def set_vgg_wg_cap(self, cap):
        self.ctx._call("lrcn_vgg_set_wg_cap", int(cap))

    def lossgradient(self, param, feats, tokens, norm_B, pdrop, seed, grads):
        L.lossgradient(self.ctx, param, feats, tokens, norm_B=norm_B, pdrop=pdrop, seed=seed, grads=grads,
                       want_loss=False)

    def update(self, param, grads, optim):
        L.update(self.ctx, param, grads, optim)

    def update_group(self, param, grads, optim, group, stream):
        L.update_group(self.ctx, param, grads, optim, group, stream)

    def update_flat(self, w, g, m, v, optim, stream):
        L.update_flat(self.ctx, w, g, m, v, optim, stream)

    def refresh_shadows_group(self, param, group, stream):
        L.refresh_shadows_group(self.ctx, param, group, stream)

    def last_loss(self):
        return L.last_loss(self.ctx)

    def grad_group_wait(self, group, stream):
        """`stream` waits until the gradients of GRAD_GROUPS[group] of the last lossgradient are final."""
        self.ctx._call("lrcn_grad_group_wait", group, C.c_void_p(stream.cuda_stream))

    # the [all-reduce -> Adam] pipeline of the torch.distributed backend: ONE update stream for all gradient groups (they become final in
    # order, so one stream loses nothing), chosen so that it does not share a hardware queue with the VGG side stream -- see
    # streams_share_a_queue.  LRCN_DP_GROUP_STREAMS=1: one stream per group, as rounds 2-3 had it.
    def make_streams(self, n, avoid=()):
        if os.environ.get("LRCN_DP_GROUP_STREAMS", "0")[:1] == "1":
            return [torch.cuda.Stream(device=self.ctx.device) for _ in range(n)]
        s = independent_stream(torch.device("cuda", self.ctx.device), list(avoid))
        self.update_stream_probed = s is not None   # False: every candidate shared a queue with one of `avoid` (reported by describe())
        return [s or torch.cuda.Stream(device=self.ctx.device)] * n

    def stream_ctx(self, stream):
        return torch.cuda.stream(stream)

    def join(self, streams):
        main = torch.cuda.current_stream(self.ctx.device)
        seen = []
        for s in streams:
            if not any(s is t for t in seen):
                main.wait_stream(s)  # the next step's shadow-weight pass reads the updated parameters
                seen.append(s)

    # the C-ABI backend: RCCL inside liblrcn_hip (lrcn_comm_init / lrcn_train_step_dp)
    def comm_probe(self):
        return L.comm_probe(self.ctx)

    def comm_init(self, world, rank, unique_id):
        L.comm_init(self.ctx, world, rank, unique_id)

    def set_embed_rows_buffer(self, rows, tok):
        L.set_embed_rows_buffer(self.ctx, rows, tok)

    def embed_grad_from_rows(self, rows, tok, n_rows, grad, stream):
        L.embed_grad_from_rows(self.ctx, rows, tok, n_rows, grad, stream)

    def set_wg_stream(self, avoid=()):
        s = independent_stream(torch.device("cuda", self.ctx.device), list(avoid))
        if s is not None:
            self.ctx._call("lrcn_set_wg_stream", C.c_void_p(s.cuda_stream))
            self._wg_stream = s   # kept alive for the context's lifetime

    def comm_set_stream(self, avoid=()):
        """Give the library's communicator an update stream that shares its hardware queue with none of `avoid` (the VGG side stream)."""
        s = independent_stream(torch.device("cuda", self.ctx.device), list(avoid))
        if s is not None:
            L.comm_set_stream(self.ctx, s)
            self._comm_stream = s   # kept alive for the communicator's lifetime

    def set_fused_update(self, on):
        """LRCN_OPT_FUSED_UPDATE: Adam writes the next step's shadow weights; the trainer owns the parameters between steps."""
        self.ctx.set_option(L._lib.LRCN_OPT_FUSED_UPDATE, 1 if on else 0)

    def comm_destroy(self):
        try:
            L.comm_destroy(self.ctx)
        except L.LrcnError:
            pass

    def train_step_dp(self, param, grads, optim, feats, tokens, norm_B, pdrop, seed):
        L.train_step_dp(self.ctx, param, optim, grads, feats, tokens, norm_B=norm_B, pdrop=pdrop, seed=seed)


class DataParallelTrainer:
    """train1's batch loop body (lrcn.jl:369-394) sharded over ranks. world_size 1 = no collective.
    `ops` defaults to the HIP operations; tests of the collective logic on CPU (gloo) inject their own.

    The trainer OWNS the parameters between steps: it turns LRCN_OPT_FUSED_UPDATE on (unless LRCN_FUSED_UPDATE=0 or the sharded update
    is selected), under which every update! also writes the NEXT step's shadow weights and the next step skips its shadow pass.  Code that
    writes the parameter tensors itself between two steps -- a checkpoint restored into the same tensors, weight clipping,
    re-initialisation -- must say so with trainer.params_touched() (restore() does it), or the next step trains on stale shadows.
    close() turns the option off again: it belongs to the context and would otherwise outlive the trainer."""

    def __init__(self, ctx, param, optim, B_global, world=1, rank=0, pdrop=0.4, seed=0, group=None, ops=None, backend=None, shard_adam=None,
                 normalize_features=False, gclip=0.0, vgg_chunk=1, rows=None, emulate_shards=0):
        """backend (world > 1): "torch" (default) = the per-group all-reduces are issued from here through torch.distributed's RCCL
        process group; "abi" = RCCL inside liblrcn_hip (lrcn_comm_init + lrcn_train_step_dp: one C call per step, what a Julia host
        would drive; the unique id travels over torch.distributed's group).  LRCN_DP_BACKEND overrides.  "abi" stays opt-in until a
        multi-GPU run has compared it with "torch" step for step: it has only ever met one rank."""
        self.ops = ops if ops is not None else HipOps(ctx)
        self.ctx, self.param, self.optim = ctx, param, optim
        self.B_global, self.world, self.rank = B_global, world, rank
        self.pdrop, self.seed = pdrop, seed
        self.group = group
        self.normalize_features = bool(normalize_features)   # input / sum(input) on the VGG's output (lrcn.jl:595-597)
        # --gclip (parsed and ignored by the reference): clip the GLOBAL gradient norm, i.e. after the exchange and before update! -- which
        # rules out the per-group [all-reduce -> Adam] pipeline: one all-reduce of the flat buffer, the norm, one Adam launch
        self.gclip = float(gclip or 0.0)
        # emulate_shards = N > 1 (one process, world = 1; bench.py --emulate-world N --shard-adam): rank 0's side of an N-rank SHARDED update
        # with the two collectives stubbed by device copies of the bytes a rank receives -- Adam on 1/N of every gradient group, the other
        # (N-1)/N of the parameters are NOT updated.  A timing aid, never a training mode.
        self._emu_shards = int(emulate_shards) if world == 1 and emulate_shards and emulate_shards > 1 else 0
        backend = os.environ.get("LRCN_DP_BACKEND") or backend or "torch"
        if backend == "auto":
            backend = "torch"
        if backend not in ("abi", "torch"):
            raise L.LrcnError("unknown data-parallel backend %r" % (backend,))
        if backend == "abi" and (not hasattr(self.ops, "train_step_dp") or self.gclip > 0):
            backend = "torch"
        self.backend = backend
        self.backend_note = ""
        # LRCN_DP_FORCE_PIPELINE=1 (tests on a one-GPU box): a ONE-rank job runs exactly the control flow of N > 1 -- communicator, per-group
        # [gradient event -> all-reduce -> Adam] pipeline, collectives issued for real (the sum over one rank is the identity)
        self._multi = world > 1 or (os.environ.get("LRCN_DP_FORCE_PIPELINE", "0")[:1] == "1" and dist.is_available() and dist.is_initialized())
        if backend == "abi" and self._multi:
            self._init_abi_comm(world, rank, group, param[0].device)
        # LRCN_OPT_FUSED_UPDATE: update! writes the next step's shadow weights, the separate shadow pass disappears from the head of the LSTM
        # chain.  Emulated rank of 8 (32 rows): 1.549 -> 1.510 ms/step (two same-box pairs).  At 256 rows it lost 1.4 % early in round 3
        # (7.19 -> 7.29 ms: the fused kernel's tile transposes want LDS and all of update! then runs on the 32 CUs the convolutions leave) and
        # is level since the convolutions got faster (three same-box pairs 6.872 / 6.876 / 6.881 -> 6.869 / 6.854 / 6.842 ms): on everywhere.
        # LRCN_FUSED_UPDATE=0 / 1 forces it.
        # Sharded update (opt-in: shard_adam=True / LRCN_DP_SHARD_ADAM=1; torch backend): per gradient group, reduce-scatter(SUM) -> Adam on
        # this rank's 1/N slice of the flat parameter buffer -> all-gather of the parameters.  Same bytes on the wire as the all-reduce,
        # 1/N of update!'s 1.1 GB of HBM traffic per rank (215 -> ~27 us at 8 ranks).  The parameters move into ONE flat buffer (the
        # caller's list keeps working: its entries are replaced by views of it) and the Adam moments exist only as this rank's slices
        # (optim.m / optim.v are not used).  Never run on more than one GPU: opt-in until it has been.
        if shard_adam is None:
            shard_adam = os.environ.get("LRCN_DP_SHARD_ADAM", "0")[:1] == "1"
        self.shard = (bool(shard_adam) and self.backend == "torch" and hasattr(self.ops, "update_flat") and hasattr(self.ops, "grad_group_wait")
                      and not self.gclip > 0)
        env = os.environ.get("LRCN_FUSED_UPDATE")
        fused = (env[:1] != "0") if env else True
        self._fused = bool(fused and not self.shard and hasattr(self.ops, "set_fused_update"))
        # Sharded update (round 5): the next step's shadow weights of a group are made on the update stream right after that group's
        # all-gather (lrcn_refresh_shadows_group), beside the rest of the backward pass, instead of as one pass at the head of the next
        # lossgradient (63 .. 138 us on the critical path of a 32-row step).  Same contract as the fused update.  LRCN_FUSED_UPDATE=0: off.
        self._shadow_groups = bool(fused and self.shard and hasattr(self.ops, "set_fused_update") and hasattr(self.ops, "refresh_shadows_group"))
        if self._fused or self._shadow_groups:
            self.ops.set_fused_update(True)
        shapes = [tuple(t.shape) for t in param]
        self._W = self._emu_shards or max(world, 1)   # number of parameter shards of the sharded update
        if self.shard:
            align = 4 * self._W  # every rank's slice is a whole number of 16-byte chunks
            dev = param[0].device
            self.flat_param, pviews, self._ranges = flat_model_like(shapes, device=dev, group_align=align)
            for k, v in enumerate(pviews):
                v.copy_(param[k])
                param[k] = v           # the caller's list now refers to the flat buffer
            self.flat_grads, self.grads, _ = flat_model_like(shapes, device=dev, group_align=align)
            n_of = [(b - a) // self._W for a, b in self._ranges]
            self._shapes, self._align = shapes, align
            self._emu_scratch = torch.empty(max(b - a for a, b in self._ranges), device=dev, dtype=torch.float32) if self._emu_shards else None
            self._m = [torch.zeros(n, device=dev, dtype=torch.float32) for n in n_of]
            self._v = [torch.zeros(n, device=dev, dtype=torch.float32) for n in n_of]
            self._gshard = [torch.zeros(n, device=dev, dtype=torch.float32) for n in n_of]
            self._scatter_optim_state()   # a resumed optimizer (optim.t > 0) keeps its moments: this rank's slices of optim.m / optim.v
        else:
            self.flat_grads, self.grads = flat_model_like(shapes, device=param[0].device)
        self.step_no = 0
        self._feat_q = collections.deque()   # (features of an upcoming step, event of the forward that makes them or None), in step order
        self._chunk_no = 0
        self._side = self.ops.side_stream() if hasattr(self.ops, "side_stream") else None
        if self._side is not None and hasattr(self.ops, "set_vgg_wg_cap"):
            env = os.environ.get("LRCN_VGG_WG_CAP")
            cap = int(env) if env is not None else vgg_wg_cap_for(param[0].device, int(rows) if rows else B_global // max(world, 1), int(vgg_chunk))
            self.ops.set_vgg_wg_cap(cap)
            self.vgg_cap = cap   # (bench.py: the CUs left to the LSTM chain = 256 - cap)
        self._feats_buf = [None, None]  # ping-pong outputs of the side-stream VGG (one chunk each)
        # the library's weight-gradient stream must not share a hardware queue with the VGG side stream either (its dW GEMMs would run in
        # order with the convolutions): hand it a probed one.  LRCN_DP_WG_STREAM_PROBE=0: leave the library's own.
        if (self._side is not None and isinstance(self._side, torch.cuda.Stream) and ctx is not None and hasattr(self.ops, "set_wg_stream")
                and os.environ.get("LRCN_DP_WG_STREAM_PROBE", "1")[:1] != "0"):
            self.ops.set_wg_stream(avoid=[self._side, torch.cuda.current_stream(ctx.device)])
        self._bucket_streams = None
        self._abi_stream_set = False
        # Sparse exchange of the embedding gradient (N > 1, torch backend, per-group pipeline): Wembed's 42.6 MB gradient is the group that
        # becomes final LAST, so its all-reduce is the exposed one; what a rank contributes is (T+1) B rows of d(x_lstm) (1.5 MB at 32 rows).
        # The ranks all-gather rows + token ids (12 MB in all at 8 x 32 rows) and every rank sums them in one fixed order.
        # LRCN_DP_SPARSE_EMBED=0 / 1 forces it off / on (1: also on a one-rank group, for tests).
        env_sp = os.environ.get("LRCN_DP_SPARSE_EMBED")
        want = (env_sp[:1] != "0") if env_sp else world > 1
        self._sparse_embed = bool(want and self.backend == "torch" and not self.shard and not self.gclip > 0 and self._multi_or_forced_sparse(env_sp)
                                  and hasattr(self.ops, "embed_grad_from_rows") and hasattr(self.ops, "update_group") and ctx is not None
                                  and self._group_pipeline())   # the per-group [exchange -> Adam] pipeline is where the rows are summed
        if self._sparse_embed:
            W = max(world, 1)
            cap = (ctx.max_T + 1) * min(ctx.max_B, int(rows) if rows else ctx.max_B)   # `rows` = this rank's rows of a training batch, if told
            if W * cap > 8192:
                self._sparse_embed = False   # the ordered sum sorts its keys in one workgroup (8192)
            else:
                dev = param[0].device
                E = ctx.E
                self._emb_rows = torch.zeros(cap * E, device=dev, dtype=torch.float32)
                self._emb_tok = torch.zeros(cap, device=dev, dtype=torch.int32)
                self._emb_rows_all = torch.zeros(W * cap * E, device=dev, dtype=torch.float32)
                self._emb_tok_all = torch.zeros(W * cap, device=dev, dtype=torch.int32)
                self.ops.set_embed_rows_buffer(self._emb_rows, self._emb_tok)
        self._prefetched = None  # (host tensor, staged device crops): the upload started by the previous step's prefetch_img_u8

    # ---- parameter / optimizer state hand-over (checkpoints) ----
    def params_touched(self):
        """The caller wrote the parameter tensors itself since the last step (see the class docstring)."""
        if hasattr(self.ops, "params_touched"):
            self.ops.params_touched()

    def restore(self, arrays=None, adam=None):
        """Load a checkpoint into the trainer's tensors: arrays = the nine parameter arrays (logical reference shapes; None = keep),
        adam = (m arrays, v arrays, t) or None.  Declares the write (params_touched) and re-slices the moments of a sharded update."""
        if arrays is not None:
            for t, a in zip(self.param, arrays):
                if t.numel():
                    t.copy_(L.to_jl(np.asarray(a, np.float32), device=t.device))
            self.params_touched()
        if adam is not None:
            m, v, step = adam
            for k in range(len(self.param)):
                if self.optim.m[k].numel():
                    self.optim.m[k].copy_(L.to_jl(np.asarray(m[k], np.float32), device=self.optim.m[k].device))
                    self.optim.v[k].copy_(L.to_jl(np.asarray(v[k], np.float32), device=self.optim.v[k].device))
            self.optim.t = int(step)
            if self.shard:
                self._scatter_optim_state()

    def _scatter_optim_state(self):
        """Sharded update: this rank's 1/N slices of the optimizer's moments (group-padded flat layout, as the parameters)."""
        W, r = self._W, self.rank
        for src, dst in ((self.optim.m, self._m), (self.optim.v, self._v)):
            flat, views, _ = flat_model_like(self._shapes, device=self.flat_param.device, group_align=self._align)
            for vw, t in zip(views, src):
                if t.numel():
                    vw.copy_(t)
            for k, (a, b) in enumerate(self._ranges):
                n = (b - a) // W
                dst[k].copy_(flat[a + r * n:a + (r + 1) * n])

    def gather_optim_state(self):
        """Sharded update: all-gather the rank-sharded moments back into optim.m / optim.v (every rank), so that a checkpoint written
        from `optim` holds the real state.  A no-op for the replicated update, whose optim.m / optim.v are the state."""
        if not self.shard:
            return
        W = self._W
        coll = self.world > 1
        if self._emu_shards:
            raise L.LrcnError("emulate_shards is a timing aid: there is no whole optimizer state to gather")
        for shards, dst in ((self._m, self.optim.m), (self._v, self.optim.v)):
            flat, views, _ = flat_model_like(self._shapes, device=self.flat_param.device, group_align=self._align)
            for k, (a, b) in enumerate(self._ranges):
                if b == a:
                    continue
                n = (b - a) // W
                if coll:
                    parts = [torch.empty_like(shards[k]) for _ in range(W)]
                    dist.all_gather(parts, shards[k].clone(), group=self.group)
                else:
                    parts = [shards[k]]
                for i, t in enumerate(parts):
                    flat[a + i * n:a + (i + 1) * n].copy_(t)
            for vw, t in zip(views, dst):
                if t.numel():
                    t.copy_(vw)

    def close(self):
        """Give the context back as it was: the fused-update option belongs to the context and would outlive the trainer."""
        if getattr(self, "_sparse_embed", False):
            try:
                self.ops.set_embed_rows_buffer(None, None)
            except Exception:
                pass
            self._sparse_embed = False
        if (getattr(self, "_fused", False) or getattr(self, "_shadow_groups", False)) and hasattr(self.ops, "set_fused_update"):
            try:
                self.ops.set_fused_update(False)
            except Exception:
                pass
            self._fused = self._shadow_groups = False

    def _init_abi_comm(self, world, rank, group, device):
        """Two phases, so that no rank can be left alone inside the collective lrcn_comm_init (= ncclCommInitRank):
        1. every rank probes LOCALLY (librccl loadable from liblrcn_hip, every symbol resolved, no communicator on the context; rank 0
           also makes the unique id) and the ranks all-reduce(MIN) the outcome;
        2. only if every rank passed: the id is broadcast and every rank enters lrcn_comm_init; its outcome is agreed the same way.
        Any failure -> ALL ranks use the torch.distributed collectives instead (still RCCL over xGMI, issued from Python)."""
        ok_local, why = self.ops.comm_probe() if hasattr(self.ops, "comm_probe") else (True, "")
        uid_bytes = None
        if ok_local and rank == 0:
            try:
                uid_bytes = L.comm_unique_id()
            except L.LrcnError as e:
                ok_local, why = False, str(e)
        ok = torch.tensor([1 if ok_local else 0], dtype=torch.int32, device=device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 1:
            uid = torch.zeros(128, dtype=torch.uint8, device=device)
            if rank == 0:
                uid.copy_(torch.frombuffer(bytearray(uid_bytes), dtype=torch.uint8))
            dist.broadcast(uid, 0, group=group)
            try:
                self.ops.comm_init(world, rank, bytes(uid.cpu().numpy().tobytes()))   # the collective: every rank is here
            except L.LrcnError as e:
                ok_local, why = False, str(e)
            ok = torch.tensor([1 if ok_local else 0], dtype=torch.int32, device=device)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=group)
        if int(ok.item()) == 0:
            if hasattr(self.ops, "comm_destroy"):
                self.ops.comm_destroy()
            self.backend = "torch"
            self.backend_note = "C-ABI communicator unavailable (%s)" % (why or "another rank failed")
            if rank == 0:
                import sys
                print("lrcn_amd.dp: %s; using torch.distributed collectives" % self.backend_note, file=sys.stderr)

    def _multi_or_forced_sparse(self, env_sp):
        return self._multi or (env_sp is not None and env_sp[:1] == "1")

    def _gather_embed_rows(self, M):
        """All-gather of this step's embedding-gradient rows and token ids over the ranks (rank order) -> (rows_all, tok_all, n_rows)."""
        E = self.ctx.E
        mine_r, mine_t = self._emb_rows[:M * E], self._emb_tok[:M]
        if not (dist.is_available() and dist.is_initialized()):
            return mine_r, mine_t, M
        W = dist.get_world_size(self.group)   # the group's own size (a test may tell the trainer another `world` than its group has)
        all_r, all_t = self._emb_rows_all[:W * M * E], self._emb_tok_all[:W * M]
        if dist.get_backend(self.group) == "nccl":
            dist.all_gather_into_tensor(all_r, mine_r, group=self.group)
            dist.all_gather_into_tensor(all_t, mine_t, group=self.group)
        else:   # gloo (ranks sharing one GPU, CPU tests): through host memory
            hr, ht = mine_r.cpu(), mine_t.cpu()
            pr, pt = [torch.empty_like(hr) for _ in range(W)], [torch.empty_like(ht) for _ in range(W)]
            dist.all_gather(pr, hr, group=self.group)
            dist.all_gather(pt, ht, group=self.group)
            all_r.copy_(torch.cat(pr))
            all_t.copy_(torch.cat(pt))
        return all_r, all_t, W * M

    def _make_update_streams(self):
        try:
            # Round 5: the update stream must share its hardware queue with NONE of the three other chains of a step -- the VGG side stream,
            # the main stream (the rest of the backward pass runs there while a group is exchanged and updated) and the library's
            # weight-gradient stream.  Until round 5 only the side stream was avoided: in about one process out of three the update chain
            # landed on the main stream's queue and ran IN ORDER with the backward pass (emulated sharded rank of 8: 1.26 instead of 1.18 ms).
            avoid = [self._side]
            if self.ctx is not None and torch.cuda.is_available():
                avoid.append(torch.cuda.current_stream(self.ctx.device))
            avoid.append(getattr(self.ops, "_wg_stream", None))
            streams = self.ops.make_streams(len(GRAD_GROUPS), avoid=[a for a in avoid if a is not None])
        except TypeError:   # stand-in ops of the CPU tests
            return self.ops.make_streams(len(GRAD_GROUPS))
        self._bucket_streams = streams
        self._keep_collectives_off_the_vgg_queue()
        return streams

    def _keep_collectives_off_the_vgg_queue(self, rounds=4):
        """torch.distributed runs an RCCL collective on a stream of ITS OWN choosing (a pool stream), which may share a hardware queue with
        the VGG side stream -- every all-reduce would then wait for the convolutions queued before it (with several batches per forward: for
        milliseconds).  That stream cannot be asked for, so it is MEASURED, with the collective the step issues: a ~2 ms sleep kernel on the
        side stream, then a small all-reduce behind the update stream; if it only completes when the sleep does, this rank takes another
        side stream (one that shares no queue with the update stream either) and the ranks try again -- the same number of rounds on every
        rank (the outcome is agreed by an all-reduce), so the collectives stay matched.  RCCL over a real N > 1 group only."""
        knob = os.environ.get("LRCN_DP_QUEUE_PROBE", "1")   # 0 = never; force = also on a one-rank group (tests)
        if knob[:1] == "0" or not ((self.world > 1 or knob == "force") and self._side is not None and dist.is_available() and dist.is_initialized()
                                   and dist.get_backend(self.group) == "nccl" and isinstance(self._side, torch.cuda.Stream)):
            return
        dev = self.param[0].device
        upd = self._bucket_streams[0]
        t = torch.zeros(1024, device=dev)
        self.queue_probe = []
        for _ in range(rounds):
            torch.cuda.synchronize(dev)
            e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            with torch.cuda.stream(self._side):
                e0.record()
                torch.cuda._sleep(4_000_000)
                e1.record()
            with torch.cuda.stream(upd):
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=True).wait()
                e2.record()
            torch.cuda.synchronize(dev)
            late = e0.elapsed_time(e2) > 0.5 * e0.elapsed_time(e1)
            flag = torch.tensor([1.0 if late else 0.0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
            self.queue_probe.append(bool(late))
            if float(flag.item()) == 0.0:
                return
            if late:   # this rank's collective stream sits behind the side stream: move the VGG forward to another queue
                prio = int(os.environ.get("LRCN_VGG_STREAM_PRIO", "0"))
                cand = independent_stream(dev, [upd]) if prio == 0 else None
                self._side = cand if cand is not None else torch.cuda.Stream(device=dev, priority=prio)

    def _group_slices(self):
        """Flat-buffer ranges of the gradient groups, in the order lossgradient finalises them."""
        sizes = [g.numel() for g in self.grads]
        offs = np.concatenate([[0], np.cumsum(sizes)])
        return [(int(offs[min(grp)]), int(offs[max(grp) + 1])) for grp in GRAD_GROUPS]

    def _allreduce_async(self):
        """-> list of pending works.  LRCN_DP_BUCKETS=0: one all-reduce of the whole flat buffer after lossgradient.
        Default: one all-reduce per gradient group, each started as soon as ITS gradients are final (an event recorded by
        the library in the middle of the backward pass), so most of the 159 MB move while the rest of the backward runs."""
        if self.world == 1:
            return []
        if os.environ.get("LRCN_DP_BUCKETS", "1")[:1] == "0":
            return [dist.all_reduce(self.flat_grads, op=dist.ReduceOp.SUM, group=self.group, async_op=True)]
        works = []
        gpu = hasattr(self.ops, "grad_group_wait") and hasattr(self.ops, "make_streams")
        if gpu and self._bucket_streams is None:
            self._bucket_streams = self._make_update_streams()
        for k, (a, b) in enumerate(self._group_slices()):
            if a == b:
                continue  # LRCN-1f has no W2 / b2: nothing to exchange for that group
            chunk = self.flat_grads[a:b]
            if gpu:
                s = self._bucket_streams[k]
                self.ops.grad_group_wait(k, s)  # s waits for the group's event; RCCL's stream then waits for s
                with self.ops.stream_ctx(s):
                    works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            else:
                works.append(dist.all_reduce(chunk, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        return works

    def vgg(self, img_u8):
        return self.ops.vgg(img_u8, normalize=True) if self.normalize_features else self.ops.vgg(img_u8)

    def _vgg_blocks(self, img_u8, rows, out=None):
        """VGG forward of the crops of m = N / rows batches -> m feature blocks (rows x 4096 each)."""
        if img_u8.shape[0] == rows or not hasattr(self.ops, "vgg_blocks"):
            if img_u8.shape[0] != rows:
                raise L.LrcnError("crops for %d rows given, the step has %d (these ops cannot run several batches in one forward)" % (img_u8.shape[0], rows))
            return [self.ops.vgg(img_u8, feats=out, normalize=True) if self.normalize_features else self.ops.vgg(img_u8, feats=out)]
        if img_u8.shape[0] % rows:
            raise L.LrcnError("crops for %d rows are not a whole number of %d-row batches" % (img_u8.shape[0], rows))
        return self.ops.vgg_blocks(img_u8, rows, feats=out, normalize=self.normalize_features)

    def _vgg_on_side_stream(self, img_u8, rows):
        """Issue the VGG forward of the next chunk of crops (one or several batches) on the side stream into a ping-pong buffer and queue
        its feature blocks; the main stream waits on the event only when it consumes the first of them."""
        main = torch.cuda.current_stream(self.ctx.device)
        k = self._chunk_no & 1
        self._chunk_no += 1
        n = img_u8.shape[0] * L.CNNOUT
        if self._feats_buf[k] is None or self._feats_buf[k].numel() != n:
            self._feats_buf[k] = (torch.empty(n, device=self.param[0].device, dtype=torch.float32) if img_u8.shape[0] != rows
                                  else L.jl_empty(rows, L.CNNOUT))
        # The crops (and the previous consumers of this buffer) are ordered before the forward.  Measured and not kept (round 3): three
        # rotating buffers with the side stream waiting only for the step TWO back, so that the VGG forward may run a step ahead of the
        # LSTM chain -- 7.115 / 7.117 -> 7.122 / 7.132 ms per step: both chains are as long as each other at 256 rows, nothing to run ahead of.
        self._side.wait_stream(main)
        self.ctx.use_stream(self._side)
        try:
            blocks = self._vgg_blocks(img_u8, rows, out=self._feats_buf[k])
        finally:
            self.ctx.use_stream(main)
        if torch.is_tensor(img_u8) and img_u8.is_cuda:
            # the crops were allocated on the main stream but are read by the side stream: tell the caching allocator, so a
            # caller that drops them right after step() cannot have the block recycled under the running convolution
            img_u8.record_stream(self._side)   # (staged crops live in the library's own buffers: nothing to record)
        ev = torch.cuda.Event()
        ev.record(self._side)
        for b in blocks:
            self._feat_q.append((b, ev))

    def _device_crops(self, img):
        """Device crops for a batch that may still be in (pinned) host memory: the staged copy that the previous step's prefetch_img_u8
        started, if `img` is that very tensor; otherwise the upload starts now (correct, but on the critical path of this batch)."""
        if not (torch.is_tensor(img) and not img.is_cuda) or not hasattr(self.ops, "upload"):
            return img
        if self._prefetched is not None and self._prefetched[0] is img:
            staged = self._prefetched[1]
            self._prefetched = None
            return staged
        return self.ops.upload(img)

    def step(self, img_u8, tokens, next_img_u8=None, feats=None, prefetch_img_u8=None):
        """One synchronous-SGD step on this rank's shard.  img_u8: this rank's uint8 crops (or feats given);
        next_img_u8: the crops of the NEXT step(s), whose VGG forward runs beside this step's LSTM work and all-reduce.
        img_u8 is IGNORED when an earlier step() already produced this batch's features through its next_img_u8.

        Several batches per forward.  next_img_u8 may hold the crops of m >= 1 consecutive steps (m * B rows): the frozen extractor does
        not depend on the parameters, so a rank whose own batch is small (32 rows of 256 on 8 GPUs) runs the convolutions of its next m
        steps in ONE forward, at a large batch's efficiency, beside m LSTM steps, and consumes one feature block per step.  The forward of
        a chunk is issued when the blocks in hand no longer cover m steps (i.e. with the first block of the previous chunk); until then
        next_img_u8 is ignored.  Returns True when this call consumed next_img_u8 -- the caller then moves on to the following chunk.

        Crops may be CPU tensors (pinned for a true asynchronous copy): the reference uploads every batch's inputs inside its loop
        (lrcn.jl:369-376); here the copy runs on the library's copy stream.  prefetch_img_u8: the chunk AFTER next_img_u8 (the tensor
        that will be passed as next_img_u8 once this one has been consumed): its upload starts when next_img_u8 is consumed, a whole
        chunk before the VGG forward that reads it, so the forward never waits for PCIe.  A host buffer handed over must stay unchanged
        until its forward has been queued AND the copy has run (lrcn.upload_wait) -- a loader rotates at least three pinned buffers."""
        rows = int(tokens.shape[1])
        self._cur_M = (int(tokens.shape[0]) + 1) * rows
        if feats is None:
            if self._feat_q:
                feats, ev = self._feat_q.popleft()
                if ev is not None:
                    torch.cuda.current_stream(self.ctx.device).wait_event(ev)
            else:
                blocks = self._vgg_blocks(self._device_crops(img_u8), rows)
                feats = blocks[0]
                self._feat_q.extend((b, None) for b in blocks[1:])
        self.step_no += 1
        consumed, deferred = False, None
        if next_img_u8 is not None and len(self._feat_q) < max(1, next_img_u8.shape[0] // rows):
            consumed = True
            nxt = self._device_crops(next_img_u8)
            if self._side is not None:
                self._vgg_on_side_stream(nxt, rows)  # concurrent with everything below
            else:
                deferred = nxt   # in-order variant: issued after lossgradient (overlaps the gradient exchange only; frozen VGG)
            if prefetch_img_u8 is not None and hasattr(self.ops, "upload") and torch.is_tensor(prefetch_img_u8) and not prefetch_img_u8.is_cuda:
                # after the forward above has been queued: that forward released (in issue order) the staging buffer this upload goes into
                self._prefetched = (prefetch_img_u8, self.ops.upload(prefetch_img_u8))

        def inorder_vgg():
            if deferred is not None:
                self._feat_q.extend((b, None) for b in self._vgg_blocks(deferred, rows))

        # rank-dependent dropout stream: masks differ per shard like rows of one big batch would
        seed = (self.seed + self.step_no) * 65536 + self.rank
        if self.backend == "abi" and self._multi:
            if not self._abi_stream_set and hasattr(self.ops, "comm_set_stream"):
                self.ops.comm_set_stream(avoid=[self._side])
                self._abi_stream_set = True
            # one C call: lossgradient + per-group [all-reduce over xGMI -> Adam] on the library's own streams
            self.ops.train_step_dp(self.param, self.grads, self.optim, feats, tokens, self.B_global, self.pdrop, seed)
            inorder_vgg()
            return consumed
        self.ops.lossgradient(self.param, feats, tokens, self.B_global, self.pdrop, seed, self.grads)
        if self.gclip > 0:
            if self.world > 1:
                dist.all_reduce(self.flat_grads, op=dist.ReduceOp.SUM, group=self.group)
            gn = float(torch.linalg.vector_norm(self.flat_grads))
            if gn > self.gclip:
                self.flat_grads.mul_(self.gclip / gn)
            inorder_vgg()
            self.ops.update(self.param, self.grads, self.optim)
            return consumed
        if self.shard:
            self._reduce_scatter_update_gather()
            inorder_vgg()
            return consumed
        if self._group_pipeline():
            # per gradient group, on its own stream: [wait for the group's event] -> [all-reduce] -> [Adam of that group], all
            # while the rest of the backward pass runs (it reads the bf16/f32 shadows, never the f32 parameters)
            self._reduce_and_update_groups()
            inorder_vgg()
            return consumed
        works = self._allreduce_async()
        inorder_vgg()
        for w in works:
            w.wait()  # the compute stream waits for RCCL
        self.ops.update(self.param, self.grads, self.optim)
        return consumed

    def _group_pipeline(self):
        """Per-group [all-reduce -> Adam] needs the device-side gradient-group events and the per-group update entry point
        (the HIP ops).  Default for world > 1: the Adam of a group (HBM-bound) runs while later groups are still being
        reduced, instead of one 0.19-ms launch after the last bucket.  With one rank there is no exchange to hide it behind
        and five launches measure 1 % slower than one (B=32: 1.728 vs 1.708 ms/step), so the single launch stays.
        LRCN_DP_GROUP_ADAM=0/1 forces either; LRCN_DP_BUCKETS=0 selects the single all-reduce + single Adam."""
        if not (hasattr(self.ops, "grad_group_wait") and hasattr(self.ops, "update_group") and hasattr(self.ops, "make_streams")):
            return False
        if os.environ.get("LRCN_DP_BUCKETS", "1")[:1] == "0":
            return False
        env = os.environ.get("LRCN_DP_GROUP_ADAM")
        return env[:1] != "0" if env else self._multi

    def _reduce_and_update_groups(self):
        if self._bucket_streams is None:
            self._bucket_streams = self._make_update_streams()
        self.optim.t += 1
        for k, (a, b) in enumerate(self._group_slices()):
            s = self._bucket_streams[k]
            self.ops.grad_group_wait(k, s)  # s waits for the group's event recorded inside lossgradient
            with self.ops.stream_ctx(s):
                if k == 4 and self._sparse_embed:
                    # the embedding gradient travels as rows: all-gather, then the ordered per-token sum into the dense gradient (every rank)
                    rows_all, tok_all, n = self._gather_embed_rows(self._cur_M)
                    self.ops.embed_grad_from_rows(rows_all, tok_all, n, self.grads[6], s)
                elif self._multi and b > a:
                    dist.all_reduce(self.flat_grads[a:b], op=dist.ReduceOp.SUM, group=self.group, async_op=True).wait()  # s waits for RCCL
                self.ops.update_group(self.param, self.grads, self.optim, k, s)
        self.ops.join(self._bucket_streams)

    def _reduce_scatter_update_gather(self):
        """The sharded form of _reduce_and_update_groups: per group, on its own stream, [wait for the group's gradients] ->
        [reduce-scatter(SUM)] -> [Adam on this rank's slice] -> [all-gather of the parameters]; then one join."""
        if self._bucket_streams is None:
            self._bucket_streams = self._make_update_streams()
        tail = os.environ.get("LRCN_DP_DEBUG_TAIL")   # development: how long the update chain runs past the end of the backward pass
        if tail:
            e_bwd = torch.cuda.Event(enable_timing=True)
            e_bwd.record(torch.cuda.current_stream(self.ctx.device))
        self.optim.t += 1
        W, r = self._W, self.rank
        # a one-rank process group given explicitly (tests on a one-GPU box): the collectives are issued all the same
        coll = self.world > 1 or (self.group is not None and dist.is_initialized())
        nccl = coll and dist.get_backend(self.group) == "nccl"
        for k, (a, b) in enumerate(self._ranges):
            s = self._bucket_streams[k]
            self.ops.grad_group_wait(k, s)
            if b == a:
                if self._shadow_groups:
                    self.ops.refresh_shadows_group(self.param, k, s)   # an empty group still counts towards "all five refreshed"
                continue  # LRCN-1f has no W2 / b2 group
            n = (b - a) // W
            mine = self.flat_param[a + r * n:a + (r + 1) * n]
            with self.ops.stream_ctx(s):
                g_all = self.flat_grads[a:b]
                if self._emu_shards:   # stub of the reduce-scatter: this rank's slice arrives (n floats)
                    self._gshard[k].copy_(g_all[:n])
                elif not coll:
                    self._gshard[k].copy_(g_all)
                elif nccl:
                    dist.reduce_scatter_tensor(self._gshard[k], g_all, op=dist.ReduceOp.SUM, group=self.group)
                else:  # gloo has no reduce-scatter: all-reduce, then keep this rank's slice (CPU tests of the indexing)
                    tmp = g_all.clone()
                    dist.all_reduce(tmp, op=dist.ReduceOp.SUM, group=self.group)
                    self._gshard[k].copy_(tmp[r * n:(r + 1) * n])
                self.ops.update_flat(mine, self._gshard[k], self._m[k], self._v[k], self.optim, s)
                if self._emu_shards:   # stub of the all-gather: (N-1) n floats arrive (device copy of as many bytes; values discarded)
                    self._emu_scratch[:b - a - n].copy_(self.flat_param[a + n:b])
                elif coll:
                    if nccl:
                        dist.all_gather_into_tensor(self.flat_param[a:b], mine, group=self.group)
                    else:
                        parts = [torch.empty_like(mine) for _ in range(W)]
                        dist.all_gather(parts, mine.clone(), group=self.group)
                        for i, t in enumerate(parts):
                            self.flat_param[a + i * n:a + (i + 1) * n].copy_(t)
                if self._shadow_groups:   # the group's parameters are final on this stream: its shadows for the next step, beside the backward pass
                    self.ops.refresh_shadows_group(self.param, k, s)
        if tail:
            e_upd = torch.cuda.Event(enable_timing=True)
            e_upd.record(self._bucket_streams[-1])
            self._tail_events = getattr(self, "_tail_events", []) + [(e_bwd, e_upd)]
        self.ops.join(self._bucket_streams)

    def describe(self):
        """What this trainer's step is made of (reported in bench.py's line: rccl.pipeline)."""
        buckets = os.environ.get("LRCN_DP_BUCKETS", "1")[:1] != "0"
        return {"backend": self.backend, "update": "sharded" if self.shard else "replicated", "fused_update": bool(self._fused),
                "shadows_by_group": bool(getattr(self, "_shadow_groups", False)),
                "per_group_pipeline": bool(self.shard or (self.backend == "abi" and self._multi) or self._group_pipeline()),
                "one_allreduce_of_the_flat_buffer": bool(self.world > 1 and not buckets and not self.shard),
                "sparse_embedding_exchange": bool(getattr(self, "_sparse_embed", False)), "vgg_side_stream": self._side is not None,
                "queue_probe": getattr(self, "queue_probe", None), "backend_note": self.backend_note or None,
                "update_stream_on_its_own_queue": getattr(self.ops, "update_stream_probed", None)}

    # ---- first-contact self-check of an N-rank job (bench.py and tools/lrcn.py run it once, before anything is timed) ----
    def _dist_on(self):
        return dist.is_available() and dist.is_initialized()

    def _gather_cat(self, t):
        """All-gather of equal-sized contiguous tensors -> one tensor, rank blocks in rank order (RCCL: on the device; gloo: through host)."""
        if not self._dist_on():
            return t.clone()
        W = dist.get_world_size(self.group)
        if dist.get_backend(self.group) == "nccl":
            out = torch.empty((W,) + tuple(t.shape), device=t.device, dtype=t.dtype)
            dist.all_gather_into_tensor(out, t.contiguous(), group=self.group)
            return out.view((W * t.shape[0],) + tuple(t.shape[1:]))
        h = t.detach().cpu().contiguous()
        parts = [torch.empty_like(h) for _ in range(W)]
        dist.all_gather(parts, h, group=self.group)
        return torch.cat(parts).to(t.device)

    def param_checksum(self):
        """An exact, order-independent fingerprint of the nine parameter tensors: the int64 sum of their float32 bit patterns."""
        tot = 0
        for t in self.param:
            if t.numel():
                tot += int(t.contiguous().view(-1).view(torch.int32).sum(dtype=torch.int64).item()) if t.is_contiguous() else \
                    int(t.permute(*reversed(range(t.dim()))).contiguous().view(-1).view(torch.int32).sum(dtype=torch.int64).item())
        return tot

    def check_replicas(self):
        """-> (identical on every rank?, the checksums in rank order).  Synchronous SGD with a summed gradient keeps replicas bit-identical:
        every rank receives the same reduced bytes and runs the same Adam arithmetic."""
        dev = self.param[0].device
        mine = torch.tensor([self.param_checksum()], dtype=torch.int64, device=dev)
        allc = [int(x) for x in self._gather_cat(mine).cpu().tolist()]
        return all(c == allc[0] for c in allc), allc

    def check_sparse_embed(self, feats, tokens):
        """The sparse exchange of the embedding gradient against the dense one ON THE SAME STEP (same features, tokens and dropout seed):
        lossgradient with the rows buffer registered -> all-gather -> ordered per-token sum, versus lossgradient scattering into the dense
        gradient -> all-reduce(SUM).  -> max |sparse - dense| / max |dense|, or None when the sparse path is not in use."""
        if not getattr(self, "_sparse_embed", False):
            return None
        seed = (self.seed + 1) * 65536 + self.rank
        M = (int(tokens.shape[0]) + 1) * int(tokens.shape[1])
        dev = self.param[0].device
        self.ops.set_embed_rows_buffer(None, None)
        try:
            self.ops.lossgradient(self.param, feats, tokens, self.B_global, self.pdrop, seed, self.grads)
            torch.cuda.synchronize(dev)
            dense = self.grads[6].clone()
        finally:
            self.ops.set_embed_rows_buffer(self._emb_rows, self._emb_tok)
        if self._dist_on():
            flat = dense.permute(1, 0).contiguous() if not dense.is_contiguous() else dense
            if dist.get_backend(self.group) == "nccl":
                dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
            else:
                h = flat.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
                flat.copy_(h)
            dense_sum = flat.permute(1, 0) if flat is not dense else flat
        else:
            dense_sum = dense
        self.ops.lossgradient(self.param, feats, tokens, self.B_global, self.pdrop, seed, self.grads)
        torch.cuda.synchronize(dev)
        rows_all, tok_all, n = self._gather_embed_rows(M)
        s = torch.cuda.current_stream(dev)
        self.ops.embed_grad_from_rows(rows_all, tok_all, n, self.grads[6], s)
        torch.cuda.synchronize(dev)
        return float((self.grads[6] - dense_sum).abs().max() / (dense_sum.abs().max() + 1e-30))

    def self_check(self, feats, tokens_global):
        """Before timing, on real inputs of step 1 (feats: this rank's rows of the first batch, tokens_global: [T][B_global] on the host):
        * the world the communicator reports, and the world an all-reduce of ones measures;
        * sharding and the loss reduction: the mean over ranks of the shard losses equals rank 0's recomputation, shard by shard, on the
          features of ALL rows (all-gathered) and its own copy of the global tokens -- a rank that took the wrong rows, or a reduction
          that lost a rank, shows here;
        * sparse = dense embedding gradient on the same step (check_sparse_embed);
        * the parameters are bit-identical on every rank before the first step.
        -> dict (every rank computes the same values).  The caller decides what a violation means (bench.py: leave the rung)."""
        dev = self.param[0].device
        W = dist.get_world_size(self.group) if self._dist_on() else 1
        ones = torch.ones(1, device=dev)
        if self._dist_on():
            dist.all_reduce(ones, op=dist.ReduceOp.SUM, group=self.group)
        B = int(feats.shape[0])
        rows = shard_rows(self.B_global, W, self.rank) if W * B == self.B_global else slice(0, B)
        toks = np.ascontiguousarray(np.asarray(tokens_global)[:, rows])
        mine = torch.tensor([float(self.ops.loss(self.param, feats, toks))], dtype=torch.float64, device=dev)
        if self._dist_on():
            dist.all_reduce(mine, op=dist.ReduceOp.SUM, group=self.group)
        mean_of_ranks = float(mine.item()) / W
        # feats is column-major (B x 4096 with the row index fastest): its transpose is the contiguous tensor that travels
        ft = feats.permute(1, 0) if not feats.is_contiguous() else feats
        all_ft = self._gather_cat(ft.contiguous())
        recomputed = 0.0
        for r in range(W):
            blk = all_ft[r * ft.shape[0]:(r + 1) * ft.shape[0]]
            fr = blk.permute(1, 0) if not feats.is_contiguous() else blk
            rr = shard_rows(self.B_global, W, r) if W * B == self.B_global else slice(0, B)
            recomputed += float(self.ops.loss(self.param, fr, np.ascontiguousarray(np.asarray(tokens_global)[:, rr]))) / W
        same, sums = self.check_replicas()
        return {"world_from_communicator": W, "world_measured_by_allreduce": int(round(float(ones.item()))),
                "loss_mean_over_ranks": mean_of_ranks, "loss_rank0_recomputed_on_all_rows": recomputed,
                "loss_rel_diff": abs(mean_of_ranks - recomputed) / (abs(recomputed) + 1e-300),
                "sparse_vs_dense_embed_grad_rel": self.check_sparse_embed(feats, toks),
                "params_identical_before_step_1": bool(same)}

    def loss_value(self):
        """Global loss of the last step: sum over ranks of the locally normalised partial losses."""
        v = torch.tensor([self.ops.last_loss()], device=self.param[0].device, dtype=torch.float64)
        if self.world > 1:
            dist.all_reduce(v, group=self.group)
        return float(v.item())
