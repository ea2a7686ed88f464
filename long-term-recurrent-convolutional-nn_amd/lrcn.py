"""Host-side mirror of the reference's function surface (lrcn.jl) over the HIP C ABI (include/lrcn.h).

Julia is not available in this image, so the host language above the C ABI is Python; names, argument meaning and
error behaviour follow lrcn.jl so that a test written against the reference reads the same here:

    initweights (lrcn.jl:489)   initstate (:512)   lstm (:528)   lrcn (:540)   loss (:553)   lossgradient (:583)
    initparams / update! (:399, :394)   train1's body -> train_step (:369-394)   average_loss's body -> loss(pdrop=0)
    generate / beam_search (:585-678)   get_convnet -> convnet (:733)   read_image_data's arithmetic (:766-772)

Arrays are torch CUDA tensors holding the reference's COLUMN-MAJOR memory: a Julia R x C matrix is a tensor of
shape (R, C) with strides (1, R) (see `jl_empty`), so `.data_ptr()` is exactly what a Julia `ccall` would pass.
torch is used for device memory, streams and (in dp.py) torch.distributed only -- all arithmetic is in
liblrcn_hip.so.  Token ids are 0-based at this layer (eos=0, bos=1, unk=2).
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import BOS, CNNOUT, EOS, LRCN_BF16, LRCN_F32, LRCN_FP8, UNK, LrcnError  # noqa: F401

PARAM_NAMES = ("W1", "b1", "W2", "b2", "Wproj", "Wcnn", "Wembed", "Wout", "bout")
VGG_COUT = (64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512)
VGG_MEAN = (123.68, 116.779, 103.939)  # per-channel mean of the VGG-16 averageImage (lrcn.jl:113)


# ------------------------------------------------------------------------------------------------ arrays
def jl_empty(*dims, device="cuda", dtype=torch.float32):
    """Uninitialised tensor of Julia shape `dims` in column-major memory (first index fastest)."""
    t = torch.empty(tuple(reversed(dims)), device=device, dtype=dtype)
    return t.permute(*reversed(range(len(dims))))


def jl_zeros(*dims, device="cuda", dtype=torch.float32):
    t = jl_empty(*dims, device=device, dtype=dtype)
    t.zero_()
    return t


def to_jl(a, device="cuda"):
    """numpy / tensor of logical shape dims -> column-major device tensor of the same logical shape."""
    a = torch.as_tensor(np.asarray(a, dtype=np.float32) if not torch.is_tensor(a) else a)
    out = jl_empty(*a.shape, device=device, dtype=torch.float32)
    out.copy_(a)
    return out


def from_jl(t):
    return t.detach().cpu().numpy()


def _is_jl(t):
    n = t.dim()
    if n == 0 or t.numel() == 0:
        return True
    return t.permute(*reversed(range(n))).is_contiguous()


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise LrcnError("expected a CUDA (HIP) tensor")
    if not _is_jl(t):
        raise LrcnError("tensor is not in the reference's column-major memory order (use to_jl/jl_empty)")
    return C.c_void_p(t.data_ptr())


def _p9(ts):
    if len(ts) != 9:
        raise LrcnError("model must have 9 tensors (lrcn.jl:492)")
    for t in ts:
        if t.dtype != torch.float32:
            raise LrcnError("model tensors must be float32")
    return _lib.P9(*[_ptr(t).value for t in ts])


def param_shapes(E, H1, H2, V, n_layers=2):
    h = (H2 + 1) // 2
    if n_layers == 1:  # LRCN-1f (include/lrcn.h, lrcn_config.n_layers): no W2 / b2 / Wproj
        return [(E + h + H1, 4 * H1), (1, 4 * H1), (0, 0), (0, 0), (0, 0), (CNNOUT, h), (V, E), (H2, V), (1, V)]
    return [(E + H1, 4 * H1), (1, 4 * H1), (H2 + H2, 4 * H2), (1, 4 * H2), (H1, h), (CNNOUT, h), (V, E), (H2, V), (1, V)]


# ------------------------------------------------------------------------------------------------ context
class Context:
    """One lrcn_ctx (one device). Owns scratch only; the caller owns models, gradients, optimizer state."""

    def __init__(self, E, H1, H2, V, max_B, max_T=_lib.MAX_T, lstm_dtype=LRCN_F32, vgg_dtype=LRCN_F32, max_images=0,
                 device=None, n_layers=2):
        if not torch.cuda.is_available():
            raise LrcnError("no MI355X visible: liblrcn_hip has no CPU path")
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.E, self.H1, self.H2, self.V = E, H1, H2, V
        self.h = H2 // 2
        self.max_B, self.max_T = max_B, max_T
        self.lstm_dtype, self.vgg_dtype = lstm_dtype, vgg_dtype
        self.n_layers = n_layers
        cfg = _lib.Config(self.device, E, H1, H2, V, max_B, max_T, lstm_dtype, vgg_dtype, max_images, n_layers)
        h = C.c_void_p()
        rc = _lib.lib().lrcn_create(C.byref(cfg), C.byref(h))
        _lib.check(None, rc)
        self._h = h
        self._stream = None
        self.use_stream(torch.cuda.current_stream(self.device))

    def use_stream(self, stream):
        self._stream = stream
        _lib.check(self._h, _lib.lib().lrcn_set_stream(self._h, C.c_void_p(stream.cuda_stream)))

    def sync(self):
        _lib.check(self._h, _lib.lib().lrcn_sync(self._h))

    def set_option(self, option, value):
        """lrcn_set_option: _lib.LRCN_OPT_FUSED_UPDATE / LRCN_OPT_DETERMINISTIC / LRCN_OPT_CONV_CHUNK_BYTES (include/lrcn.h)."""
        self._call("lrcn_set_option", int(option), int(value))

    def params_touched(self):
        """The caller wrote parameter arrays itself (checkpoint load, clipping ...): required under LRCN_OPT_FUSED_UPDATE."""
        self._call("lrcn_params_touched")

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.lib().lrcn_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _call(self, name, *args):
        _lib.check(self._h, getattr(_lib.lib(), name)(self._h, *args))


def _dropout(pdrop, seed, mask1, mask2):
    if (pdrop is None or pdrop == 0) and mask1 is None:
        return None, None
    d = _lib.Dropout(float(pdrop or 0.0), int(seed or 0), None, None)
    keep = None
    if mask1 is not None:
        # masks: (T+1, B, E) / (T+1, B, H2) logical arrays; the ABI wants (T+1) column-major blocks (LRCN-1f: mask1 only, (T+1, B, E+h))
        m1 = torch.as_tensor(np.ascontiguousarray(np.transpose(np.asarray(mask1, np.float32), (0, 2, 1)))).cuda()
        d.mask1 = m1.data_ptr()
        keep = [m1]
        if mask2 is not None:
            m2 = torch.as_tensor(np.ascontiguousarray(np.transpose(np.asarray(mask2, np.float32), (0, 2, 1)))).cuda()
            d.mask2 = m2.data_ptr()
            keep.append(m2)
    return d, keep


def _tokens(tokens, device):
    t = torch.as_tensor(np.ascontiguousarray(np.asarray(tokens, dtype=np.int32)) if not torch.is_tensor(tokens) else tokens)
    if t.dim() != 2:
        raise LrcnError("tokens must be [T][B]")
    return t.to(device=device, dtype=torch.int32).contiguous()


# ------------------------------------------------------------------------------------------------ model
def initweights(ctx, seed=42):
    """initweights(atype, hidden, vocab, embed) (lrcn.jl:489-510) -> list of 9 column-major tensors."""
    model = [jl_empty(*s) for s in param_shapes(ctx.E, ctx.H1, ctx.H2, ctx.V, ctx.n_layers)]
    ctx._call("lrcn_init_weights", _p9(model), C.c_uint64(seed))
    return model


def model_from_arrays(arrays):
    """dict name->array or list of 9 arrays (logical reference shapes) -> device model."""
    if isinstance(arrays, dict):
        arrays = [arrays[n] for n in PARAM_NAMES]
    return [to_jl(a) for a in arrays]


def zeros_like_model(model):
    out = []
    for t in model:
        z = jl_empty(*t.shape)
        z.zero_()
        out.append(z)
    return out


def initstate(ctx, batch):
    """initstate(model, batch) (lrcn.jl:512-526): zero (B x H) hidden/cell per layer -- without the reference's
    spurious third layer and without aliasing all entries to one array."""
    if ctx.n_layers == 1:
        return [jl_zeros(batch, ctx.H1), jl_zeros(batch, ctx.H1)]
    return [jl_zeros(batch, ctx.H1), jl_zeros(batch, ctx.H1), jl_zeros(batch, ctx.H2), jl_zeros(batch, ctx.H2)]


def lstm(ctx, weight, bias, hidden, cell, input):
    """lstm(weight,bias,hidden,cell,input) (lrcn.jl:528-538) -> (hidden, cell)."""
    B, X = input.shape
    H = hidden.shape[1]
    h_out, c_out = jl_empty(B, H), jl_empty(B, H)
    ctx._call("lrcn_lstm", _ptr(weight), _ptr(bias), X, H, B, _ptr(input), _ptr(hidden), _ptr(cell), _ptr(h_out), _ptr(c_out))
    return h_out, c_out


def lrcn(ctx, w, s, x_cnn, x_lstm, mask1=None, mask2=None):
    """lrcn(w, s, x_cnn, x_lstm; pdrop) (lrcn.jl:540-551): one timestep; mutates s like the reference; returns logits.
    Dropout enters as explicit multiplier arrays (B x E, B x H2) so a step is reproducible."""
    B = x_lstm.shape[0]
    logits = jl_empty(B, ctx.V)
    st = _lib.P4(*([_ptr(t).value for t in s] + [None] * (4 - len(s))))
    ctx._call("lrcn_step", _p9(w), st, B, _ptr(x_cnn), _ptr(x_lstm), _ptr(mask1), _ptr(mask2), _ptr(logits))
    return logits


def loss(ctx, param, feats, tokens, norm_B=None, pdrop=0.0, seed=0, mask1=None, mask2=None):
    """loss(param,state,input,sequence,range; pdrop) (lrcn.jl:553-581).  feats: B x 4096 (column-major);
    tokens: [T][B] = sequence[range]; norm_B = the reference's global `batchsize` (default B)."""
    tok = _tokens(tokens, feats.device)
    T, B = tok.shape
    d, keep = _dropout(pdrop, seed, mask1, mask2)
    out = C.c_double()
    ctx._call("lrcn_loss", _p9(param), _ptr(feats), C.c_void_p(tok.data_ptr()), T, B, norm_B or B,
              C.byref(d) if d else None, C.byref(out))
    del keep
    return out.value


def lossgradient(ctx, param, feats, tokens, norm_B=None, pdrop=0.0, seed=0, mask1=None, mask2=None, grads=None,
                 want_loss=True):
    """lossgradient = grad(loss) (lrcn.jl:583) -> (grads, loss).  `grads` may be a preallocated 9-list."""
    tok = _tokens(tokens, feats.device)
    T, B = tok.shape
    if grads is None:
        grads = [jl_empty(*t.shape) for t in param]
    d, keep = _dropout(pdrop, seed, mask1, mask2)
    out = C.c_double()
    ctx._call("lrcn_loss_grad", _p9(param), _ptr(feats), C.c_void_p(tok.data_ptr()), T, B, norm_B or B,
              C.byref(d) if d else None, _p9(grads), C.byref(out) if want_loss else None)
    del keep
    return grads, (out.value if want_loss else None)


def last_loss(ctx):
    out = C.c_double()
    ctx._call("lrcn_last_loss", C.byref(out))
    return out.value


def forward_logits(ctx, param, feats, tokens):
    """Per-step logits of the loss forward pass: (T+1, B, V) numpy (parity probe)."""
    tok = _tokens(tokens, feats.device)
    T, B = tok.shape
    out = torch.empty((T + 1, ctx.V, B), device=feats.device, dtype=torch.float32)  # blocks of B x V column-major
    ctx._call("lrcn_forward_logits", _p9(param), _ptr(feats), C.c_void_p(tok.data_ptr()), T, B, C.c_void_p(out.data_ptr()))
    return out.permute(0, 2, 1).cpu().numpy()


class Adam:
    """One Knet Adam() per tensor (initparams, lrcn.jl:399-405) with Knet's defaults."""

    def __init__(self, model, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8):
        self.lr, self.beta1, self.beta2, self.eps = lr, beta1, beta2, eps
        self.t = 0
        self.m = zeros_like_model(model)
        self.v = zeros_like_model(model)


def initparams(model):
    return Adam(model)


def update(ctx, param, grads, optim):
    """update!(param, gloss, optim) (lrcn.jl:394)."""
    optim.t += 1
    ctx._call("lrcn_adam_update", _p9(param), _p9(grads), _p9(optim.m), _p9(optim.v), optim.t, optim.lr, optim.beta1,
              optim.beta2, optim.eps)


def update_group(ctx, param, grads, optim, group, stream=None):
    """update! for the tensors of one gradient group (include/lrcn.h lrcn_adam_update_group) on `stream`; the caller
    advances optim.t once per step, before the first group."""
    ctx._call("lrcn_adam_update_group", _p9(param), _p9(grads), _p9(optim.m), _p9(optim.v), int(group), optim.t, optim.lr,
              optim.beta1, optim.beta2, optim.eps, C.c_void_p(stream.cuda_stream) if stream is not None else None)


def update_flat(ctx, w, g, m, v, optim, stream=None):
    """The Adam arithmetic on one flat run of floats (lrcn_adam_update_flat): w, g, m, v are 1-D float32 tensors of equal length; the
    caller advances optim.t once per step."""
    n = w.numel()
    assert g.numel() == n and m.numel() == n and v.numel() == n
    ctx._call("lrcn_adam_update_flat", C.c_void_p(w.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(m.data_ptr()), C.c_void_p(v.data_ptr()),
              n, optim.t, optim.lr, optim.beta1, optim.beta2, optim.eps, C.c_void_p(stream.cuda_stream) if stream is not None else None)


def refresh_shadows_group(ctx, param, group, stream=None):
    """lrcn_refresh_shadows_group: the next step's shadow weights of one gradient group, on `stream`, from the parameters as they are now."""
    ctx._call("lrcn_refresh_shadows_group", _p9(param), int(group), C.c_void_p(stream.cuda_stream) if stream is not None else None)


def train_step(ctx, param, optim, grads, feats, tokens, norm_B=None, pdrop=0.4, seed=0, want_loss=False):
    """Body of train1's batch loop (lrcn.jl:369-394): lossgradient + update!, one C call."""
    tok = _tokens(tokens, feats.device)
    T, B = tok.shape
    d, keep = _dropout(pdrop, seed, None, None)
    optim.t += 1
    out = C.c_double()
    ctx._call("lrcn_train_step", _p9(param), _p9(grads), _p9(optim.m), _p9(optim.v), _ptr(feats), C.c_void_p(tok.data_ptr()),
              T, B, norm_B or B, C.byref(d) if d else None, optim.t, optim.lr, optim.beta1, optim.beta2, optim.eps,
              C.byref(out) if want_loss else None)
    del keep
    return out.value if want_loss else None


def comm_unique_id():
    """RCCL unique id (bytes) for lrcn_comm_init: rank 0 creates it, the host program distributes it."""
    buf = (C.c_char * 128)()
    _lib.check(None, _lib.lib().lrcn_comm_unique_id(buf))
    return bytes(buf)


def comm_probe(ctx):
    """LOCAL check that comm_init can be entered on this rank (librccl loadable, symbols resolved, no communicator yet): lrcn_comm_probe.
    -> (ok, message)"""
    try:
        ctx._call("lrcn_comm_probe")
        return True, ""
    except LrcnError as e:
        return False, str(e)


def comm_init(ctx, world, rank, unique_id):
    """Bind ctx to rank `rank` of a `world`-rank RCCL communicator (collective over the ranks): lrcn_comm_init."""
    buf = (C.c_char * 128).from_buffer_copy(bytes(unique_id))
    ctx._call("lrcn_comm_init", int(world), int(rank), buf)


def set_embed_rows_buffer(ctx, rows, tok):
    """Sparse exchange of the embedding gradient (lrcn_set_embed_rows_buffer): lossgradient writes its (T+1) B rows of d(x_lstm) and their token
    ids into `rows` (float32, capacity x E) / `tok` (int32, capacity) instead of the dense gradient of Wembed; (None, None) turns it off."""
    if rows is None:
        ctx._call("lrcn_set_embed_rows_buffer", None, None, 0)
    else:
        ctx._call("lrcn_set_embed_rows_buffer", C.c_void_p(rows.data_ptr()), C.c_void_p(tok.data_ptr()), int(tok.numel()))


def embed_grad_from_rows(ctx, rows, tok, n_rows, grad_wembed, stream=None):
    """The dense d Wembed (V x E column-major) from the rows of ALL ranks, summed per token in one fixed order (lrcn_embed_grad_from_rows)."""
    ctx._call("lrcn_embed_grad_from_rows", C.c_void_p(rows.data_ptr()), C.c_void_p(tok.data_ptr()), int(n_rows), _ptr(grad_wembed),
              C.c_void_p(stream.cuda_stream) if stream is not None else None)


def comm_set_stream(ctx, stream):
    """The stream for the context's collectives and per-group updates (lrcn_comm_set_stream); the caller keeps `stream` alive."""
    ctx._call("lrcn_comm_set_stream", C.c_void_p(stream.cuda_stream))


def comm_destroy(ctx):
    ctx._call("lrcn_comm_destroy")


def allreduce_grads(ctx, grads, group=-1):
    """lrcn_allreduce_grads: in-place all-reduce(SUM) of one gradient group (or all) on the context's group stream."""
    ctx._call("lrcn_allreduce_grads", _p9(grads), int(group))


def comm_join(ctx):
    ctx._call("lrcn_comm_join")


def train_step_dp(ctx, param, optim, grads, feats, tokens, norm_B=None, pdrop=0.4, seed=0, img_u8=None, mean=VGG_MEAN, normalize=False,
                  want_loss=False):
    """The data-parallel step in one C call (lrcn_train_step_dp): [VGG forward of img_u8 into feats] + lossgradient + per-group
    all-reduce over the ranks + per-group Adam.  feats: B x 4096 column-major (input, or output buffer when img_u8 is given)."""
    tok = _tokens(tokens, feats.device)
    T, B = tok.shape
    d, keep = _dropout(pdrop, seed, None, None)
    optim.t += 1
    out = C.c_double()
    m = (C.c_float * 3)(*mean) if mean is not None else None
    ctx._call("lrcn_train_step_dp", _p9(param), _p9(grads), _p9(optim.m), _p9(optim.v),
              C.c_void_p(img_u8.data_ptr()) if img_u8 is not None else None, m, int(bool(normalize)), _ptr(feats), C.c_void_p(tok.data_ptr()),
              T, B, norm_B or B, C.byref(d) if d else None, optim.t, optim.lr, optim.beta1, optim.beta2, optim.eps,
              C.byref(out) if want_loss else None)
    del keep
    return out.value if want_loss else None


def average_loss(ctx, param, batches):
    """average_loss (lrcn.jl:407-486): forward-only NLL over batches [(feats, tokens), ...], pdrop 0, captions longer
    than 28 tokens skipped (:438); returns -total/count with count = sum of B*(T+1)."""
    total, count = 0.0, 0
    for feats, tokens in batches:
        T, B = np.asarray(tokens).shape if not torch.is_tensor(tokens) else tokens.shape
        if T > 28:
            continue
        n = B * (T + 1)
        total += avg_loss_batch(ctx, param, feats, tokens) * n
        count += n
    return total / max(count, 1)


def avg_loss_batch(ctx, param, feats, tokens):
    """The body of average_loss's batch loop (lrcn.jl:452-475): pdrop 0, the loss divided by the batch's own size (lrcn_avg_loss_batch)."""
    tok = _tokens(tokens, feats.device)
    T, B = tok.shape
    out = C.c_double()
    ctx._call("lrcn_avg_loss_batch", _p9(param), _ptr(feats), C.c_void_p(tok.data_ptr()), T, B, C.byref(out))
    return out.value


def profile(ctx, level):
    """lrcn_profile: 0 off, 1 the convolution launches of every VGG forward, 2 also the HBM-bound segments (lrcn_profile_segment)."""
    ctx._call("lrcn_profile", int(level))


def profile_segments(ctx):
    """-> {segment: (milliseconds, brackets, algorithmic bytes)} accumulated since profile(ctx, 2) (synchronises the device)."""
    out = {}
    for i, name in enumerate(_lib.SEGMENTS):
        ms, n, by = C.c_double(), C.c_int64(), C.c_double()
        ctx._call("lrcn_profile_segment", i, C.byref(ms), C.byref(n), C.byref(by))
        out[name] = (ms.value, n.value, by.value)
    return out


def beam_search(ctx, param, feat, beam_width, nword):
    """generate's decode (lrcn.jl:609-633) + beam_search (:644-678) -> (token ids incl. bos, probability)."""
    out = (C.c_int32 * (nword + 3))()
    n = C.c_int()
    p = C.c_float()
    ctx._call("lrcn_beam_search", _p9(param), _ptr(feat), beam_width, nword, out, C.byref(n), C.byref(p))
    return list(out[:n.value]), p.value


def beam_search_batch(ctx, param, feats, beam_width, nword):
    """beam_search for N images in one device-resident decode (lrcn_beam_search_batch): feats N x 4096 ->
    [(token ids incl. bos, probability)] per image; N * beam_width <= max_B."""
    N = feats.shape[0]
    L = nword + 2
    out = (C.c_int32 * (N * L))()
    n = (C.c_int * N)()
    p = (C.c_float * N)()
    ctx._call("lrcn_beam_search_batch", _p9(param), _ptr(feats), N, beam_width, nword, out, n, p)
    return [(list(out[i * L:i * L + n[i]]), p[i]) for i in range(N)]


def generate(ctx, param, feat, index_to_word, nword, beam_width, normalize=False):
    """generate (lrcn.jl:585-642): caption text "w1 w2 ... ." -- words after bos up to the first eos (:634-640)."""
    if normalize:
        feat = to_jl(from_jl(feat) / from_jl(feat).sum())  # input/sum(input) (lrcn.jl:597)
    seq, _ = beam_search(ctx, param, feat, beam_width, nword)
    words = []
    for t in seq[1:]:
        if t == EOS:
            break
        words.append(index_to_word[t])
    return " ".join(words + ["."])


# ------------------------------------------------------------------------------------------------ VGG
def vgg_load(ctx, conv_w, conv_b, fc6, fc7):
    """get_params_cnn's output (lrcn.jl:697-721) -> the context. conv_w[l]: (3,3,Cin,Cout) column-major tensors."""
    cw = _lib.P13(*[_ptr(t).value for t in conv_w])
    cb = _lib.P13(*[_ptr(t).value for t in conv_b])
    ctx._call("lrcn_vgg_load", cw, cb, _ptr(fc6[0]), _ptr(fc6[1]), _ptr(fc7[0]), _ptr(fc7[1]))


def convnet(ctx, x):
    """convnet(xs) (lrcn.jl:734-747): x (224,224,3,N) column-major -> feats N x 4096."""
    N = x.shape[3]
    feats = jl_empty(N, CNNOUT)
    ctx._call("lrcn_vgg_forward", _ptr(x), N, _ptr(feats))
    return feats


def convnet_u8(ctx, img_u8, mean=VGG_MEAN, feats=None, normalize=False):
    """Fused read_image_data arithmetic (lrcn.jl:766-772) + convnet on uint8 crops img[n][row][col][c].  mean=None: the
    averageImage registered with set_average_image.  normalize: divide each feature row by its sum (lrcn.jl:595-597)."""
    if torch.is_tensor(img_u8) and not img_u8.is_cuda:  # host crops: through the staging buffers (asynchronous when pinned)
        img_u8 = upload_crops(ctx, img_u8)
    N = img_u8.shape[0]
    if feats is None:
        feats = jl_empty(N, CNNOUT)
    m = (C.c_float * 3)(*mean) if mean is not None else None
    ctx._call("lrcn_vgg_forward_u8", C.c_void_p(img_u8.data_ptr()), N, m, _ptr(feats))
    if normalize:
        normalize_features(ctx, feats)
    return feats


def convnet_u8_blocks(ctx, img_u8, block_rows, mean=VGG_MEAN, feats=None, normalize=False):
    """The VGG forward for the crops of m = N / block_rows training batches at once (lrcn_vgg_forward_u8_blocks) -> list of m feature
    tensors block_rows x 4096 (column-major views into one buffer `feats` of m * block_rows * 4096 floats)."""
    if torch.is_tensor(img_u8) and not img_u8.is_cuda:
        img_u8 = upload_crops(ctx, img_u8)
    N = img_u8.shape[0]
    m = N // block_rows
    if feats is None:
        feats = torch.empty(m * block_rows * CNNOUT, device="cuda", dtype=torch.float32)
    mm = (C.c_float * 3)(*mean) if mean is not None else None
    ctx._call("lrcn_vgg_forward_u8_blocks", C.c_void_p(img_u8.data_ptr()), N, mm, int(block_rows), int(bool(normalize)), C.c_void_p(feats.data_ptr()))
    n = block_rows * CNNOUT
    return [feats[b * n:(b + 1) * n].view(CNNOUT, block_rows).permute(1, 0) for b in range(m)]


class StagedCrops:
    """uint8 crops [N][224][224][3] in one of the context's two device staging buffers (lrcn_upload_crops): stands wherever a device crop
    tensor does (convnet_u8, train_step_dp).  `host` keeps the source buffer alive until the asynchronous copy has certainly been issued
    AND consumed (the trainer drops it after the forward that reads the staging buffer has been queued)."""

    def __init__(self, ptr, N, host):
        self.ptr, self.shape, self.host = int(ptr), (int(N), 224, 224, 3), host
        self.is_cuda = True

    def data_ptr(self):
        return self.ptr


def upload_crops(ctx, host_u8):
    """The per-batch host -> device copy of the training loop (lrcn.jl:369-376) off the critical path: `host_u8` (CPU uint8 tensor
    [N][224][224][3]; pinned = a true asynchronous DMA) is copied on the context's copy stream into a device staging buffer.  Returns at
    once; the forward that is handed the result waits on the device (include/lrcn.h "input feed")."""
    if host_u8.is_cuda or host_u8.dtype != torch.uint8 or not host_u8.is_contiguous() or tuple(host_u8.shape[1:]) != (224, 224, 3):
        raise LrcnError("upload_crops takes a contiguous CPU uint8 tensor [N][224][224][3]")
    out = C.c_void_p()
    ctx._call("lrcn_upload_crops", C.c_void_p(host_u8.data_ptr()), int(host_u8.shape[0]), C.byref(out))
    return StagedCrops(out.value, host_u8.shape[0], host_u8)


def upload_wait(ctx):
    """Block until every upload issued so far has run (a loader calls it before refilling a pinned buffer)."""
    ctx._call("lrcn_upload_wait")


def read_image_data_u8(ctx, img_u8, mean=VGG_MEAN):
    """read_image_data's arithmetic tail (lrcn.jl:766-772) -> (224,224,3,N) column-major float tensor."""
    N = img_u8.shape[0]
    out = jl_empty(224, 224, 3, N)
    m = (C.c_float * 3)(*mean) if mean is not None else None
    ctx._call("lrcn_preprocess_u8", C.c_void_p(img_u8.data_ptr()), N, m, _ptr(out))
    return out


def set_average_image(ctx, average_image):
    """Register the VGG averageImage (lrcn.jl:113), a (224,224,3) array, or None to go back to per-channel means."""
    if average_image is None:
        ctx._call("lrcn_set_average_image", None)
        return
    a = to_jl(np.asarray(average_image, np.float32).reshape(224, 224, 3))
    ctx._call("lrcn_set_average_image", _ptr(a))
    ctx.sync()  # `a` may be freed on return


def resize_crop_u8(ctx, images):
    """read_image_data's geometry (lrcn.jl:755-765) on the GPU for a list of decoded images (uint8 arrays [h][w], [h][w][1],
    [h][w][3] or [h][w][4]) of any sizes -> uint8 crops tensor [N][224][224][3] on the device (lrcn_resize_crop_u8)."""
    arrs = [np.ascontiguousarray(np.asarray(im, dtype=np.uint8)) for im in images]
    arrs = [a[:, :, None] if a.ndim == 2 else a for a in arrs]
    N = len(arrs)
    offs = np.zeros(N, np.int64)
    pos = 0
    for i, a in enumerate(arrs):
        offs[i] = pos
        pos += a.size
    flat = torch.as_tensor(np.concatenate([a.reshape(-1) for a in arrs])).cuda()
    hs = (C.c_int * N)(*[a.shape[0] for a in arrs])
    ws = (C.c_int * N)(*[a.shape[1] for a in arrs])
    cs = (C.c_int * N)(*[a.shape[2] for a in arrs])
    out = torch.empty((N, 224, 224, 3), dtype=torch.uint8, device=flat.device)
    ctx._call("lrcn_resize_crop_u8", C.c_void_p(flat.data_ptr()), offs.ctypes.data_as(C.POINTER(C.c_int64)), hs, ws, cs, N,
              C.c_void_p(out.data_ptr()))
    ctx.sync()  # `flat` may be freed on return
    return out


def normalize_features(ctx, feats):
    """feats (N x 4096, column-major) <- rows divided by their sums, in place: generate's input/sum(input) (lrcn.jl:595-597)."""
    ctx._call("lrcn_normalize_features", _ptr(feats), feats.shape[0])
    return feats


def conv3x3(ctx, x, w, b, relu=True, pool=False):
    """convx/relux/poolx probe (lrcn.jl:724-726) in reference layouts."""
    W, H, Cin, N = x.shape
    Cout = w.shape[3]
    y = jl_empty(W // 2 if pool else W, H // 2 if pool else H, Cout, N)
    ctx._call("lrcn_conv3x3", _ptr(x), W, H, Cin, N, _ptr(w), _ptr(b), Cout, int(relu), int(pool), _ptr(y))
    return y


def conv1_fused(ctx, img_u8, mean, w11, b11, w12, b12):
    """The bf16 stack's first launch as a probe (conv64f.hip): crops [N][S][S][3] uint8 (device) -> pool1 (S/2, S/2, 64, N); w11 (3,3,3,64),
    w12 (3,3,64,64) in the reference layouts (lrcn.jl:770 + 724-726 twice)."""
    N, S = img_u8.shape[0], img_u8.shape[1]
    y = jl_empty(S // 2, S // 2, 64, N)
    m = (C.c_float * 3)(*[float(v) for v in mean])
    ctx._call("lrcn_conv1_fused", C.c_void_p(img_u8.data_ptr()), N, S, m, _ptr(w11), _ptr(b11), _ptr(w12), _ptr(b12), _ptr(y))
    return y


def vgg_set_wg_cap(ctx, cap):
    """Cap the VGG convolution grids at `cap` workgroups (0 = off): include/lrcn.h lrcn_vgg_set_wg_cap."""
    ctx._call("lrcn_vgg_set_wg_cap", int(cap))


def vgg_calibrate(ctx, img_u8, mean=VGG_MEAN, margin=1.25):
    """vgg_dtype = LRCN_FP8 only: one bf16 pass over `img_u8` (uint8 crops [n][row][col][c]) that fixes the per-layer
    activation scales of the e4m3 layers (include/lrcn.h lrcn_vgg_calibrate).  No counterpart in the reference."""
    m = (C.c_float * 3)(*mean) if mean is not None else None
    ctx._call("lrcn_vgg_calibrate", C.c_void_p(img_u8.data_ptr()), img_u8.shape[0], m, float(margin))


def conv3x3_fp8(ctx, x, w, b, sa_in, sa_out, relu=True, pool=False):
    """e4m3 probe of one layer (include/lrcn.h lrcn_conv3x3_fp8) -> (y dequantised, per-channel weight scales)."""
    W, H, Cin, N = x.shape
    Cout = w.shape[3]
    y = jl_empty(W // 2 if pool else W, H // 2 if pool else H, Cout, N)
    sw = torch.empty(Cout, dtype=torch.float32, device=x.device)
    ctx._call("lrcn_conv3x3_fp8", _ptr(x), W, H, Cin, N, _ptr(w), _ptr(b), Cout, int(relu), int(pool), float(sa_in), float(sa_out),
              _ptr(y), C.c_void_p(sw.data_ptr()))
    return y, sw


def debug_route(ctx, which=0):
    """Kernel family of the last contraction (which=0) / per layer of the last VGG forward (which=1): lrcn_debug_route."""
    r = _lib.lib().lrcn_debug_route(ctx._h, int(which))
    return r.decode() if r else ""


def synthetic_vgg_weights(seed=1, device="cuda", bias_std=0.0):
    """He-normal(fan_in) VGG-16 weights (no pretrained file offline); biases zero, or N(0, bias_std)
    so that every bias path of the kernels sees non-zero values (parity tests)."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    conv_w, conv_b = [], []
    cin = 3
    for cout in VGG_COUT:
        w = jl_empty(3, 3, cin, cout, device=device)
        w.copy_(torch.randn((3, 3, cin, cout), generator=g, device=device) * float(np.sqrt(2.0 / (9 * cin))))
        conv_w.append(w)
        conv_b.append(torch.randn(cout, generator=g, device=device) * bias_std if bias_std else torch.zeros(cout, device=device))
        cin = cout
    fc6 = jl_empty(4096, 25088, device=device)
    fc6.copy_(torch.randn((4096, 25088), generator=g, device=device) * float(np.sqrt(2.0 / 25088)))
    fc7 = jl_empty(4096, 4096, device=device)
    fc7.copy_(torch.randn((4096, 4096), generator=g, device=device) * float(np.sqrt(2.0 / 4096)))
    b6 = torch.randn(4096, generator=g, device=device) * bias_std if bias_std else torch.zeros(4096, device=device)
    b7 = torch.randn(4096, generator=g, device=device) * bias_std if bias_std else torch.zeros(4096, device=device)
    return conv_w, conv_b, (fc6, b6), (fc7, b7)
