"""ctypes binding of the CPU oracle (oracle/lrcn_oracle.c).

TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product package never does.  Arrays follow the reference's conventions: numpy arrays in
Fortran (column-major) order with exactly the shapes lrcn.jl uses; token ids are 0-based (eos=0,bos=1,unk=2).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
EOS, BOS, UNK = 0, 1, 2
CNNOUT = 4096
PARAM_NAMES = ("W1", "b1", "W2", "b2", "Wproj", "Wcnn", "Wembed", "Wout", "bout")
VGG_COUT = (64, 64, 128, 128, 256, 256, 256, 512, 512, 512, 512, 512, 512)
VGG_POOL_AFTER = (0, 1, 0, 1, 0, 0, 1, 0, 0, 1, 0, 0, 1)

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)


class _Model(C.Structure):
    _fields_ = [("E", C.c_int), ("H1", C.c_int), ("H2", C.c_int), ("V", C.c_int)] + [
        (n, _fp) for n in PARAM_NAMES
    ]


class _Vgg(C.Structure):
    _fields_ = [("conv_w", _fp * 13), ("conv_b", _fp * 13), ("fc6_w", _fp), ("fc6_b", _fp), ("fc7_w", _fp),
                ("fc7_b", _fp)]


def build(force=False):
    """Compile the oracle's shared libraries (gcc). Building the checker is not using it."""
    libs = [os.path.join(HERE, n) for n in ("liblrcn_oracle.so", "liblrcn_oracle_f32.so", "liblrcn_cpu.so", "liblrcn_cpu_f32.so")]
    src = [os.path.join(HERE, n) for n in ("lrcn_oracle.c", "lrcn_oracle.h", "lrcn_cpu_abi.c", "Makefile")] + [
        os.path.join(HERE, "..", "include", "lrcn.h")]
    stale = force or any(
        not os.path.exists(l) or os.path.getmtime(l) < max(os.path.getmtime(s) for s in src) for l in libs)
    if stale:
        subprocess.check_call(["make", "-s", "-C", HERE, "-f", os.path.join(HERE, "Makefile")])
    return libs


def _has_avx2():
    try:
        with open("/proc/cpuinfo") as f:
            txt = f.read()
        return " avx2" in txt and " fma" in txt
    except OSError:
        return False


_LIBS = {}


def lib(fast=False):
    """fast=False: the checker (double accumulation). fast=True: float-accumulation build for CPU timing."""
    key = bool(fast) and _has_avx2()
    if key not in _LIBS:
        path = os.path.join(HERE, "liblrcn_oracle_f32.so" if key else "liblrcn_oracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        L.orc_loss.restype = C.c_double
        L.orc_loss.argtypes = [C.POINTER(_Model), _fp, _ip, C.c_int, C.c_int, C.c_int, _fp, _fp, C.POINTER(_Model)]
        L.orc_forward_logits.restype = None
        L.orc_forward_logits.argtypes = [C.POINTER(_Model), _fp, _ip, C.c_int, C.c_int, _fp]
        L.orc_param_count.restype = C.c_int64
        L.orc_param_count.argtypes = [C.c_int] * 4
        L.orc_init_weights.restype = None
        L.orc_init_weights.argtypes = [C.POINTER(_Model), C.c_uint64]
        L.orc_lstm.restype = None
        L.orc_lstm.argtypes = [_fp, _fp, C.c_int, C.c_int, C.c_int, _fp, _fp, _fp, _fp, _fp, _fp]
        L.orc_lrcn_step.restype = None
        L.orc_lrcn_step.argtypes = [C.POINTER(_Model), C.c_int] + [_fp] * 9
        L.orc_adam.restype = None
        L.orc_adam.argtypes = [_fp, _fp, _fp, _fp, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]
        L.orc_beam_search.restype = C.c_int
        L.orc_beam_search.argtypes = [C.POINTER(_Model), _fp, C.c_int, C.c_int, _ip, _fp]
        L.orc1_init_weights.restype = None
        L.orc1_init_weights.argtypes = [C.POINTER(_Model), C.c_uint64]
        L.orc1_step.restype = None
        L.orc1_step.argtypes = [C.POINTER(_Model), C.c_int] + [_fp] * 6
        L.orc1_loss.restype = C.c_double
        L.orc1_loss.argtypes = [C.POINTER(_Model), _fp, _ip, C.c_int, C.c_int, C.c_int, _fp, C.POINTER(_Model)]
        L.orc1_forward_logits.restype = None
        L.orc1_forward_logits.argtypes = [C.POINTER(_Model), _fp, _ip, C.c_int, C.c_int, _fp]
        L.orc1_beam_search.restype = C.c_int
        L.orc1_beam_search.argtypes = [C.POINTER(_Model), _fp, C.c_int, C.c_int, _ip, _fp]
        L.orc_conv3x3.restype = None
        L.orc_conv3x3.argtypes = [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp, _fp, C.c_int, C.c_int, _fp]
        L.orc_pool2.restype = None
        L.orc_pool2.argtypes = [_fp, C.c_int, C.c_int, C.c_int, C.c_int, _fp]
        L.orc_fc.restype = None
        L.orc_fc.argtypes = [_fp, _fp, C.c_int, C.c_int, C.c_int, _fp, C.c_int, _fp]
        L.orc_vgg_forward.restype = None
        L.orc_vgg_forward.argtypes = [C.POINTER(_Vgg), _fp, C.c_int, C.c_int, _fp]
        L.orc_preprocess_u8.restype = None
        L.orc_preprocess_u8.argtypes = [C.POINTER(C.c_uint8), C.c_int, C.c_int, _fp, _fp]
        L.orc_set_emulate_bf16.restype = None
        L.orc_set_emulate_bf16.argtypes = [C.c_int]
        L.orc_get_emulate_bf16.restype = C.c_int
        L.orc_bf16_round.restype = C.c_float
        L.orc_bf16_round.argtypes = [C.c_float]
        L.orc_num_threads.restype = C.c_int
        L.orc_set_num_threads.restype = None
        L.orc_set_num_threads.argtypes = [C.c_int]
        _LIBS[key] = L
    return _LIBS[key]


class emulate_bf16:
    """`with orc.emulate_bf16():` -- the checker rounds to bfloat16 wherever the HIP library's bf16 arithmetic does
    (ORC_EMULATE_BF16, lrcn_oracle.h).  Outside the block the oracle is the plain fp32-storage restatement again."""

    def __init__(self, on=True):
        self.on = bool(on)

    def __enter__(self):
        self.prev = lib().orc_get_emulate_bf16()
        lib().orc_set_emulate_bf16(int(self.on))
        return self

    def __exit__(self, *exc):
        lib().orc_set_emulate_bf16(self.prev)
        return False


def bf16_round(a):
    """Round a float32 array to bfloat16 (nearest even), returned as float32 -- numpy twin of orc_bf16_round."""
    a = np.ascontiguousarray(np.asarray(a, np.float32))
    u = a.view(np.uint32).astype(np.uint64)
    nan = (u & 0x7FFFFFFF) > 0x7F800000
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    out = r.view(np.float32).reshape(a.shape).copy()
    out[nan.reshape(a.shape)] = a[nan.reshape(a.shape)]
    return out


def _f(a):
    return a.ctypes.data_as(_fp) if a is not None else None


def fa(x, shape=None):
    """float32 Fortran-order array (a copy unless already so)."""
    a = np.asfortranarray(np.asarray(x, dtype=np.float32))
    if shape is not None:
        a = np.asfortranarray(a.reshape(shape, order="F"))
    return a


def param_shapes(E, H1, H2, V, n_layers=2):
    """Reference shapes of the 9 tensors (lrcn.jl:489-510); n_layers=1: LRCN-1f (lrcn_oracle.h), absent tensors (0, 0)."""
    h = (H2 + 1) // 2
    if n_layers == 1:
        assert H1 == H2
        return {"W1": (E + h + H1, 4 * H1), "b1": (1, 4 * H1), "W2": (0, 0), "b2": (0, 0), "Wproj": (0, 0),
                "Wcnn": (CNNOUT, h), "Wembed": (V, E), "Wout": (H2, V), "bout": (1, V)}
    return {"W1": (E + H1, 4 * H1), "b1": (1, 4 * H1), "W2": (H2 + H2, 4 * H2), "b2": (1, 4 * H2),
            "Wproj": (H1, h), "Wcnn": (CNNOUT, h), "Wembed": (V, E), "Wout": (H2, V), "bout": (1, V)}


class Model:
    """The reference's `model` vector: 9 Fortran-order float32 arrays with lrcn.jl's shapes (n_layers=1: LRCN-1f)."""

    def __init__(self, E, H1, H2, V, arrays=None, n_layers=2):
        self.E, self.H1, self.H2, self.V, self.n_layers = E, H1, H2, V, n_layers
        shp = param_shapes(E, H1, H2, V, n_layers)
        self.p = {}
        for n in PARAM_NAMES:
            if arrays is not None:
                a = fa(arrays[n])
                assert a.shape == shp[n], (n, a.shape, shp[n])
            else:
                a = np.zeros(shp[n], dtype=np.float32, order="F")
            self.p[n] = a

    def cstruct(self):
        m = _Model(self.E, self.H1, self.H2, self.V)
        for n in PARAM_NAMES:
            setattr(m, n, _f(self.p[n]))
        return m

    def zeros_like(self):
        return Model(self.E, self.H1, self.H2, self.V, n_layers=self.n_layers)

    def arrays(self):
        return [self.p[n] for n in PARAM_NAMES]


def init_weights(E, H1, H2, V, seed=42, n_layers=2):
    m = Model(E, H1, H2, V, n_layers=n_layers)
    cs = m.cstruct()
    (lib().orc1_init_weights if n_layers == 1 else lib().orc_init_weights)(C.byref(cs), seed)
    return m


def _tok(tokens):
    t = np.ascontiguousarray(np.asarray(tokens, dtype=np.int32))  # [T][B]
    assert t.ndim == 2
    return t


def loss(model, feats, tokens, norm_B=None, mask1=None, mask2=None, want_grad=False, fast=False):
    """loss / lossgradient (lrcn.jl:553-583). feats: B x 4096; tokens: [T][B] int (0-based).
    mask1: [(T+1)] x (B x E), mask2: [(T+1)] x (B x H2) dropout multipliers, as arrays of shape (T+1, B, E) /
    (T+1, B, H2) (each block is stored column-major)."""
    tokens = _tok(tokens)
    T, B = tokens.shape
    feats = fa(feats)
    assert feats.shape == (B, CNNOUT)
    m1 = m2 = None
    if mask1 is not None:
        m1 = np.ascontiguousarray(np.stack([np.asfortranarray(b).ravel(order="F") for b in mask1]), dtype=np.float32)
    if mask2 is not None:
        m2 = np.ascontiguousarray(np.stack([np.asfortranarray(b).ravel(order="F") for b in mask2]), dtype=np.float32)
    cs = model.cstruct()
    g = model.zeros_like() if want_grad else None
    gs = g.cstruct() if want_grad else None
    if model.n_layers == 1:  # LRCN-1f: one mask over the (E + h) columns of hcat(x_lstm, x_cnn)
        val = lib(fast).orc1_loss(C.byref(cs), _f(feats), tokens.ctypes.data_as(_ip), T, B, norm_B or B, _f(m1),
                                  C.byref(gs) if want_grad else None)
        return (val, g) if want_grad else val
    val = lib(fast).orc_loss(C.byref(cs), _f(feats), tokens.ctypes.data_as(_ip), T, B, norm_B or B, _f(m1), _f(m2),
                             C.byref(gs) if want_grad else None)
    return (val, g) if want_grad else val


def forward_logits(model, feats, tokens):
    tokens = _tok(tokens)
    T, B = tokens.shape
    feats = fa(feats)
    out = np.zeros((T + 1, model.V, B), dtype=np.float32)  # each block B x V column-major == [V][B] C-order
    cs = model.cstruct()
    (lib().orc1_forward_logits if model.n_layers == 1 else lib().orc_forward_logits)(C.byref(cs), _f(feats), tokens.ctypes.data_as(_ip),
                                                                                      T, B, _f(out))
    return np.transpose(out, (0, 2, 1))  # -> (T+1, B, V)


def lstm(W, b, x, h, c):
    W, b, x, h, c = fa(W), fa(b), fa(x), fa(h), fa(c)
    B, X = x.shape
    H = h.shape[1]
    ho = np.zeros((B, H), np.float32, order="F")
    co = np.zeros((B, H), np.float32, order="F")
    lib().orc_lstm(_f(W), _f(b), X, H, B, _f(x), _f(h), _f(c), _f(ho), _f(co), None)
    return ho, co


def lrcn_step(model, state, x_cnn, x_lstm, mask1=None, mask2=None):
    """lrcn (lrcn.jl:540-551). state = [h1,c1,h2,c2] (updated in place, Fortran arrays). Returns logits B x V."""
    B = x_lstm.shape[0]
    for i in range(len(state)):
        state[i] = fa(state[i])
    logits = np.zeros((B, model.V), np.float32, order="F")
    cs = model.cstruct()
    if model.n_layers == 1:  # state = [h, c]; mask1 = B x (E + h)
        lib().orc1_step(C.byref(cs), B, _f(state[0]), _f(state[1]), _f(fa(x_cnn)), _f(fa(x_lstm)),
                        _f(fa(mask1)) if mask1 is not None else None, _f(logits))
        return logits
    lib().orc_lrcn_step(C.byref(cs), B, _f(state[0]), _f(state[1]), _f(state[2]), _f(state[3]), _f(fa(x_cnn)),
                        _f(fa(x_lstm)), _f(fa(mask1)) if mask1 is not None else None,
                        _f(fa(mask2)) if mask2 is not None else None, _f(logits))
    return logits


def adam(w, g, m, v, t, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8):
    """In-place Adam on flat views of Fortran arrays (update!, lrcn.jl:394)."""
    for a in (w, m, v):
        assert a.flags.f_contiguous or a.flags.c_contiguous
    lib().orc_adam(_f(w), _f(np.asfortranarray(g)), _f(m), _f(v), w.size, t, lr, beta1, beta2, eps)


def beam_search(model, feat, K, nword):
    feat = fa(np.asarray(feat).reshape(1, CNNOUT))
    out = np.zeros(nword + 3, np.int32)
    prob = C.c_float(0)
    cs = model.cstruct()
    fn = lib().orc1_beam_search if model.n_layers == 1 else lib().orc_beam_search
    n = fn(C.byref(cs), _f(feat), K, nword, out.ctypes.data_as(_ip), C.byref(prob))
    return out[:n].copy(), prob.value


def conv3x3(x, w, b, relu=True):
    """x: (W,H,Cin,N) F-order; w: (3,3,Cin,Cout) F-order; b: (Cout,)"""
    x, w, b = fa(x), fa(w), fa(b)
    W, H, Cin, N = x.shape
    Cout = w.shape[3]
    y = np.zeros((W, H, Cout, N), np.float32, order="F")
    lib().orc_conv3x3(_f(x), W, H, Cin, N, _f(w), _f(b), Cout, int(relu), _f(y))
    return y


def pool2(x):
    x = fa(x)
    W, H, Cc, N = x.shape
    y = np.zeros((W // 2, H // 2, Cc, N), np.float32, order="F")
    lib().orc_pool2(_f(x), W, H, Cc, N, _f(y))
    return y


def fc(w, b, x, relu=False):
    w, b, x = fa(w), fa(b), fa(x)
    O, K = w.shape
    N = x.shape[1]
    y = np.zeros((O, N), np.float32, order="F")
    lib().orc_fc(_f(w), _f(b), O, K, N, _f(x), int(relu), _f(y))
    return y


def vgg_forward(conv_w, conv_b, fc6, fc7, x, fast=False):
    """convnet (lrcn.jl:733-748). conv_w[l]: (3,3,Cin,Cout) F; fc6=(w 4096x25088 F, b); x: (224,224,3,N) F.
    Returns feats N x 4096 (F-order)."""
    x = fa(x)
    S, _, _, N = x.shape
    keep = [fa(a) for a in conv_w] + [fa(a) for a in conv_b] + [fa(fc6[0]), fa(fc6[1]), fa(fc7[0]), fa(fc7[1])]
    v = _Vgg()
    for l in range(13):
        v.conv_w[l] = _f(keep[l])
        v.conv_b[l] = _f(keep[13 + l])
    v.fc6_w, v.fc6_b, v.fc7_w, v.fc7_b = (_f(a) for a in keep[26:])
    feats = np.zeros((N, CNNOUT), np.float32, order="F")
    lib(fast).orc_vgg_forward(C.byref(v), _f(x), S, N, _f(feats))
    return feats


def preprocess_u8(img, mean):
    """img: uint8 [N][S][S][3] (decoder order). Returns (S,S,3,N) F-order float32 per lrcn.jl:766-772."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    N, S = img.shape[0], img.shape[1]
    out = np.zeros((S, S, 3, N), np.float32, order="F")
    mean = np.ascontiguousarray(mean, dtype=np.float32)
    lib().orc_preprocess_u8(img.ctypes.data_as(C.POINTER(C.c_uint8)), S, N, _f(mean), _f(out))
    return out


def preprocess_u8_avg(img, average_image):
    """read_image_data's arithmetic with the FULL averageImage (lrcn.jl:766-772), restated with numpy in the reference's own
    order of operations: c1(col,row,ch) [:762-766] -> 255 e1 - averageImage [:770] -> permutedims [2,1,3,4] [:771].
    img: uint8 [N][S][S][3]; average_image: (S,S,3).  Returns (S,S,3,N) F-order float32."""
    img = np.asarray(img, np.uint8)
    avg = np.asarray(average_image, np.float32)
    outs = []
    for n in range(img.shape[0]):
        c1 = np.transpose(img[n].astype(np.float32), (1, 0, 2))   # channelview (3,rows,cols) permuted (3,2,1) -> (cols, rows, 3)
        f1 = c1 - avg                                             # 255 * e1 .- averageImage  (e1 holds x/255)
        outs.append(np.transpose(f1, (1, 0, 2)))                  # g1 = permutedims(f1, [2,1,3,4])
    return np.asfortranarray(np.stack(outs, axis=3))


def resize_crop_u8(images, S=224):
    """NumPy restatement of read_image_data's geometry (lrcn.jl:755-765) with the resampling rule the HIP kernel states
    (include/lrcn.h lrcn_resize_crop_u8): bilinear between pixel centres, exact integer arithmetic, round half up.
    images: list of uint8 arrays [h][w] / [h][w][c], c in {1,3,4}.  Returns uint8 [N][S][S][3]."""
    out = np.zeros((len(images), S, S, 3), np.uint8)
    for n, im in enumerate(images):
        im = np.asarray(im, np.uint8)
        if im.ndim == 2:
            im = im[:, :, None]
        h, w, ch = im.shape
        sm = min(h, w)
        nh, nw = (h * S) // sm, (w * S) // sm                      # :756
        R = np.arange(S, dtype=np.int64) + (nh - S) // 2           # :758-760
        Q = np.arange(S, dtype=np.int64) + (nw - S) // 2
        ny = np.maximum((2 * R + 1) * h - nh, 0)
        nx = np.maximum((2 * Q + 1) * w - nw, 0)
        y0, fy = ny // (2 * nh), ny % (2 * nh)
        x0, fx = nx // (2 * nw), nx % (2 * nw)
        y1, x1 = np.minimum(y0 + 1, h - 1), np.minimum(x0 + 1, w - 1)
        src = im[:, :, :3].astype(np.int64) if ch >= 3 else np.repeat(im[:, :, :1].astype(np.int64), 3, axis=2)  # :762-764
        p00, p01 = src[y0][:, x0], src[y0][:, x1]
        p10, p11 = src[y1][:, x0], src[y1][:, x1]
        FX, FY = fx[None, :, None], fy[:, None, None]
        top = (2 * nw - FX) * p00 + FX * p01
        bot = (2 * nw - FX) * p10 + FX * p11
        out[n] = (((2 * nh - FY) * top + FY * bot + 2 * nh * nw) // (4 * nh * nw)).astype(np.uint8)
    return out


def num_threads():
    return lib().orc_num_threads()


def cpu_abi(signatures, fast=False):
    """liblrcn_cpu.so (fast=True: liblrcn_cpu_f32.so, the baseline build) bound with the product binding's own signature table
    (lrcn_amd._lib.SIGNATURES): the C ABI of include/lrcn.h on the host."""
    name = "liblrcn_cpu_f32.so" if (fast and _has_avx2()) else "liblrcn_cpu.so"
    path = os.path.join(HERE, name)
    if not os.path.exists(path):
        build()
    L = C.CDLL(path)
    for n, (res, args) in signatures.items():
        fn = getattr(L, n)
        fn.restype, fn.argtypes = res, args
    L.orc_set_num_threads.restype = None
    L.orc_set_num_threads.argtypes = [C.c_int]
    return L


def effective_cpus():
    """CPUs this process may actually use: the affinity mask capped by the cgroup CPU quota (a GPU box hands a container a share
    of the host's cores; OpenMP's default thread count is the host's)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def set_num_threads(n, fast=None):
    for f in ((False, True) if fast is None else (fast,)):
        lib(f).orc_set_num_threads(int(n))
