"""ctypes binding of liblrcn_hip.so (the C ABI declared in include/lrcn.h).

There is NO fallback: if the shared library is missing or a symbol is absent this module raises, and every
compute entry point needs a visible MI355X.  Nothing here imports or calls the CPU oracle.
"""
import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.environ.get("LRCN_HIP_LIB") or os.path.join(CSRC, "liblrcn_hip.so")  # override: A/B builds of the kernels in one session
HEADER = os.path.normpath(os.path.join(HERE, "..", "include", "lrcn.h"))

LRCN_F32, LRCN_BF16, LRCN_FP8 = 0, 1, 2
LRCN_ABI_VERSION = 5   # include/lrcn.h: the revision this binding's struct layouts and signatures were written against
LRCN_OPT_FUSED_UPDATE, LRCN_OPT_DETERMINISTIC, LRCN_OPT_CONV_CHUNK_BYTES = 1, 2, 3
SEGMENTS = ("update", "rec_fwd", "rec_bwd", "embed_gather", "embed_grad", "preprocess", "upload")   # LRCN_SEG_* of include/lrcn.h, in order
EOS, BOS, UNK = 0, 1, 2
CNNOUT = 4096
MAX_T = 28


class LrcnError(RuntimeError):
    pass


class Config(C.Structure):
    _fields_ = [("device", C.c_int), ("E", C.c_int), ("H1", C.c_int), ("H2", C.c_int), ("V", C.c_int),
                ("max_B", C.c_int), ("max_T", C.c_int), ("lstm_dtype", C.c_int), ("vgg_dtype", C.c_int),
                ("max_images", C.c_int), ("n_layers", C.c_int)]


class Dropout(C.Structure):
    _fields_ = [("pdrop", C.c_float), ("seed", C.c_uint64), ("mask1", C.c_void_p), ("mask2", C.c_void_p)]


P9 = C.c_void_p * 9
P4 = C.c_void_p * 4
P13 = C.c_void_p * 13

# name -> (restype, argtypes); exactly the symbols include/lrcn.h declares
SIGNATURES = {
    "lrcn_create": (C.c_int, [C.POINTER(Config), C.POINTER(C.c_void_p)]),
    "lrcn_destroy": (None, [C.c_void_p]),
    "lrcn_last_error": (C.c_char_p, [C.c_void_p]),
    "lrcn_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lrcn_set_wg_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lrcn_sync": (C.c_int, [C.c_void_p]),
    "lrcn_malloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "lrcn_free": (C.c_int, [C.c_void_p]),
    "lrcn_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "lrcn_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "lrcn_version": (C.c_char_p, []),
    "lrcn_abi_version": (C.c_int, []),
    "lrcn_set_option": (C.c_int, [C.c_void_p, C.c_int, C.c_int64]),
    "lrcn_params_touched": (C.c_int, [C.c_void_p]),
    "lrcn_comm_probe": (C.c_int, [C.c_void_p]),
    "lrcn_param_sizes": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "lrcn_param_sizes_n": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int64)]),
    "lrcn_init_weights": (C.c_int, [C.c_void_p, P9, C.c_uint64]),
    "lrcn_lstm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                            C.c_void_p, C.c_void_p, C.c_void_p]),
    "lrcn_step": (C.c_int, [C.c_void_p, P9, P4, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lrcn_loss": (C.c_int, [C.c_void_p, P9, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(Dropout),
                            C.POINTER(C.c_double)]),
    "lrcn_loss_grad": (C.c_int, [C.c_void_p, P9, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(Dropout),
                                 P9, C.POINTER(C.c_double)]),
    "lrcn_refresh_shadows_group": (C.c_int, [C.c_void_p, P9, C.c_int, C.c_void_p]),
    "lrcn_avg_loss_batch": (C.c_int, [C.c_void_p, P9, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "lrcn_grad_group_wait": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p]),
    "lrcn_last_loss": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "lrcn_forward_logits": (C.c_int, [C.c_void_p, P9, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "lrcn_adam_update": (C.c_int, [C.c_void_p, P9, P9, P9, P9, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float]),
    "lrcn_adam_update_group": (C.c_int, [C.c_void_p, P9, P9, P9, P9, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                         C.c_void_p]),
    "lrcn_adam_update_flat": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_float, C.c_float, C.c_float,
                                        C.c_float, C.c_void_p]),
    "lrcn_train_step": (C.c_int, [C.c_void_p, P9, P9, P9, P9, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                  C.POINTER(Dropout), C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                  C.POINTER(C.c_double)]),
    "lrcn_comm_unique_id": (C.c_int, [C.c_void_p]),
    "lrcn_comm_init": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "lrcn_comm_destroy": (C.c_int, [C.c_void_p]),
    "lrcn_comm_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lrcn_set_embed_rows_buffer": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]),
    "lrcn_embed_grad_from_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "lrcn_allreduce_grads": (C.c_int, [C.c_void_p, P9, C.c_int]),
    "lrcn_comm_join": (C.c_int, [C.c_void_p]),
    "lrcn_train_step_dp": (C.c_int, [C.c_void_p, P9, P9, P9, P9, C.c_void_p, C.POINTER(C.c_float), C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                     C.c_int, C.c_int, C.POINTER(Dropout), C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                     C.POINTER(C.c_double)]),
    "lrcn_beam_search": (C.c_int, [C.c_void_p, P9, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_int32),
                                   C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    "lrcn_beam_search_batch": (C.c_int, [C.c_void_p, P9, C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int32),
                                         C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    "lrcn_vgg_load": (C.c_int, [C.c_void_p, P13, P13, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "lrcn_vgg_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    "lrcn_preprocess_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_void_p]),
    "lrcn_vgg_forward_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_void_p]),
    "lrcn_vgg_forward_u8_blocks": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_int, C.c_int, C.c_void_p]),
    "lrcn_set_average_image": (C.c_int, [C.c_void_p, C.c_void_p]),
    "lrcn_resize_crop_u8": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int),
                                      C.c_int, C.c_void_p]),
    "lrcn_normalize_features": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "lrcn_host_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "lrcn_host_free": (C.c_int, [C.c_void_p]),
    "lrcn_upload_crops": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_void_p)]),
    "lrcn_upload_wait": (C.c_int, [C.c_void_p]),
    "lrcn_profile": (C.c_int, [C.c_void_p, C.c_int]),
    "lrcn_profile_get": (C.c_int, [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "lrcn_profile_segment": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "lrcn_debug_stamps": (C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_int64]),
    "lrcn_bench_conv": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "lrcn_bench_gemm": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]),
    "lrcn_conv1_fused": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p]),
    "lrcn_conv3x3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                               C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "lrcn_vgg_set_wg_cap": (C.c_int, [C.c_void_p, C.c_int]),
    "lrcn_vgg_calibrate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float), C.c_float]),
    "lrcn_debug_route": (C.c_char_p, [C.c_void_p, C.c_int]),
    "lrcn_conv3x3_fp8": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
}


def build(force=False):
    """hipcc --offload-arch=gfx950 -> csrc/liblrcn_hip.so (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))] + [HEADER]
    stale = force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in srcs)
    if stale:
        subprocess.check_call(["make", "-s", "-j4", "-C", CSRC, "-f", os.path.join(CSRC, "Makefile")])
    return LIB_PATH


_LIB = None


def lib():
    global _LIB
    if _LIB is None:
        if not os.path.exists(LIB_PATH):
            raise LrcnError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                            "(there is no CPU fallback)" % LIB_PATH)
        # torch ships its own libamdhip64; device pointers and streams are shared with it, so its HIP runtime must be
        # the one this process uses: import torch BEFORE the library so the dynamic linker binds to that instance.
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(L, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if L.lrcn_abi_version() != LRCN_ABI_VERSION:
            raise LrcnError("%s implements ABI revision %d, this binding was written against %d (lrcn_config's layout differs)"
                            % (LIB_PATH, L.lrcn_abi_version(), LRCN_ABI_VERSION))
        _LIB = L
    return _LIB


def check(ctx_handle, rc):
    if rc != 0:
        msg = lib().lrcn_last_error(ctx_handle)
        raise LrcnError("liblrcn_hip error %d: %s" % (rc, msg.decode() if msg else "?"))
