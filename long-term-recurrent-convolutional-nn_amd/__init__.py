"""MI355X-native LRCN hot path (training step + beam-search decode) behind the reference's function surface.

Import as `lrcn_amd` (see /lrcn_amd.py): the directory name carries the reference repository's hyphens and is not a
Python identifier.  Layout: csrc/ = HIP kernels + the C ABI (liblrcn_hip.so); _lib.py = ctypes binding;
lrcn.py = mirror of lrcn.jl's functions; dp.py = data-parallel step over torch.distributed (RCCL).
"""
from . import _lib  # noqa: F401
from ._lib import BOS, CNNOUT, EOS, LRCN_BF16, LRCN_F32, LRCN_FP8, UNK, LrcnError, build  # noqa: F401
