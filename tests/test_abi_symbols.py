"""CPU: liblrcn_hip.so loads and exports every symbol include/lrcn.h declares (no compute calls)."""
import ctypes
import os
import re

import lrcn_amd
from lrcn_amd import _lib


def declared_symbols():
    txt = open(_lib.HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(lrcn_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lrcn_amd.build()
    assert os.path.exists(_lib.LIB_PATH)
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(L, n), "missing export: " + n
    # the Python binding covers exactly the declared set
    assert sorted(_lib.SIGNATURES) == names


def test_param_sizes_and_version_without_gpu():
    L = _lib.lib()
    assert L.lrcn_version().startswith(b"lrcn-hip")
    s = (ctypes.c_int64 * 9)()
    assert L.lrcn_param_sizes(1000, 1000, 1000, 10640, s) == 0
    assert sum(s) == 39846640  # C4 parameter count
    assert L.lrcn_param_sizes(512, 512, 512, 2540, s) == 0
    assert L.lrcn_param_sizes(8, 8, 7, 17, s) != 0  # odd H2 is rejected (lrcn.jl:496-505 needs 2h == H2)


def test_product_package_does_not_import_the_oracle():
    import sys
    pkg = os.path.dirname(_lib.__file__)
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("the CPU oracle", "").replace("CPU oracle", "") or f == "_lib.py", (root, f)
