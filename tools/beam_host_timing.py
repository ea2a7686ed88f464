#!/usr/bin/env python3
"""Kernel-development aid: host-side anatomy of one lrcn_beam_search_batch call (ctypes call vs result unpacking)."""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import lrcn_amd  # noqa: E402
from lrcn_amd import lrcn as L  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
K, nword, V = 5, 30, 10640
ctx = L.Context(1000, 1000, 1000, V, max_B=N * K, max_T=1, lstm_dtype=lrcn_amd.LRCN_BF16)
param = L.initweights(ctx, seed=42)
fj = L.to_jl((np.random.default_rng(0).standard_normal((N, 4096)) * 0.01).astype(np.float32))
Lh = nword + 2
out = (C.c_int32 * (N * Lh))()
n = (C.c_int * N)()
p = (C.c_float * N)()
fresh = len(sys.argv) > 2 and sys.argv[2] == "fresh"   # fresh result arrays per call, as lrcn.beam_search_batch makes them
for it in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if fresh:
        out = (C.c_int32 * (N * Lh))()
        n = (C.c_int * N)()
        p = (C.c_float * N)()
    ta = time.perf_counter()
    ctx._call("lrcn_beam_search_batch", L._p9(param), L._ptr(fj), N, K, nword, out, n, p)
    t1 = time.perf_counter()
    res = [(list(out[i * Lh:i * Lh + n[i]]), p[i]) for i in range(N)]
    t2 = time.perf_counter()
    print("N=%d alloc %.2f ms  call %.2f ms  unpack %.2f ms" % (N, (ta - t0) * 1e3, (t1 - ta) * 1e3, (t2 - t1) * 1e3))
