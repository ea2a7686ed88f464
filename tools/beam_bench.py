#!/usr/bin/env python3
"""Caption-generation throughput (BASELINE.json configs[4]: beam-search-5 decode), 1 GPU: captions/s of the batched
device-resident beam search (lrcn_beam_search_batch) and of the per-image decode the reference's control flow implies.
Synthetic model (E = H = 1000, V = 10640, random weights) and features; nword = 30.  Needs an MI355X."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import lrcn_amd  # noqa: E402
from lrcn_amd import lrcn as L  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    K, nword, V = 5, 30, 10640
    ctx = L.Context(1000, 1000, 1000, V, max_B=N * K, max_T=1, lstm_dtype=lrcn_amd.LRCN_BF16)
    param = L.initweights(ctx, seed=42)
    feats = (np.random.default_rng(0).standard_normal((N, 4096)) * 0.01).astype(np.float32)
    fj = L.to_jl(feats)
    L.beam_search_batch(ctx, param, fj, K, nword)
    torch.cuda.synchronize()
    # the decode returns N Python lists per call; with torch imported a full collection of the cyclic GC scans millions of objects
    # (measured: +13 ms on a 13 ms decode of 512 images, erratically by N).  Freeze what exists: later collections see only new objects.
    import gc
    gc.collect()
    gc.freeze()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        out = L.beam_search_batch(ctx, param, fj, K, nword)
    dt = (time.perf_counter() - t0) / reps
    print("batched: N=%d K=%d nword=%d  %.1f ms  %.0f captions/s (mean length %.1f)" % (N, K, nword, dt * 1e3, N / dt,
                                                                                        np.mean([len(t) for t, _ in out])))
    if os.environ.get("BEAM_BENCH_BATCH_ONLY"):  # profiling runs: only the batched decode
        return
    n1 = min(N, 8)
    t0 = time.perf_counter()
    for i in range(n1):
        L.beam_search(ctx, param, L.to_jl(feats[i:i + 1]), K, nword)
    dt1 = (time.perf_counter() - t0) / n1
    print("per image: %.1f ms per caption  %.0f captions/s" % (dt1 * 1e3, 1.0 / dt1))


if __name__ == "__main__":
    main()
