#!/usr/bin/env python3
"""Kernel-development aid: TFLOP/s of bf16 NT GEMMs through the library's dispatch. Needs an MI355X.
Usage: tools/gemm_bench.py M,N,K [M,N,K ...]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import lrcn_amd  # noqa: E402
from lrcn_amd import lrcn as L  # noqa: E402


def main():
    shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [(4096, 4096, 4096), (8192, 8192, 8192)]
    ctx = L.Context(8, 8, 8, 17, max_B=2, max_T=1, lstm_dtype=lrcn_amd.LRCN_BF16, vgg_dtype=lrcn_amd.LRCN_BF16)
    lib = lrcn_amd._lib.lib()
    for M, N, K in shapes:
        ms = C.c_double()
        lrcn_amd._lib.check(ctx._h, lib.lrcn_bench_gemm(ctx._h, M, N, K, 10, C.byref(ms)))
        print("M=%6d N=%6d K=%6d  %8.3f ms  %7.1f TF" % (M, N, K, ms.value, 2.0 * M * N * K / ms.value / 1e9))


if __name__ == "__main__":
    main()
