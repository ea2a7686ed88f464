#!/usr/bin/env python3
"""Reduces the four SQ passes of tools/pmc_sq_passes.sh (rocprofv3 --pmc, --kernel-trace only, of `bench.py --steps 8 --warmup 2
--no-cpu-baseline`) to the matrix-pipe utilisation of the two contraction families of the headline step (north_star: "rocprof-reported MFMA
utilisation"):
  conv  -- the 12 convolution launches per step (conv64f_kernel, conv64_kernel, gemm8p_kernel<*, CONV3, *>): the dominant kernels;
  lstm  -- the LSTM chain's contractions (gemm8p_kernel<*, PLAIN, *> / gemm_glds / gemm_skinny as the step launches them).
mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 * SQ_BUSY_CU_CYCLES): matrix-pipe busy cycles summed over SIMDs against the busy CUs' cycles
(4 SIMDs per CU), summed over the family's launches (i.e. time-weighted).  Under --pmc every kernel runs ALONE on the chip (counters
serialise the streams), so this is the instruction streams' own utilisation -- not what the two chains leave each other inside the step.
The file carries the digest of csrc/ it was measured on; bench.py quotes it (roofline.mfma_busy) only while that digest matches.
Usage: tools/pmc_mfma.py gpurun_out/<name>_sq.json out.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def family(k):
    if "conv64_kernel" in k or "conv64f_kernel" in k:
        return "conv"
    if "gemm8p_kernel<" in k or "gemm_glds_kernel<" in k:
        args = k[k.index("<") + 1:k.index(">")].split(",")
        return "conv" if args[4].strip() == "1" else "lstm"   # AMODE == GEMM_A_CONV3
    if "gemm_skinny_kernel" in k or "lstm_rec_" in k:
        return "lstm"
    return None


def main():
    src, out = sys.argv[1:3]
    d = json.load(open(src))
    fam = {}
    for k, v in d.items():
        f = family(k)
        if f is None:
            continue
        o = fam.setdefault(f, {"launches": 0, "sum": {}, "kernels": {}})
        n = v["dispatches"]
        o["launches"] += n
        for c, x in v["avg"].items():
            o["sum"][c] = o["sum"].get(c, 0.0) + x * n
        a = v["avg"]
        o["kernels"][k[:100]] = {"launches": n, "mfma_busy": a["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * a["SQ_BUSY_CU_CYCLES"]) if a.get("SQ_BUSY_CU_CYCLES") else None,
                                 "gpu_cycles_per_launch": a.get("GRBM_GUI_ACTIVE")}
    res = {"what": "rocprofv3 --pmc SQ passes (tools/pmc_sq_passes.sh) of `bench.py --steps 8 --warmup 2 --no-cpu-baseline` (C4 step, 256 rows, bf16); "
                   "kernels serialised by the counters: each family's own utilisation, not the step's contention",
           "families": {}}
    for f, o in fam.items():
        s = o["sum"]
        g = lambda c: s.get(c, 0.0)
        wave = g("SQ_WAVE_CYCLES") or 1.0
        mf = g("SQ_INSTS_MFMA") or 1.0
        res["families"][f] = {
            "launches_counted": o["launches"],
            "mfma_busy": g("SQ_VALU_MFMA_BUSY_CYCLES") / (4.0 * g("SQ_BUSY_CU_CYCLES")) if g("SQ_BUSY_CU_CYCLES") else None,
            "waves_parked (SQ_WAIT_ANY / SQ_WAVE_CYCLES)": g("SQ_WAIT_ANY") / wave,
            "issue_stall (SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES)": g("SQ_WAIT_INST_ANY") / wave,
            "issuing (SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES)": g("SQ_ACTIVE_INST_ANY") / wave,
            "lds_wait (SQ_WAIT_INST_LDS / SQ_WAVE_CYCLES)": g("SQ_WAIT_INST_LDS") / wave,
            "lds_bank_conflict_of_lds_active": g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE") if g("SQ_LDS_IDX_ACTIVE") else None,
            "valu_per_mfma": g("SQ_INSTS_VALU") / mf, "salu_per_mfma": g("SQ_INSTS_SALU") / mf, "vmem_per_mfma": g("SQ_INSTS_VMEM") / mf,
            "lds_per_mfma": g("SQ_INSTS_LDS") / mf,
            "kernels": o["kernels"],
        }
    import bench
    res["csrc_digest"] = bench.csrc_digest()
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps({f: {k: v for k, v in r.items() if k != "kernels"} for f, r in res["families"].items()}, indent=1))


if __name__ == "__main__":
    main()
