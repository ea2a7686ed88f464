#!/bin/bash
# usage (GPU box, repo root): tools/pmc_sq_passes.sh <name> <script.py> [args...]
# Four rocprofv3 --pmc passes (SQ wave / instruction / LDS / wait counters; --kernel-trace only, as gpurun requires) of `python3 script.py args`;
# per-kernel averages of every pass -> gpurun_out/<name>_sq.json (tools/pmc_summary.py per pass, merged).
set -e
name=$1; shift
P1="GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES"
P2="SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM"
P3="SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS"
P4="SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY"
i=0
for p in "$P1" "$P2" "$P3" "$P4"; do
  i=$((i+1))
  tools/pmc_pass.sh ${name}_p$i "$p" "$@" > /dev/null
  python3 tools/pmc_summary.py gpurun_out/${name}_p$i > gpurun_out/${name}_p$i.json
  rm -rf gpurun_out/${name}_p$i
done
python3 - <<PY
import json
out = {}
for i in range(1, 5):
    d = json.load(open("gpurun_out/${name}_p%d.json" % i))
    for k, v in d.items():
        o = out.setdefault(k, {"dispatches": v["dispatches"], "avg": {}})
        o["avg"].update(v["avg"])
json.dump(out, open("gpurun_out/${name}_sq.json", "w"), indent=1)
print("kernels:", len(out))
PY
