#!/bin/bash
# usage (GPU box, repo root): tools/collect_pmc_only.sh <tag>
# The counter passes of tools/collect_profiles.sh alone (FETCH_SIZE / WRITE_SIZE of the headline command and of C5, the four SQ passes of the
# headline command): what has to be re-taken when only the kernel sources' TEXT changed (the digest that bench.py checks covers comments too).
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/profiles_$tag
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -o p -- python3 "$root/bench.py" --steps 8 --warmup 2 --no-cpu-baseline > /dev/null 2> "$out/fetch.log"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write" -o p -- python3 "$root/bench.py" --steps 8 --warmup 2 --no-cpu-baseline > /dev/null 2> "$out/write.log"
LRCN_C5_LIGHT=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetchc5" -o p -- python3 "$root/bench.py" --config c5 --steps 20 > /dev/null 2> "$out/fetchc5.log"
LRCN_C5_LIGHT=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/writec5" -o p -- python3 "$root/bench.py" --config c5 --steps 20 > /dev/null 2> "$out/writec5.log"
cd "$root"
tools/pmc_sq_passes.sh profiles_${tag}_sqbench bench.py --steps 8 --warmup 2 --no-cpu-baseline > "$out/sq.log" 2>&1 || tail -5 "$out/sq.log"
python3 tools/pmc_mfma.py gpurun_out/profiles_${tag}_sqbench_sq.json "$out/pmc_mfma_busy.json" > /dev/null
python3 tools/pmc_traffic.py "$out"/fetch/p_counter_collection.csv "$out"/write/p_counter_collection.csv "$out/pmc_traffic.json" > /dev/null
python3 tools/pmc_traffic.py "$out"/fetchc5/p_counter_collection.csv "$out"/writec5/p_counter_collection.csv "$out/pmc_traffic_c5.json" > /dev/null
rm -rf "$out/fetch" "$out/write" "$out/fetchc5" "$out/writec5" gpurun_out/profiles_${tag}_sqbench_p?.json gpurun_out/profiles_${tag}_sqbench_sq.json
python3 -c "import json; [print(f, json.load(open('$out/'+f))['csrc_digest']) for f in ('pmc_traffic.json','pmc_traffic_c5.json','pmc_mfma_busy.json')]"
