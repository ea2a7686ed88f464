#!/bin/bash
# usage (GPU box, repo root): tools/collect_profiles.sh <tag>     e.g. r02a
# Produces under gpurun_out/profiles_<tag>/ what profiles/ keeps per round: the rocprofv3 kernel-trace --stats summary of bench.py,
# the FETCH_SIZE and WRITE_SIZE PMC passes (separate runs, --kernel-trace only) reduced by tools/pmc_traffic.py, and the bench line.
set -e
tag=$1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/profiles_$tag
rm -rf "$out"; mkdir -p "$out"
cd "$root"
python3 bench.py > "$out/bench_line.json" 2> "$out/bench.err" || { tail "$out/bench.err"; exit 1; }
# the driver's own shape (5 warm-up + 20 timed steps) and the other BASELINE configs, one line each
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > "$out/bench_line_driver_shape.json" 2>> "$out/bench.err"
python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --inputs hbm > "$out/bench_line_driver_shape_inputs_hbm.json" 2>> "$out/bench.err"
python3 bench.py --emulate-world 8 --steps 192 --warmup 32 --no-cpu-baseline > "$out/bench_line_emulated_rank_of_8.json" 2>> "$out/bench.err"
bash tools/emulated_modes.sh > "$out/emulated_rank_of_8_update_modes.txt" 2>> "$out/bench.err"
python3 bench.py --emulate-world 8 --steps 96 --warmup 16 --no-cpu-baseline --replicated-update > "$out/bench_line_emulated_rank_of_8_replicated_update.json" 2>> "$out/bench.err"
python3 bench.py --emulate-world 8 --steps 96 --warmup 16 --no-cpu-baseline --vgg-chunk-images 0 > "$out/bench_line_emulated_rank_of_8_one_forward_per_step.json" 2>> "$out/bench.err"
python3 bench.py --emulate-world 8 --steps 20 --warmup 5 --no-cpu-baseline > "$out/bench_line_emulated_rank_of_8_driver_shape.json" 2>> "$out/bench.err"
python3 bench.py --emulate-world 4 --steps 48 --warmup 16 --no-cpu-baseline > "$out/bench_line_emulated_rank_of_4.json" 2>> "$out/bench.err"
python3 bench.py --emulate-world 2 --steps 40 --warmup 10 --no-cpu-baseline > "$out/bench_line_emulated_rank_of_2.json" 2>> "$out/bench.err"
python3 bench.py --config c2 --no-cpu-baseline > "$out/bench_line_c2.json" 2>> "$out/bench.err"
python3 bench.py --config c3 --no-cpu-baseline > "$out/bench_line_c3.json" 2>> "$out/bench.err"
python3 bench.py --config c5 > "$out/bench_line_c5.json" 2>> "$out/bench.err"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -o p -- python3 "$root/bench.py" --steps 25 --warmup 5 --no-cpu-baseline > "$out/bench_under_rocprof.json" 2> "$out/stats.log"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -o p -- python3 "$root/bench.py" --steps 8 --warmup 2 --no-cpu-baseline > /dev/null 2> "$out/fetch.log"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/write" -o p -- python3 "$root/bench.py" --steps 8 --warmup 2 --no-cpu-baseline > /dev/null 2> "$out/write.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats32" -o p -- python3 "$root/bench.py" --emulate-world 8 --steps 48 --warmup 16 --no-cpu-baseline > "$out/bench_under_rocprof_b32.json" 2> "$out/stats32.log"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/statsc2" -o p -- python3 "$root/bench.py" --config c2 --steps 20 --warmup 5 --no-cpu-baseline > "$out/bench_under_rocprof_c2.json" 2> "$out/statsc2.log"
LRCN_C5_LIGHT=1 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/statsc5" -o p -- python3 "$root/bench.py" --config c5 --steps 30 > "$out/bench_under_rocprof_c5.json" 2> "$out/statsc5.log"
LRCN_C5_LIGHT=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/fetchc5" -o p -- python3 "$root/bench.py" --config c5 --steps 20 > /dev/null 2> "$out/fetchc5.log"
LRCN_C5_LIGHT=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/writec5" -o p -- python3 "$root/bench.py" --config c5 --steps 20 > /dev/null 2> "$out/writec5.log"
cd "$root"
# matrix-pipe utilisation of the convolution family and of the LSTM chain's contractions (round 6): four SQ-counter passes of the headline command
tools/pmc_sq_passes.sh profiles_${tag}_sqbench bench.py --steps 8 --warmup 2 --no-cpu-baseline > "$out/sq.log" 2>&1 || tail -5 "$out/sq.log"
python3 tools/pmc_mfma.py gpurun_out/profiles_${tag}_sqbench_sq.json "$out/pmc_mfma_busy.json" > /dev/null || true
rm -f gpurun_out/profiles_${tag}_sqbench_p?.json gpurun_out/profiles_${tag}_sqbench_p?.log gpurun_out/profiles_${tag}_sqbench_sq.json
f32=$(find "$out/stats32" -name "*kernel_trace.csv" | head -1)
python3 tools/beside_vs_alone.py "$f32" 400 > "$out/beside_vs_alone_b32.txt" 2>&1 || true
python3 tools/step_chain.py "$f32" 3 > "$out/step_chain_b32.txt" 2>&1 || true
cp "$out"/stats/p_kernel_stats.csv "$out/kernel_stats.csv"
cp "$out"/stats32/p_kernel_stats.csv "$out/kernel_stats_b32.csv"
cp "$out"/statsc2/p_kernel_stats.csv "$out/kernel_stats_c2.csv"
cp "$out"/statsc5/p_kernel_stats.csv "$out/kernel_stats_c5.csv"
python3 tools/pmc_traffic.py "$out"/fetch/p_counter_collection.csv "$out"/write/p_counter_collection.csv "$out/pmc_traffic.json" > /dev/null
python3 tools/pmc_traffic.py "$out"/fetchc5/p_counter_collection.csv "$out"/writec5/p_counter_collection.csv "$out/pmc_traffic_c5.json" > /dev/null
rm -rf "$out/stats" "$out/stats32" "$out/statsc2" "$out/statsc5" "$out/fetch" "$out/write" "$out/fetchc5" "$out/writec5"   # the raw traces are tens of MB; the summaries are what is kept
tail -c 1200 "$out/bench_line.json"; echo; python3 - <<PY
import csv
rows=list(csv.DictReader(open("$out/kernel_stats.csv")))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:12]: print("%-88s %5s calls %9.1f us avg %5.1f%%"%(r['Name'][:88], r['Calls'], float(r['AverageNs'])/1e3, 100*float(r['TotalDurationNs'])/tot))
PY
python3 -c "import json; d=json.load(open('$out/pmc_traffic.json')); print({k:v for k,v in d.items() if k!='per_kernel'})"
