#!/bin/bash
# usage (GPU box, repo root): tools/sweep_env.sh <ENV_NAME> "<v1 v2 ...>" <out-prefix> [bench.py args...]
# Same-box A/B/C...: one bench.py run per value of the environment variable, two rounds, ms_per_step (mean / median) per run.
name=$1; vals=$2; out=$3; shift 3
for round in 1 2; do
  for v in $vals; do
    env $name=$v python3 bench.py "$@" --no-cpu-baseline > gpurun_out/${out}_${v}_$round.json 2>/dev/null
    python3 -c "
import json; d=json.load(open('gpurun_out/${out}_${v}_$round.json')); print('$name=$v round $round: %.4f ms mean  %.4f median' % (d['ms_per_step'], d['ms_per_step_median']))"
  done
done
