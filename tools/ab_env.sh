#!/bin/bash
# usage: tools/ab_env.sh <VAR> <a> <b> <pairs> -- <bench.py args...>   (GPU box, repo root)
# Same-box A/B of one environment knob: alternating runs of bench.py, `pairs` of each, ms_per_step printed per run.
var=$1; a=$2; b=$3; pairs=$4; shift 5
for i in $(seq 1 $pairs); do
  for v in $a $b; do
    ms=$(env $var=$v python3 bench.py "$@" --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4f median %.4f conv_launch %.4f' % (d['ms_per_step'], d['ms_per_step_median'], d['roofline']['avg_launch_ms']))")
    echo "$var=$v  ms_per_step $ms"
  done
done
