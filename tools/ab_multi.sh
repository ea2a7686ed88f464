#!/bin/bash
# usage: tools/ab_multi.sh <VAR> <rounds> <v1> <v2> ... -- <bench.py args>   (GPU box, repo root)
# Same-box comparison of several values of one environment knob: `rounds` alternating passes over the values; per value the sorted
# ms_per_step of its runs (process-to-process spread on one box is bimodal at small batches: compare minima and medians, not single runs).
var=$1; rounds=$2; shift 2
vals=()
while [ "$1" != "--" ]; do vals+=("$1"); shift; done
shift
declare -A res
for i in $(seq 1 $rounds); do
  for v in "${vals[@]}"; do
    ms=$(env $var=$v python3 bench.py "$@" --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4f' % d['ms_per_step'])")
    res[$v]="${res[$v]} $ms"
  done
done
for v in "${vals[@]}"; do
  echo "$var=$v: $(echo ${res[$v]} | tr ' ' '\n' | sort -n | tr '\n' ' ')"
done
