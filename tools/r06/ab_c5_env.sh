#!/bin/bash
# usage: tools/r06/ab_c5_env.sh OUT VAR a b pairs      C5 caption bench (1024 images, one chunk) with VAR=a / VAR=b alternating
OUT=$1; VAR=$2; A=$3; B=$4; PAIRS=${5:-3}
c5() { python3 tools/caption_bench.py --images ${C5_N:-2048} --chunk ${C5_N:-2048} --no-cpu-baseline --no-fixture 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.0f captions/s  %.2f ms/pass  conv frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))"; }
for i in $(seq 1 $PAIRS); do
  echo "C5 $VAR=$A: $(env $VAR=$A bash -c "$(declare -f c5); c5")" >> $OUT
  echo "C5 $VAR=$B: $(env $VAR=$B bash -c "$(declare -f c5); c5")" >> $OUT
done
