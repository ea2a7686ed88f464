#!/bin/bash
# Round 6 A/B (GPU box): C5 captions/s, library at tools/r06/_base (round 5's decode step) against the working tree's; 1024 images per pass as
# the bench line; alternating pairs.     usage: tools/r06/ab_c5.sh OUT [pairs] [images]
OUT=${1:-gpurun_out/r06_ab_c5.txt}; PAIRS=${2:-3}; N=${3:-1024}; BASE=${4:-_base}
: > $OUT
c5() { python3 tools/caption_bench.py --images $N --chunk $N --no-cpu-baseline --no-fixture 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.0f captions/s  %.2f ms/pass  conv frac %.3f' % (d['value'], d['ms_per_step'], d['roofline']['frac']))"; }
for i in $(seq 1 $PAIRS); do
  echo "C5 base: $(LRCN_HIP_LIB=$PWD/tools/r06/$BASE/liblrcn_hip.so c5)" >> $OUT
  echo "C5 new : $(c5)" >> $OUT
done
cat $OUT
