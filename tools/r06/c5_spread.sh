#!/bin/bash
# Round 6: spread of the C5 fixture's training trajectory.  Each line of the output = one run of tools/c5_fixture.py.
# usage: tools/r06/c5_spread.sh OUT  (run on the GPU box)
OUT=${1:-gpurun_out/r06_c5_spread.txt}
: > $OUT
run() {  # label, env...
  lab=$1; shift
  for i in 1 2; do
    echo "== $lab run $i" >> $OUT
    env "$@" C5_STEPS=1500 timeout 300 python tools/c5_fixture.py 2>&1 | tail -1 | python -c "
import sys, json
l = sys.stdin.read().strip()
try:
    r = json.loads(l)
    print(json.dumps({k: r[k] for k in ('train_steps', 'train_deterministic', 'feat_scale', 'train_loss_trace', 'correct_f32', 'correct_fp8', 'fp8_vs_f32', 'bf16_vs_f32', 'fp8_vs_bf16')}))
except Exception as e:
    print('ERR', e, l[-400:])
" >> $OUT
  done
}
run "det scale 0.01" C5_DET=1
run "atomic scale 0.01 fixed 1500" C5_DET=0 C5_TARGET=-1 C5_MAX_STEPS=1500
run "det scale 0.03" C5_DET=1 C5_FEAT_SCALE=0.03
run "det scale 0.1" C5_DET=1 C5_FEAT_SCALE=0.1
run "atomic scale 0.1" C5_DET=0 C5_FEAT_SCALE=0.1
cat $OUT
