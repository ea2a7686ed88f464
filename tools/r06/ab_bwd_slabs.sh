#!/bin/bash
# Round 6 A/B (GPU box): the C4 step with the backward dh GEMM cut into K slices summed by the cell kernel (LRCN_BWD_SLABS=4) against the default
OUT=${1:-gpurun_out/r06_ab_bwd_slabs.txt}
: > $OUT
tr() { python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); s=d['roofline']['sub']; h=d.get('hw_held_in_timed_region') or {}; print('%.4f ms/step median %.4f conv_launch %.4f rec_fwd %.3f rec_bwd %.3f ms  sclk %s W %s' % (d['ms_per_step'], d['ms_per_step_median'], d['roofline']['avg_launch_ms'], s['recurrence_weight_stream_fwd']['ms_per_step'], s['recurrence_weight_stream_bwd']['ms_per_step'], (h.get('sclk_mhz') or {}).get('median'), (h.get('socket_power_w') or {}).get('median')))"; }
for i in 1 2 3 4; do
  echo "C4 default      : $(tr)" >> $OUT
  echo "C4 bwd slabs 4  : $(LRCN_BWD_SLABS=4 tr)" >> $OUT
  echo "C4 slabs 4+epi f: $(LRCN_BWD_SLABS=4 LRCN_LSTM_EPI=f tr)" >> $OUT
done
cat $OUT
