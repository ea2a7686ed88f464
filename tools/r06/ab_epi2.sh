#!/bin/bash
# Round 6 A/B (GPU box): the C4 training step with the forward cell epilogue (LRCN_LSTM_EPI=f) now that the fused update keeps the
# gate-interleaved copies current (no full shadow pass per step) against the two-launch recurrence; 5 alternating pairs, driver shape too.
OUT=${1:-gpurun_out/r06_ab_epi2.txt}
: > $OUT
tr() { python3 bench.py "$@" --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); s=d['roofline']['sub']; h=d.get('hw_held_in_timed_region') or {}; print('%.4f ms/step median %.4f conv_launch %.4f rec_fwd %.3f rec_bwd %.3f ms  sclk %s W %s' % (d['ms_per_step'], d['ms_per_step_median'], d['roofline']['avg_launch_ms'], s['recurrence_weight_stream_fwd']['ms_per_step'], s['recurrence_weight_stream_bwd']['ms_per_step'], (h.get('sclk_mhz') or {}).get('median'), (h.get('socket_power_w') or {}).get('median')))"; }
for i in 1 2 3 4 5; do
  echo "C4 epi off: $(tr --steps 40 --warmup 10)" >> $OUT
  echo "C4 epi f  : $(LRCN_LSTM_EPI=f tr --steps 40 --warmup 10)" >> $OUT
done
for i in 1 2; do
  echo "C4 driver shape epi off: $(tr --gpus 1 --steps 20 --warmup 5)" >> $OUT
  echo "C4 driver shape epi f  : $(LRCN_LSTM_EPI=f tr --gpus 1 --steps 20 --warmup 5)" >> $OUT
done
cat $OUT
