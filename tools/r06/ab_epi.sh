#!/bin/bash
# Round 6 A/B (GPU box): the forward cell epilogue, old form (tools/r06/_base: one unit per thread and iteration) vs new (four units per
# thread, operands requested before the staging).  (1) C5 captions/s, 3 alternating pairs; (2) the C4 training step with
# LRCN_LSTM_EPI unset / f / 1 on the new library, 3 rounds.
OUT=${1:-gpurun_out/r06_ab_epi.txt}
: > $OUT
c5() { python3 tools/caption_bench.py --images 1024 --chunk 1024 --no-cpu-baseline --no-fixture 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.0f captions/s  %.2f ms/pass' % (d['value'], d['ms_per_step']))"; }
for i in 1 2 3; do
  echo "C5 base: $(LRCN_HIP_LIB=$PWD/tools/r06/_base/liblrcn_hip.so c5)" >> $OUT
  echo "C5 new : $(c5)" >> $OUT
done
tr() { python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); s=d['roofline']['sub']; print('%.4f ms/step median %.4f conv_launch %.4f rec_fwd %.3f rec_bwd %.3f ms  sclk %s W %s' % (d['ms_per_step'], d['ms_per_step_median'], d['roofline']['avg_launch_ms'], s['recurrence_weight_stream_fwd']['ms_per_step'], s['recurrence_weight_stream_bwd']['ms_per_step'], (d.get('hw_held_in_timed_region') or {}).get('sclk_mhz'), (d.get('hw_held_in_timed_region') or {}).get('socket_power_w')))"; }
for i in 1 2 3; do
  echo "C4 epi off: $(tr)" >> $OUT
  echo "C4 epi f  : $(LRCN_LSTM_EPI=f tr)" >> $OUT
  echo "C4 epi 1  : $(LRCN_LSTM_EPI=1 tr)" >> $OUT
done
cat $OUT
