#!/bin/bash
# Round 6 (GPU box): with the backward recurrence in K slices the LSTM chain is ~1 ms shorter -- does a higher convolution-grid cap pay now?
OUT=${1:-gpurun_out/r06_ab_cap.txt}
: > $OUT
tr() { python3 bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); s=d['roofline']['sub']; h=d.get('hw_held_in_timed_region') or {}; print('%.4f ms/step median %.4f conv_launch %.4f rec_fwd %.3f rec_bwd %.3f ms  sclk %s' % (d['ms_per_step'], d['ms_per_step_median'], d['roofline']['avg_launch_ms'], s['recurrence_weight_stream_fwd']['ms_per_step'], s['recurrence_weight_stream_bwd']['ms_per_step'], (h.get('sclk_mhz') or {}).get('median')))"; }
for i in 1 2; do
  echo "cap 224 slabs 4: $(LRCN_BWD_SLABS=4 tr)" >> $OUT
  echo "cap 232 slabs 3: $(LRCN_VGG_WG_CAP=232 LRCN_BWD_SLABS=3 tr)" >> $OUT
  echo "cap 228 slabs 3: $(LRCN_VGG_WG_CAP=228 LRCN_BWD_SLABS=3 tr)" >> $OUT
  echo "cap 216 slabs 5: $(LRCN_VGG_WG_CAP=216 LRCN_BWD_SLABS=5 tr)" >> $OUT
done
cat $OUT
