#!/bin/bash
# development sweep at 32 rows per rank (emulated rank of 8): one line per setting, ms per step and the average convolution launch
run() { ms=$(env "$@" python3 bench.py --emulate-world 8 --steps 96 --warmup 16 --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('%.4f conv_launch %.4f' % (d['ms_per_step'], d['roofline']['avg_launch_ms']))"); echo "$* -> $ms"; }
run A=0
run LRCN_8P_SPLITK=0
run LRCN_8P_SPLITK_MIN=200
for cap in 144 152 168 176; do run LRCN_VGG_WG_CAP=$cap; run LRCN_VGG_WG_CAP=$cap LRCN_LSTM_REC3=0; done
run A=0
run LRCN_LSTM_REC3=0
