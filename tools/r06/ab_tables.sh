#!/bin/bash
# Round 6 A/B (GPU box): C5 with the decode's input-projection tables on / off, alternating pairs (bench configuration: 2048 images per pass)
OUT=${1:-gpurun_out/r06_ab_tables.txt}
: > $OUT
tools/r06/ab_c5_env.sh $OUT LRCN_DECODE_TABLES 0 1 ${2:-3}
cat $OUT
