#!/bin/bash
# on the GPU box: kernel-trace stats of bench.py with the default library and with LRCN_HIP_LIB=$1; prints the kernels matching $2
cd /tmp && export TMPDIR=/tmp
for v in base var base var; do
  if [ $v = var ]; then export LRCN_HIP_LIB=$GRAFT_REPO_ROOT/$1; else unset LRCN_HIP_LIB; fi
  rm -rf /tmp/ab_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab_$v -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 4 --no-cpu-baseline > /tmp/ab_$v.json 2> /tmp/ab_$v.log
  echo "$v: $(grep -i "$2" /tmp/ab_$v/p_kernel_stats.csv | cut -d, -f1-5 | cut -c1-140)"
done
