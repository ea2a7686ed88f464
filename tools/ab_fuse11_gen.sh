#!/bin/bash
# on the GPU box: kernel-trace stats of bench.py for both generations of the fused conv1 kernel
cd /tmp && export TMPDIR=/tmp
for g in ${GENS:-1 2}; do
  export LRCN_FUSE11_GEN=$g
  rm -rf /tmp/ab$g
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ab$g -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 3 --no-cpu-baseline > /tmp/ab$g.json 2> /tmp/ab$g.log
  echo "gen $g: $(python3 -c "import json;d=json.loads(open('/tmp/ab$g.json').read().strip().splitlines()[-1]);print(d['ms_per_step'], d['roofline']['frac'])")"
  grep -i "conv64" /tmp/ab$g/p_kernel_stats.csv | cut -c1-160
done
