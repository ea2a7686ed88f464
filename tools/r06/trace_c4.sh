#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tc4
rocprofv3 --kernel-trace --output-format csv -d /tmp/tc4 -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 16 --warmup 6 --no-cpu-baseline > /tmp/tc4.json 2> /tmp/tc4.log
f=$(find /tmp/tc4 -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/r06/two_chains.py $f 6 > $GRAFT_REPO_ROOT/gpurun_out/r06_two_chains_c4.txt 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/r06_two_chains_c4.txt
