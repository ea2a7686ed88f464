#!/bin/bash
# usage: tools/rocprof_stats.sh <name> <script.py> [args...]   (on the GPU box, from the repo root)
# kernel-trace + stats of `python3 script.py args` as CSV under gpurun_out/<name>/, then the per-kernel breakdown.
set -e
name=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$name
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o p -- python3 "$root/$1" "${@:2}" > "$out.log" 2>&1 || { tail -20 "$out.log"; exit 1; }
cd "$root"
grep -v "rocprofv3\|simple_timer\|amdgpu.ids" "$out.log" | tail -3
