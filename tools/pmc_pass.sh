#!/bin/bash
# usage: tools/pmc_pass.sh <name> "<COUNTER1 COUNTER2 ...>" <script.py> [args...]     (GPU box, repo root)
# One rocprofv3 --pmc pass (with --kernel-trace only: gpurun refuses --pmc together with other trace domains) of
# `python3 script.py args`, CSV under gpurun_out/<name>/; summarise with tools/pmc_summary.py.
set -e
name=$1; ctrs=$2; shift 2
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$name
rm -rf "$out"; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d "$out" -o p -- python3 "$root/$1" "${@:2}" > "$out.log" 2>&1 || { tail -20 "$out.log"; exit 1; }
cd "$root"
ls "$out" | head
