#!/usr/bin/env bash
# usage: scripts/dev/profile_step.sh <tag> <bench.py args...>   -> gpurun_out/<tag>_breakdown.txt
set -euo pipefail
tag="$1"; shift
cd /tmp && export TMPDIR=/tmp
out="$GRAFT_REPO_ROOT/gpurun_out/prof_$tag"
rm -rf "$out"
timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d "$out" -- python3 "$GRAFT_REPO_ROOT/bench.py" "$@" > "$GRAFT_REPO_ROOT/gpurun_out/${tag}_prof_bench.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/${tag}_prof.err"
csv=$(find "$out" -name "*kernel_trace.csv" | head -1)
python3 "$GRAFT_REPO_ROOT/benchmarks/step_breakdown.py" "$csv" > "$GRAFT_REPO_ROOT/gpurun_out/${tag}_breakdown.txt"
rm -rf "$out"
head -25 "$GRAFT_REPO_ROOT/gpurun_out/${tag}_breakdown.txt"
