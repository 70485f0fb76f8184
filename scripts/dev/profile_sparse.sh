#!/usr/bin/env bash
# usage: scripts/dev/profile_sparse.sh <tag> <model> <quantization|none> <batch> [layers]  -> gpurun_out/<tag>_breakdown.txt
set -euo pipefail
tag="$1"; model="$2"; quant="$3"; batch="$4"; layers="${5:-8}"
cd /tmp && export TMPDIR=/tmp
out="$GRAFT_REPO_ROOT/gpurun_out/prof_$tag"
rm -rf "$out"
timeout -k 10 280 rocprofv3 --kernel-trace --output-format csv -d "$out" -- python3 "$GRAFT_REPO_ROOT/benchmarks/qwen480b_single_gpu.py" --model "$model" --quantization "$quant" --layers "$layers" --batch "$batch" --steps 8 > "$GRAFT_REPO_ROOT/gpurun_out/${tag}_prof.json" 2> "$GRAFT_REPO_ROOT/gpurun_out/${tag}_prof.err"
csv=$(find "$out" -name "*kernel_trace.csv" | head -1)
python3 "$GRAFT_REPO_ROOT/benchmarks/step_breakdown.py" "$csv" > "$GRAFT_REPO_ROOT/gpurun_out/${tag}_breakdown.txt"
rm -rf "$out"
head -40 "$GRAFT_REPO_ROOT/gpurun_out/${tag}_breakdown.txt"
