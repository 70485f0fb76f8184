#!/usr/bin/env bash
# usage: scripts/dev/qwen_ab.sh <batch> "<ENV=.. ENV=..>" ...   one decode measurement per environment set
set -euo pipefail
batch="$1"; shift
for envs in "$@"; do
  echo "== batch $batch env: $envs"
  env $envs timeout -k 10 200 python benchmarks/qwen480b_single_gpu.py --batch "$batch" --steps 48 2>/dev/null | grep -o '"decode_ms_per_step": [0-9.]*'
done
