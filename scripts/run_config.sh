#!/usr/bin/env bash
# Bring up one of the five benchmark deployments on this host (head + `remote`
# worker processes, as the compose files do across hosts), run the HTTP serving
# benchmark against it and tear everything down.
#
#   scripts/run_config.sh 1   OPT-125m, head + 1 worker, PP=2, CPU / gloo (plumbing check)
#   scripts/run_config.sh 2   Llama-3-8B, TP=1, one GPU
#   scripts/run_config.sh 3   Llama-3-8B, head (1 GPU) + 7 single-GPU workers, TP=8
#   scripts/run_config.sh 4   Llama-3-70B, TP=8 (head owns all 8 GPUs)
#   scripts/run_config.sh 5   Llama-3-70B, TP=4 x PP=2 (head 4 GPUs + one 4-GPU worker node)
#   scripts/run_config.sh 6   Llama-3-70B, TP=1: the headline model on one MI355X (bf16 weights fit in 288 GB)
#
# Environment: PORT (8000), NUM_PROMPTS (256), CONCURRENCY (64), INPUT_LEN (128), OUTPUT_LEN (64),
# OUTPUT_JITTER (0: every request the same length; 0.5: lengths uniform in 32..96, requests desynchronised),
# REQUEST_RATE (0: all requests released at once; > 0: Poisson arrivals per second).
set -euo pipefail
cfg="${1:?config number 1-5}"
here="$(cd "$(dirname "$0")/.." && pwd)"
port="${PORT:-8000}"
export VLLM_HOST_IP=127.0.0.1 VLLM_RPC_SECRET="${VLLM_RPC_SECRET:-bench-secret}"
export GLOO_SOCKET_IFNAME="${GLOO_SOCKET_IFNAME:-lo}" NCCL_SOCKET_IFNAME="${NCCL_SOCKET_IFNAME:-lo}"
export HSA_ENABLE_IPC_MODE_LEGACY=0
pids=()
cleanup() { for p in "${pids[@]:-}"; do kill "$p" 2>/dev/null || true; done; wait 2>/dev/null || true; }
trap cleanup EXIT

serve() { python "$here/launch.py" serve "$@" --host 127.0.0.1 --port "$port" --load-format dummy & pids+=($!); }
remote() { python "$here/launch.py" remote 127.0.0.1 "$@" & pids+=($!); }

case "$cfg" in
  1) model=opt-125m; vocab=50272
     VLLM_HEAD_LOCAL_DEVICES=1 serve "$model" -pp 2 --device cpu --dtype float32 --max-model-len 512
     remote --num-devices 1 ;;
  2) model=llama-3-8b; vocab=128256
     serve "$model" -tp 1 --max-model-len 4096 ;;
  3) model=llama-3-8b; vocab=128256
     VLLM_HEAD_LOCAL_DEVICES=1 serve "$model" -tp 8 --max-model-len 4096
     for i in 1 2 3 4 5 6 7; do remote --num-devices 1 --device-offset "$i"; done ;;
  4) model=llama-3-70b; vocab=128256
     serve "$model" -tp 8 --max-model-len 4096 ;;
  5) model=llama-3-70b; vocab=128256
     VLLM_HEAD_LOCAL_DEVICES=4 serve "$model" -tp 4 -pp 2 --max-model-len 4096
     remote --num-devices 4 --device-offset 4 ;;
  6) model=llama-3-70b; vocab=128256
     serve "$model" -tp 1 --max-model-len 4096 --max-num-seqs "${MAX_NUM_SEQS:-1024}" ;;
  *) echo "unknown config $cfg" >&2; exit 2 ;;
esac

for _ in $(seq 1 600); do
  if python - "$port" <<'PY'
import sys, urllib.request
try:
    urllib.request.urlopen(f"http://127.0.0.1:{sys.argv[1]}/health", timeout=2)
except Exception:
    sys.exit(1)
PY
  then break; fi
  kill -0 "${pids[0]}" 2>/dev/null || { echo "server exited during start-up" >&2; exit 1; }
  sleep 1
done

python "$here/benchmarks/serve_bench.py" --base-url "http://127.0.0.1:$port" --model "$model" \
  --vocab-size "$vocab" --num-prompts "${NUM_PROMPTS:-256}" --concurrency "${CONCURRENCY:-64}" \
  --input-len "${INPUT_LEN:-128}" --output-len "${OUTPUT_LEN:-64}" --output-len-jitter "${OUTPUT_JITTER:-0}" --request-rate "${REQUEST_RATE:-0}"
