#!/bin/bash
# Debug helper: run the SPMD test script of tests/test_comm_gpu.py with N ranks sharing GPU 0.
# usage: scripts/dev/spmd_shared.sh N [ENV=VALUE ...]
N=${1:-2}; shift
PORT=$((20000 + RANDOM % 20000))
python - > /tmp/spmd_script.py <<'PY'
import re
src = open("tests/test_comm_gpu.py").read()
m = re.search(r'SPMD_SCRIPT = r"""(.*?)"""', src, re.S)
print(m.group(1).format(root="."))
PY
pids=()
for r in $(seq 0 $((N-1))); do
  env RANK=$r LOCAL_RANK=$r WORLD_SIZE=$N MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT VLLM_LOGGING_LEVEL=INFO \
      GLOO_SOCKET_IFNAME=lo VDAMD_SHARE_DEVICE=1 HSA_ENABLE_IPC_MODE_LEGACY=0 LOGPROBS=8 "$@" \
      timeout -k 5 100 python /tmp/spmd_script.py > gpurun_out/spmd_rank$r.log 2>&1 &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=$?; done
exit $rc
