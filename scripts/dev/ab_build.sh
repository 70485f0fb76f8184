#!/usr/bin/env bash
# usage: scripts/dev/ab_build.sh <git-rev> <name>   -> ab/libvdamd_ops_<name>.so built from that revision's csrc/
# (run a benchmark against it with VDAMD_OPS_LIB=$GRAFT_REPO_ROOT/ab/libvdamd_ops_<name>.so: kernels compared inside
# one GPU call, since the same kernel measures +-5 % from one box to the next)
set -euo pipefail
rev="$1"; name="$2"
root="$(cd "$(dirname "$0")/../.." && pwd)"
wt="$(mktemp -d /tmp/abwt.XXXXXX)"
git -C "$root" worktree add --detach "$wt" "$rev" > /dev/null
( cd "$wt" && python -c "from vllm_distributed_amd import build; build.build()" > /dev/null )
mkdir -p "$root/ab"
cp "$wt/vllm_distributed_amd/_C/libvdamd_ops.so" "$root/ab/libvdamd_ops_$name.so"
git -C "$root" worktree remove --force "$wt"
echo "built ab/libvdamd_ops_$name.so from $rev"
