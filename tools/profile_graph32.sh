#!/bin/bash
# Kernel trace of the hipGraph-replayed training step at batch 32 (the launch-latency-bound configurations).
# usage: tools/profile_graph32.sh <tag> [config.json]
set -u
tag=${1:-prof32}
cfg=${2:-disentangled_simple.json}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o trace -- python3 "$root/bench.py" --config $cfg --batch 32 --steps 20 --warmup 3 --no-cpu-baseline --no-kernel-table --no-extras > "$out/trace.log" 2>&1 || exit 1
echo done
