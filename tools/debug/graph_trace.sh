#!/bin/bash
# Kernel trace of the hipGraph-replayed default step, per-kernel table of the last replays.
# usage: tools/debug/graph_trace.sh <tag> [bench args]     (DVAE_HIP_LIB in the environment picks another build)
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag; mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d "$out/trace" -o trace -- python3 "$root/bench.py" --steps 12 --warmup 3 --no-cpu-baseline --no-kernel-table --no-extras "$@" > "$out/trace.log" 2>&1 || exit 1
python3 "$root/tools/graph_kernel_table.py" $(find "$out/trace" -name '*kernel_trace.csv' | head -1) 8 > "$out/table.csv" || exit 2
echo done
