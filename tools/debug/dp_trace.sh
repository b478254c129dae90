#!/bin/bash
# kernel traces of two variants of the one-rank data-parallel step (tools/debug/dp_overhead_probe.py --only ...)
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for v in "staged, collectives skipped" "two stages, collectives skipped" "single graph"; do
  tag=$(echo "$v" | tr ' ,' '__')
  out=$root/gpurun_out/dp_trace/$tag; mkdir -p "$out"
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$out" -o trace -- python3 "$root/tools/debug/dp_overhead_probe.py" --steps 10 --only "$v" > "$out/log.txt" 2>&1 || exit 1
  python3 "$root/tools/debug/timeline_gaps.py" $(find "$out" -name '*kernel_trace.csv' | head -1) 6 > "$out/gaps.txt" 2>&1
  python3 "$root/tools/graph_kernel_table.py" $(find "$out" -name '*kernel_trace.csv' | head -1) 6 > "$out/table.csv" 2>&1
done
echo done
