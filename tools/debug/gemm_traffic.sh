#!/bin/bash
# L2 hit/miss and fabric read bytes per launch of the Linear-layer GEMMs (tools/bench_gemm.py), two counter passes.
# usage: tools/debug/gemm_traffic.sh <tag> [precision] [layers]
set -u
tag=${1:-gemmtraffic}
prec=${2:-bf16}
layers=${3:-2}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
args="$root/tools/bench_gemm.py --layers $layers --reps 3 --precision $prec"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE TCC_HIT_sum --output-format csv -d "$out/p1" -- python3 $args > "$out/p1.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_MISS_sum TCC_REQ_sum WRITE_SIZE --output-format csv -d "$out/p2" -- python3 $args > "$out/p2.log" 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/kt" -- python3 $args > "$out/kt.log" 2>&1 || exit 3
echo done
