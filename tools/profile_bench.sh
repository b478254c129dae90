#!/bin/bash
# Profiles the default bench command on the GPU box: one kernel-trace pass and three counter passes
# (counters in their own runs, as the profiler requires).  Output under gpurun_out/<tag>/.
# usage: tools/profile_bench.sh <tag> [extra bench.py arguments, e.g. --precision bf16]
set -u
tag=${1:-prof}
shift || true
extra="$*"
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
args="--steps 10 --warmup 3 --no-cpu-baseline --no-kernel-table --no-extras --graph 0 $extra"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -o trace -- python3 "$root/bench.py" $args > "$out/trace.log" 2>&1 || exit 1
pargs="--steps 2 --warmup 1 --no-cpu-baseline --no-kernel-table --no-extras --graph 0 $extra"
timeout -s KILL 240 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc_mfma" -- python3 "$root/bench.py" $pargs > "$out/pmc_mfma.log" 2>&1 || exit 2
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/pmc_fetch" -- python3 "$root/bench.py" $pargs > "$out/pmc_fetch.log" 2>&1 || exit 3
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/pmc_write" -- python3 "$root/bench.py" $pargs > "$out/pmc_write.log" 2>&1 || exit 4
grep -h '^{"metric"' "$out/trace.log" | tail -1 > "$out/bench_under_rocprof.json"
echo done
