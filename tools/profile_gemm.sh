#!/bin/bash
# Counter passes over the largest Linear-layer GEMM (three kinds), each group of counters in its own run.
# usage: tools/profile_gemm.sh <tag> [precision]
set -u
tag=${1:-gemmprof}
prec=${2:-fp32}
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp
args="$root/tools/bench_gemm.py --layers 1 --reps 3 --precision $prec"
timeout -s KILL 120 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum --output-format csv -d "$out/p1" -- python3 $args > "$out/p1.log" 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum --output-format csv -d "$out/p2" -- python3 $args > "$out/p2.log" 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_LEVEL_VMEM SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$out/p3" -- python3 $args > "$out/p3.log" 2>&1 || exit 3
timeout -s KILL 120 rocprofv3 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum --output-format csv -d "$out/p4" -- python3 $args > "$out/p4.log" 2>&1 || echo "pass 4 failed"
echo done
