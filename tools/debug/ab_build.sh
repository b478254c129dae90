#!/bin/bash
# A/B of a compile-time experiment on the GPU box: micro-benchmarks with the shipped library, then with one rebuilt
# with the given extra flags.   usage: tools/debug/ab_build.sh <tag> "<extra flags>"
set -u
tag=$1; extra=$2
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag; mkdir -p $out
cd $root
run() {
  timeout -k 10 200 python tools/bench_gemm.py --reps 30 > $out/gemm_$1.log 2>&1 || return 1
  timeout -k 10 200 python tools/bench_conv.py --reps 10 > $out/conv_$1.log 2>&1 || return 1
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --no-kernel-table > $out/bench_$1.json 2> $out/bench_$1.err || return 1
}
run A || exit 1
cd disentangled-vae_amd/csrc && rm -f gemm.o conv.o && make -j16 EXTRA="$extra" > $out/make.log 2>&1 || exit 2
cd $root
run B || exit 3
echo done
