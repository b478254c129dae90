#!/bin/bash
# A/B of two builds of the library on one GPU box: the default bench, alternating A (in-tree) and B (the build named
# by $1), three times each.   usage: tools/debug/ab_lib.sh <other .so> [tag] [extra bench args]
set -u
other=$(realpath "$1"); tag=${2:-ab}; shift 2 || shift 1
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=$root/gpurun_out/$tag; mkdir -p "$out"
cd "$root"
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --no-kernel-table "$@" > "$out/A$i.json" 2> "$out/A$i.err" || exit 1
  DVAE_HIP_LIB=$other timeout -k 10 200 python bench.py --no-cpu-baseline --no-extras --no-kernel-table "$@" > "$out/B$i.json" 2> "$out/B$i.err" || exit 2
done
for f in "$out"/A?.json "$out"/B?.json; do echo "$(basename $f) $(grep -o '"ms_per_step": [0-9.]*' $f)"; done
