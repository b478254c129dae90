#!/bin/bash
# A/B of an environment switch on one box: usage tools/debug/ab_env.sh <tag> <VAR> "<config batch> ..."
set -u
out=gpurun_out/$1; var=$2; shift 2
mkdir -p "$out"
for cb in "$@"; do
  set -- $cb; cfg=$1; b=$2
  for f in 0 1; do
    env $var=$f timeout -k 10 200 python bench.py --config $cfg --batch $b --no-cpu-baseline --no-kernel-table --no-extras \
      > "$out/${cfg%.json}_${b}_$f.json" 2> "$out/${cfg%.json}_${b}_$f.err" || { tail -5 "$out/${cfg%.json}_${b}_$f.err"; exit 1; }
    python - "$out/${cfg%.json}_${b}_$f.json" $cfg $b $var $f <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print(sys.argv[2], 'batch', sys.argv[3], sys.argv[4] + '=' + sys.argv[5], 'ms', d['ms_per_step'], 'loss', d['loss'])
PY
  done
done
