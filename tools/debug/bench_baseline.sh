#!/bin/bash
# BaselineVAE at batch 32 and 256: training, inference and bf16 step times from bench.py, one line each.
# usage: tools/debug/bench_baseline.sh <tag>
set -u
out=gpurun_out/${1:-base}
mkdir -p "$out"
for b in 32 256; do
  timeout -k 10 200 python bench.py --config small_baseline.json --batch $b --no-cpu-baseline --no-kernel-table \
    > "$out/base_$b.json" 2> "$out/base_$b.err" || exit 1
  python - "$out/base_$b.json" $b <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print('batch', sys.argv[2], 'train ms', d['ms_per_step'], 'clips/s', d['value'], 'infer ms',
      d.get('inference', {}).get('ms_per_step'), 'bf16 ms', d.get('bf16_variant', {}).get('ms_per_step'))
PY
done
