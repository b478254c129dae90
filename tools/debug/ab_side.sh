#!/bin/bash
# A/B of DVAE_SIDE_ENCODER on one box: disentangled step at batch 32 and 256, then the model tests with it on.
set -u
out=gpurun_out/${1:-side}
mkdir -p "$out"
for b in 32 256; do
  for f in 0 1; do
    DVAE_SIDE_ENCODER=$f timeout -k 10 200 python bench.py --batch $b --no-cpu-baseline --no-kernel-table --no-extras \
      > "$out/b${b}_$f.json" 2> "$out/b${b}_$f.err" || { tail -5 "$out/b${b}_$f.err"; exit 1; }
    python - "$out/b${b}_$f.json" $b $f <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
print('batch', sys.argv[2], 'side', sys.argv[3], 'ms', d['ms_per_step'], 'loss', d['loss'])
PY
  done
done
DVAE_SIDE_ENCODER=1 timeout -k 10 400 python -u -m pytest tests/test_models_gpu.py tests/test_trainer_gpu.py -m gpu -x -q \
  --timeout 120 --timeout-method thread > "$out/tests.log" 2>&1
tail -3 "$out/tests.log"
