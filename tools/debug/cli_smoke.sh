set -u
cd $GRAFT_REPO_ROOT
run() { name=$1; shift; timeout -k 10 200 python bench.py "$@" --steps 5 --warmup 2 --no-cpu-baseline --no-extras > gpurun_out/cli_$name.json 2> gpurun_out/cli_$name.err; echo "$name rc=$? $(python -c "import json,sys; d=json.load(open('gpurun_out/cli_$name.json')); print(d['ms_per_step'], d['value'], d['dtype'])" 2>&1 | tail -1)"; }
run infer --mode infer
run bf16 --precision bf16
run split --precision fp32_split
run eager --graph 0
run lstm32 --config lstm_simple.json --batch 32
run base256 --config small_baseline.json --batch 256
run hostpre --host-batches prefetch
run wide --config disentangled_128x128x40.json --batch 8
run b7 --batch 7
