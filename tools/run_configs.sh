set -u
cd $GRAFT_REPO_ROOT
for spec in "disentangled_simple.json 32" "lstm_simple.json 32" "lstm_simple.json 256" "small_baseline.json 32" "small_baseline.json 256" "disentangled_128x128x40.json 32"; do
  set -- $spec
  timeout -k 10 300 python bench.py --config $1 --batch $2 --no-cpu-baseline --no-kernel-table > gpurun_out/cfg_$1_$2.json 2> gpurun_out/cfg_$1_$2.err || exit 1
  echo "$1 $2 done"
done
