set -u
cd $GRAFT_REPO_ROOT
run() { # name, env value
  DVAE_BENCH_FAIL_CAPTURE=$2 timeout -k 10 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --force-dist --no-cpu-baseline --no-extras --no-kernel-table > gpurun_out/dp_$1.json 2> gpurun_out/dp_$1.err
  echo "$1 rc=$?"
}
run ok ""
run capture capture:0
run warmup warmup:0
run pre 1
