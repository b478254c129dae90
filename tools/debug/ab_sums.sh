for rep in 1 2; do
for v in 1 0; do
for b in 256 32; do
DVAE_ONE_LAUNCH_SUMS=$v timeout -k 10 200 python bench.py --batch $b --no-cpu-baseline --no-extras --no-kernel-table --steps 40 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('one_launch=$v', d['config']['global_batch'], d['ms_per_step'], d['loss'])"
done; done; done
