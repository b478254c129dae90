"""Where the one-rank data-parallel graph step loses time against the single-graph step (same box, same process).

usage: python tools/debug/dp_overhead_probe.py [--steps 40]
Variants: the single graph; the staged step as it ships; the same with the collectives skipped (what the split into
graphs and the copies into the flat buffer cost); the same with the level-2 boundary ignored (two stages)."""
import argparse
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'disentangled-vae_amd'))
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--batch', type=int, default=256)
    ap.add_argument('--only', default='', help='run only the variants whose name contains this')
    a = ap.parse_args()
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(0)
    os.dup2(2, 1)
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29541')
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
    import models
    from hipvae import losses as L, parallel
    from hipvae.graph import GraphedDataParallelStep, GraphedTrainStep
    from hipvae.optim import FusedAdam
    cfg = bench.load_config('disentangled_simple.json')
    kind, t = cfg['model']['model_type'], cfg['training']
    T, size = cfg['data']['sequence_length'], cfg['data']['frame_size']
    ring = [bench.synthetic_clips(a.batch, T, size, i).to(dev) for i in range(4)]
    loss_of = lambda m, v: bench.loss_call(kind, L, m(v), v, t, 4.0)    # noqa: E731

    def fresh():
        torch.manual_seed(0)
        model = models.get_model(kind, **bench.model_kwargs(cfg)).to(dev).train()
        opt = FusedAdam(model.parameters(), lr=t['learning_rate'], betas=(t['beta1'], t['beta2']),
                        weight_decay=t['weight_decay'])
        return model, opt

    def timeit(step):
        for i in range(8):
            step(ring[i % 4])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(a.steps):
            step(ring[i % 4])
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / a.steps * 1e3

    rows = []

    def variant(name, cls):
        if a.only and a.only not in name:
            return
        model, opt = fresh()
        rows.append((name, timeit(cls(model, opt, loss_of, ring[0], gradient_clip=t['gradient_clip']))))

    variant('single graph', GraphedTrainStep)
    variant('staged, as shipped', GraphedDataParallelStep)
    plain = parallel.StagedGrads._all_reduce
    parallel.StagedGrads._all_reduce = lambda self, buf, op: None
    variant('staged, collectives skipped', GraphedDataParallelStep)
    boundary = parallel.stage_boundary
    parallel.stage_boundary = lambda *ts, level=1: boundary(*ts, level=level) if level == 1 else ts
    variant('two stages, collectives skipped', GraphedDataParallelStep)
    import models.disentangled_vae as mdv
    parallel.stage_boundary = mdv.stage_boundary = lambda *ts, level=1: ts
    variant('one stage, collectives skipped', GraphedDataParallelStep)
    parallel.StagedGrads._all_reduce = plain
    variant('one stage, one all-reduce', GraphedDataParallelStep)
    parallel.stage_boundary = mdv.stage_boundary = boundary
    for name, ms in rows:
        print(f'{name:36s} {ms:7.3f} ms', file=sys.stderr)
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
