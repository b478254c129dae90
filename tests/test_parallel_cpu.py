"""Data-parallel gradient synchronisation (hipvae.parallel.GradSync) exercised with two gloo ranks on
CPU: bucket filling from gradient hooks, averaging, parameters without gradients, repeated steps, and
parameter broadcast from rank 0."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


class Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a = torch.nn.Linear(8, 16)
        self.b = torch.nn.Linear(16, 4)
        self.unused = torch.nn.Parameter(torch.ones(5))

    def forward(self, x):
        return self.b(torch.tanh(self.a(x)))


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    'disentangled-vae_amd'))
    from hipvae.parallel import GradSync
    torch.manual_seed(100 + rank)               # different init per rank: the broadcast must fix it
    net = Net()
    sync = GradSync(net, bucket_mb=0.0005)      # ~130 floats per bucket: several buckets
    assert len(sync.buckets) >= 3
    states = []
    for step in range(3):
        g = torch.Generator().manual_seed(10 * step + rank)
        x = torch.randn(6, 8, generator=g)
        sync.zero_grad()
        net(x).pow(2).sum().backward()
        sync.finish()
        states.append({k: p.grad.clone() for k, p in net.named_parameters()})
        with torch.no_grad():
            for p in net.parameters():
                p -= 0.01 * p.grad
    torch.save({'grads': states, 'params': {k: v.clone() for k, v in net.state_dict().items()}},
               os.path.join(out_dir, f'rank{rank}.pt'))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gradsync_two_ranks_gloo(tmp_path):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / 'rank0.pt')
    r1 = torch.load(tmp_path / 'rank1.pt')
    # every rank ends with identical gradients and parameters
    for s0, s1 in zip(r0['grads'], r1['grads']):
        for k in s0:
            assert torch.equal(s0[k], s1[k]), k
    for k in r0['params']:
        assert torch.equal(r0['params'][k], r1['params'][k]), k
    # and they equal the average of the per-rank gradients of a single-process replay
    torch.manual_seed(100)                      # rank 0's initialisation is what was broadcast
    net = Net()
    for step in range(3):
        grads = []
        for rank in range(world):
            g = torch.Generator().manual_seed(10 * step + rank)
            x = torch.randn(6, 8, generator=g)
            net.zero_grad()
            net(x).pow(2).sum().backward()
            grads.append({k: (p.grad.clone() if p.grad is not None else torch.zeros_like(p))
                          for k, p in net.named_parameters()})
        avg = {k: (grads[0][k] + grads[1][k]) / 2 for k in grads[0]}
        for k in avg:
            torch.testing.assert_close(r0['grads'][step][k], avg[k], rtol=1e-6, atol=1e-7)
        assert float(r0['grads'][step]['unused'].abs().sum()) == 0.0
        with torch.no_grad():
            for k, p in net.named_parameters():
                p -= 0.01 * avg[k]


class CutNet(torch.nn.Module):
    """stem | encoder | decoder with the boundaries the models of this package call: level 1 between encoders and
    decoder, level 2 inside one encoder (below its large layer)"""

    def __init__(self):
        super().__init__()
        self.stem = torch.nn.Linear(8, 8)
        self.enc = torch.nn.Linear(8, 6)
        self.enc2 = torch.nn.Linear(8, 6)
        self.dec = torch.nn.Linear(6, 4)
        self.unused = torch.nn.Parameter(torch.ones(3))

    def forward(self, x):
        from hipvae.parallel import stage_boundary
        (h,) = stage_boundary(torch.tanh(self.stem(x)), level=2)
        z, mu = torch.tanh(self.enc(h)), self.enc2(x)
        z, mu = stage_boundary(z, mu)
        # the loss uses the latent both through the decoder and directly, like the KL / independence terms
        return self.dec(z).pow(2).sum() + 0.1 * mu.pow(2).sum() + 0.05 * (z * mu).sum()


def _staged_worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                    'disentangled-vae_amd'))
    from hipvae.parallel import StagedGrads
    torch.manual_seed(7)                        # same initialisation on both ranks (broadcast is tested above)
    net = CutNet()
    staged = StagedGrads(list(net.parameters()))
    events, states = [], []
    orig = staged._all_reduce
    staged._all_reduce = lambda buf, op: (events.append(('reduce', buf.numel())), orig(buf, op))
    for step in range(3):
        g = torch.Generator().manual_seed(10 * step + rank)
        x = torch.randn(5, 8, generator=g)
        staged.stage1(lambda: (None, net(x)))
        # after stage 1 the decoder's gradients are complete and nothing has reached the encoders
        assert net.dec.weight.grad is not None and net.enc.weight.grad is None and net.enc2.weight.grad is None
        assert staged.nstages == 3
        events.append(('stage1',))
        staged.reduce(0)
        staged.stage(2)
        assert net.enc.weight.grad is not None and net.enc2.weight.grad is not None and net.stem.weight.grad is None
        events.append(('stage2',))
        staged.reduce(1)
        staged.stage(3)
        assert net.stem.weight.grad is not None
        events.append(('stage3',))
        staged.reduce(2)
        staged.finish()
        assert all(p.grad.data_ptr() == staged.views[id(p)].data_ptr() for p in staged.params)
        states.append({k: p.grad.clone() for k, p in net.named_parameters()})
        with torch.no_grad():
            for p in net.parameters():
                p -= 0.01 * p.grad
    pad = lambda ps: sum((p.numel() + 3) // 4 * 4 for p in ps)       # noqa: E731  (views start on 16-byte boundaries)
    n_dec = pad(net.dec.parameters())
    n_enc = pad(list(net.enc.parameters()) + list(net.enc2.parameters()))
    n_all = pad(net.parameters())
    # the parameter that never receives a gradient travels (as zeros) with the last run
    assert events[:6] == [('stage1',), ('reduce', n_dec), ('stage2',), ('reduce', n_enc), ('stage3',),
                          ('reduce', n_all - n_dec - n_enc)], events[:6]
    # a weight used on both sides of a boundary would be averaged before its gradient is complete: refused
    class Tied(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.w = torch.nn.Linear(6, 6)

        def forward(self, x):
            from hipvae.parallel import stage_boundary
            (z,) = stage_boundary(torch.tanh(self.w(x)))
            return self.w(z).pow(2).sum()
    tied = Tied()
    st2 = StagedGrads(list(tied.parameters()))
    st2.stage1(lambda: (None, tied(torch.randn(3, 6))))
    try:
        st2.stage(2)
        raise AssertionError('a shared weight across the boundary was not refused')
    except RuntimeError as e:
        assert 'both sides of a stage boundary' in str(e)
    torch.save({'grads': states}, os.path.join(out_dir, f'rank{rank}.pt'))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_staged_backward_two_ranks_gloo(tmp_path):
    """hipvae.parallel.StagedGrads: backward cut at the model's stage boundaries (two levels: three stages), each
    run of the flat buffer reduced as soon as its stage has filled it; the averaged gradients equal those of one
    uncut backward pass per rank, averaged.  Also: a weight shared across a boundary is refused."""
    world = 2
    port = _free_port()
    mp.spawn(_staged_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0 = torch.load(tmp_path / 'rank0.pt')
    r1 = torch.load(tmp_path / 'rank1.pt')
    for s0, s1 in zip(r0['grads'], r1['grads']):
        for k in s0:
            assert torch.equal(s0[k], s1[k]), k
    torch.manual_seed(7)
    net = CutNet()
    for step in range(3):
        grads = []
        for rank in range(world):
            g = torch.Generator().manual_seed(10 * step + rank)
            x = torch.randn(5, 8, generator=g)
            net.zero_grad()
            net(x).backward()                   # no staged step is recording: the boundary is the identity
            grads.append({k: (p.grad.clone() if p.grad is not None else torch.zeros_like(p))
                          for k, p in net.named_parameters()})
        avg = {k: (grads[0][k] + grads[1][k]) / 2 for k in grads[0]}
        for k in avg:
            torch.testing.assert_close(r0['grads'][step][k], avg[k], rtol=1e-6, atol=1e-7)
        assert float(r0['grads'][step]['unused'].abs().sum()) == 0.0
        with torch.no_grad():
            for k, p in net.named_parameters():
                p -= 0.01 * avg[k]
