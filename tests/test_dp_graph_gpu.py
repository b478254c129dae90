"""Two data-parallel ranks on ONE GPU: the captured step (hipvae.graph.GraphedDataParallelStep: decoder-side
backward, reduction of its gradients beside the backward of the encoders' upper parts, reduction of those beside
the backward of the convolution stacks, reduction of the rest, update)
against the plain statement of its semantics (broadcast from rank 0; every step: backward, all-reduce and
average every gradient, clip, Adam), from the same initialisation and on the same per-rank batches.
The collectives themselves run on host copies (gloo; RCCL refuses two ranks on one device): what is
checked is the sequence of graphs and reductions with more than one rank, not a backend.  The gradients of both ranks really are averaged between the two
graphs, both ranks take the same step, and that step is the plain one's."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

CFG = dict(content_dim=8, motion_dim=8, content_channels=[4, 8, 16, 32], motion_channels=[4, 8],
           decoder_channels=[32, 16, 8, 4], content_feature_dim=32, motion_feature_dim=32, lstm_hidden=16,
           lstm_layers=1, bidirectional=True, sequence_length=6)


def _clips(B, T, seed):
    g = torch.Generator().manual_seed(seed)
    u1 = torch.rand(B, T, 1, 64, 64, generator=g)
    u2 = torch.rand(B, T, 1, 64, 64, generator=g)
    return (u1 > 0.9).float() * u2


def _via_host(t, collective):
    """run a gloo collective on a host copy of a device tensor and return the result on the device (gloo's
    own device-tensor path stalls with two ranks on one GPU; DP_TEST_DEVICE_COLLECTIVES=1 selects it, for
    diagnosis: see the __main__ block)"""
    if os.environ.get('DP_TEST_DEVICE_COLLECTIVES') == '1':
        collective(t)
        return t
    h = t.detach().cpu()
    collective(h)
    return h.to(t.device)


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'disentangled-vae_amd'))
    torch.cuda.set_device(0)
    if os.environ.get('DP_TEST_DEVICE_COLLECTIVES') == '1':
        import faulthandler                                 # where is every thread when nothing moves?
        faulthandler.dump_traceback_later(45, exit=True, file=sys.stderr)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import models
    from hipvae import losses, rng
    from hipvae.graph import GraphedDataParallelStep
    from hipvae.optim import FusedAdam
    rng.draw_like = lambda t: torch.zeros_like(t)          # the same (zero) draws in both runs
    batches = [_clips(4, 6, 300 + 10 * step + rank).cuda() for step in range(3)]
    loss_of = lambda m, v: losses.disentangled_loss(m(v), v, beta=0.5, recon_loss_type='bce', independence_weight=0.1,
                                                    temporal_weight=0.05)
    out = {}
    trace = lambda *a: print(f'[rank {rank}]', *a, file=sys.stderr, flush=True) if os.environ.get('DP_TEST_TRACE') else None
    for mode in ('eager', 'graph'):
        trace('mode', mode)
        torch.manual_seed(40 + rank)                        # different per rank: the broadcast must fix it
        m = models.DisentangledVAE(**CFG).cuda().train()
        opt = FusedAdam(m.parameters(), lr=1e-3, weight_decay=1e-5)
        if mode == 'eager':
            # the plain statement of the semantics: broadcast, then per step average every gradient
            # (one collective per step: gloo on device tensors is slow per call)
            ps = list(m.parameters())
            flat = torch.cat([t.data.reshape(-1) for t in ps + [b for b in m.buffers() if b.is_floating_point()]])
            flat = _via_host(flat, lambda h: dist.broadcast(h, src=0))
            off = 0
            for t in ps + [b for b in m.buffers() if b.is_floating_point()]:
                t.data.copy_(flat[off:off + t.numel()].view_as(t))
                off += t.numel()
            trace('broadcast done')
            for v in batches:
                opt.zero_grad(set_to_none=True)
                loss_of(m, v)['total_loss'].backward()
                trace('backward done')
                g = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in ps])
                g = _via_host(g, dist.all_reduce) / world
                off = 0
                for p in ps:
                    p.grad = g[off:off + p.numel()].view_as(p).clone()
                    off += p.numel()
                opt.clip_grad_norm_(1.0, fold_into_step=True)
                opt.step()
        else:
            from hipvae.parallel import StagedGrads

            class HostStaged(StagedGrads):
                def _all_reduce(self, buf, op):
                    buf.copy_(_via_host(buf, dist.all_reduce))

            class HostReducedStep(GraphedDataParallelStep):
                def _make_staged(self, params, process_group):
                    return HostStaged(params, process_group)

            for t in list(m.parameters()) + list(m.buffers()):       # (the class would broadcast on the device)
                if t.is_floating_point():
                    t.data.copy_(_via_host(t.data, lambda h: dist.broadcast(h, src=0)))
            step = HostReducedStep(m, opt, loss_of, batches[0], gradient_clip=1.0, broadcast_parameters=False)
            trace('captured')
            # three runs, three backward graphs: decoder | encoders above the motion encoder's convolution stack (the
            # content encoder, one frame per clip, is not cut: its whole backward runs beside the recurrence) | that stack
            st = step.staged
            assert st.nstages == 3 and len(st.runs) == 3 and len(step.stage_graphs) == 2
            assert 0 < st.runs[0][1] == st.runs[1][0] < st.runs[1][1] == st.runs[2][0] < st.runs[2][1] == st.flat.numel()
            run_of = {n: st._run_of[id(p)] for n, p in m.named_parameters()}
            assert all((r == 0) == n.startswith('decoder.') for n, r in run_of.items())
            assert all((r == 2) == ('motion_encoder.cnn_encoder.conv.' in n) for n, r in run_of.items()), run_of
            assert run_of['content_encoder.cnn_encoder.conv.0.weight'] == 1
            assert run_of['motion_encoder.cnn_encoder.fc.0.weight'] == 1
            for v in batches:
                step(v)
        torch.cuda.synchronize()
        trace('mode done')
        out[mode] = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    torch.save(out, os.path.join(out_dir, f'rank{rank}.pt'))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_graphed_step_matches_plain_averaging(tmp_path):
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / 'rank0.pt'), torch.load(tmp_path / 'rank1.pt')
    for k in r0['graph']:
        if not r0['graph'][k].is_floating_point():
            continue
        if 'running_' not in k:                  # BatchNorm statistics are per rank by design
            assert torch.equal(r0['graph'][k], r1['graph'][k]), k
            assert torch.equal(r0['eager'][k], r1['eager'][k]), k
        a, b = r0['eager'][k], r0['graph'][k]
        # the staged backward sums the contributions to a latent's gradient in another order than one
        # backward pass does: last-bit differences, nothing at the level a missed reduction would show
        d = (a - b).abs().max().item()
        assert d <= 1e-6 + 1e-5 * a.abs().max().item(), (k, d)
    # and the ranks did influence each other: rank 1's batches moved rank 0's parameters
    moved = max((r0['graph'][k] - r1['graph'][k]).abs().max().item() for k in r0['graph'] if 'running_mean' in k)
    assert moved > 0


if __name__ == '__main__':
    # diagnosis of the stall of gloo's device-tensor collectives with two ranks on one GPU:
    #   DP_TEST_DEVICE_COLLECTIVES=1 DP_TEST_TRACE=1 python tests/test_dp_graph_gpu.py
    # every collective then runs on the device tensor itself; after 45 s without finishing, each rank
    # prints the Python stack of all its threads and exits
    import tempfile
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, tempfile.mkdtemp()), nprocs=2, join=True)
    print('finished without a stall')
