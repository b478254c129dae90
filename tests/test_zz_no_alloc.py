"""The entry point that allocates nothing (dvae_recon_metrics_noalloc_f32) against the entry point it is a
variant of (dvae_recon_metrics_f32, caller-provided workspace), through the C ABI: same kernels, so the
three values must agree bit for bit -- on two calls in a row, the second with a smaller input (the
library's per-GPU scratch array keeps the first call's partial sums)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _clips(B, T, S, seed):
    g = torch.Generator().manual_seed(seed)
    target = (torch.rand(B, T, 1, S, S, generator=g) > 0.9).float() * torch.rand(B, T, 1, S, S, generator=g)
    recon = torch.sigmoid(torch.randn(B, T, 1, S, S, generator=g))
    return recon.cuda().contiguous(), target.cuda().contiguous()


def _with_workspace(lib, recon, target, max_val):
    B, T = recon.shape[:2]
    P = recon[0, 0].numel()
    ws = torch.empty(lib.dvae_recon_loss_ws_floats(B, T, P), dtype=torch.float32, device='cuda')
    out = torch.empty(3, dtype=torch.float32, device='cuda')
    rc = lib.dvae_recon_metrics_f32(recon.data_ptr(), target.data_ptr(), out.data_ptr(), ws.data_ptr(), B, T, P,
                                    max_val, torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.dvae_last_error()
    return out


def _no_alloc(lib, recon, target, max_val, out):
    B, T = recon.shape[:2]
    P = recon[0, 0].numel()
    rc = lib.dvae_recon_metrics_noalloc_f32(recon.data_ptr(), target.data_ptr(), out.data_ptr(), B, T, P, max_val,
                                            torch.cuda.current_stream().cuda_stream)
    assert rc == 0, lib.dvae_last_error()
    return out


def test_no_alloc_entry_point_matches_the_workspace_one_on_two_calls_in_a_row():
    from hipvae._lib import lib
    assert lib.dvae_recon_metrics_noalloc_max_elements() == 1 << 31
    big = _clips(300, 20, 64, 1)          # 2048 workgroups: every slot of the scratch array is written
    small = _clips(3, 5, 64, 2)           # 12 workgroups: the rest of the array still holds the first call's sums
    single = _clips(2, 1, 32, 3)          # one frame per clip: temporal consistency is NaN in both
    outs = [torch.full((3,), -1.0, device='cuda') for _ in range(3)]      # allocated before the calls under test
    refs = [_with_workspace(lib, *c, 1.0) for c in (big, small, single)]
    torch.cuda.synchronize()
    before = torch.cuda.memory_stats()['allocation.all.allocated']
    for c, o in zip((big, small, single), outs):
        _no_alloc(lib, *c, 1.0, o)
    assert torch.cuda.memory_stats()['allocation.all.allocated'] == before      # torch's allocator was not used
    torch.cuda.synchronize()
    for name, o, r in zip(('big', 'small', 'single'), outs, refs):
        got, ref = o.cpu(), r.cpu()
        assert torch.equal(got[:2], ref[:2]), (name, got, ref)
        assert torch.equal(got[2:], ref[2:]) or (torch.isnan(got[2]) and torch.isnan(ref[2])), (name, got, ref)
    # the Python wrapper goes through the same entry point
    from hipvae import ops
    o = ops.recon_metrics_noalloc(*small, torch.empty(3, device='cuda'))
    assert torch.equal(o.cpu(), refs[1].cpu())


def test_no_alloc_entry_point_under_graph_capture():
    """What the entry point is for: recorded into a HIP graph (after the explicit per-device set-up call, which is the
    only part that may not run under capture) and replayed on two different inputs, against the workspace variant."""
    from hipvae._lib import lib
    assert lib.dvae_recon_metrics_noalloc_init() == 0, lib.dvae_last_error()
    a, b = _clips(5, 4, 64, 11), _clips(5, 4, 64, 12)
    recon, target = torch.empty_like(a[0]), torch.empty_like(a[1])
    out = torch.full((3,), -1.0, device='cuda')
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph()
    from hipvae.graph import quiet_collector      # (a dead graph of an earlier test must not be collected in here)
    with quiet_collector(), torch.cuda.graph(g, stream=side):
        _no_alloc(lib, recon, target, 1.0, out)
    for clips in (a, b):
        recon.copy_(clips[0])
        target.copy_(clips[1])
        g.replay()
        ref = _with_workspace(lib, *clips, 1.0)
        torch.cuda.synchronize()
        assert torch.equal(out.cpu(), ref.cpu())


def test_no_alloc_entry_point_rejects_what_it_cannot_do():
    from hipvae._lib import lib
    recon, target = _clips(2, 3, 32, 4)
    out = torch.empty(3, device='cuda')
    st = torch.cuda.current_stream().cuda_stream
    # beyond the stated maximum: refused before anything is launched
    rc = lib.dvae_recon_metrics_noalloc_f32(recon.data_ptr(), target.data_ptr(), out.data_ptr(), 1 << 20, 1 << 8,
                                            1 << 12, 1.0, st)
    assert rc != 0 and b'exceeds the limit' in lib.dvae_last_error()
    rc = lib.dvae_recon_metrics_noalloc_f32(recon.data_ptr(), target.data_ptr(), out.data_ptr(), 2, 3, 1022, 1.0, st)
    assert rc != 0 and b'P%4==0' in lib.dvae_last_error()
    rc = lib.dvae_recon_metrics_noalloc_f32(recon.data_ptr(), target.data_ptr(), None, 2, 3, 1024, 1.0, st)
    assert rc != 0 and b'null argument' in lib.dvae_last_error()
