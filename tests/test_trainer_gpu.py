"""The Trainer surface on the GPU against the reference's own run (tests/golden/trainer_run.*): same
config, same synthetic batches, same injected normal draws, three epochs; then the data-parallel
semantics (per-replica BatchNorm statistics, averaged gradients) against the CPU oracle."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import vae_oracle as O

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')


def synthetic_clips(B, T, seed):
    g = torch.Generator().manual_seed(seed)
    u1 = torch.rand(B, T, 1, 64, 64, generator=g)
    u2 = torch.rand(B, T, 1, 64, 64, generator=g)
    return (u1 > 0.9).float() * u2


def test_trainer_run_matches_reference(tmp_path):
    from hipvae import rng
    from training.config import DataConfig, ExperimentConfig, ModelConfig, TrainingConfig
    from training.trainer import create_trainer
    info = json.load(open(os.path.join(GOLDEN, 'trainer_run.json')))
    arrays = np.load(os.path.join(GOLDEN, 'trainer_run.npz'))
    raw = info['config']
    cfg = ExperimentConfig(name=raw['name'])
    cfg.data, cfg.model, cfg.training = DataConfig(**raw['data']), ModelConfig(**raw['model']), \
        TrainingConfig(**raw['training'])
    cfg.training.device = 'cuda'
    cfg.training.checkpoint_dir = str(tmp_path / 'ck')
    cfg.training.results_dir = str(tmp_path / 'res')
    train = [(synthetic_clips(4, 6, 20 + i), [{}] * 4) for i in range(3)]
    val = [(synthetic_clips(4, 6, 40 + i), [{}] * 4) for i in range(2)]
    torch.manual_seed(5)
    tr = create_trainer(cfg, train, val)
    for k, v in tr.model.state_dict().items():
        assert np.array_equal(v.cpu().numpy(), arrays['sd0/' + k]), k       # identical initialisation
    eps = [torch.from_numpy(arrays[f'eps{k}']) for k in range(9)]
    with rng.inject_eps(eps):
        hist = tr.train()
    # default flags: every training step and every validation batch was a hipGraph replay (the injected draws
    # travel through the captured step's draw buffers), one capture for all three values of the annealed beta
    assert tr.use_graph and tr.graph_fallback_reason is None
    assert len(tr._graphs) == 1 and len(tr._val_graphs) == 1 and all(tr._graphs.values())
    assert next(iter(tr._graphs.values())).eps_slots is not None
    ref = info['history']
    assert sorted(hist) == sorted(ref)
    for k in ref:
        np.testing.assert_allclose(hist[k], ref[k], rtol=2e-4, err_msg=k)
    ck_dir = os.path.join(cfg.training.checkpoint_dir, cfg.name)
    assert sorted(os.listdir(ck_dir)) == info['files']
    assert sorted(os.listdir(os.path.join(cfg.training.results_dir, cfg.name))) == info['results_files']
    ck = torch.load(os.path.join(ck_dir, 'final_model.pt'), weights_only=False)
    assert sorted(ck.keys()) == info['checkpoint_keys']
    assert ck['epoch'] == info['epoch'] and ck['global_step'] == info['global_step']
    assert ck['config'] == {**raw, 'training': {**raw['training'], 'device': 'cuda',
                                                 'checkpoint_dir': cfg.training.checkpoint_dir,
                                                 'results_dir': cfg.training.results_dir}}
    osd = ck['optimizer_state_dict']
    assert sorted(osd['param_groups'][0].keys()) == info['optimizer_param_group_keys']
    assert sorted(osd['state'][0].keys()) == info['optimizer_state_keys']
    assert str(osd['state'][0]['step'].dtype) == info['optimizer_step_dtype']
    assert float(osd['state'][0]['step']) == info['optimizer_step_value']
    for k, v in ck['model_state_dict'].items():
        ref_v = arrays['final/' + k]
        if ref_v.dtype.kind == 'f':
            np.testing.assert_allclose(v.cpu().numpy(), ref_v, rtol=2e-3, atol=2e-5, err_msg=k)
        else:
            assert np.array_equal(v.cpu().numpy(), ref_v), k
    # a checkpoint written by this build loads into a freshly built trainer (strict state_dict)
    tr2 = create_trainer(cfg, train, val)
    assert tr2.load_checkpoint('final_model.pt') is True
    assert tr2.current_epoch == info['epoch'] and tr2.load_checkpoint('checkpoint.pt') is False


def test_two_replica_semantics_match_oracle():
    """N data-parallel ranks = N independent micro-batches (own BatchNorm statistics), gradients averaged,
    one clip + Adam step.  Emulated on one GPU by running the two shards one after the other."""
    import models
    from hipvae import losses, rng
    from hipvae.optim import FusedAdam
    cfg = dict(content_dim=16, motion_dim=16, content_channels=[4, 8, 16, 32], motion_channels=[4, 8],
               decoder_channels=[32, 16, 8, 4], content_feature_dim=32, motion_feature_dim=32, lstm_hidden=32,
               lstm_layers=1, bidirectional=True, sequence_length=6)
    tcfg = dict(beta_content=0.4, beta_motion=0.4, recon_loss_type='bce', temporal_weight=0.05,
                independence_weight=0.02, gradient_clip=5.0, weight_decay=1e-5, beta1=0.9, beta2=0.999)
    torch.manual_seed(3)
    m = models.DisentangledVAE(**cfg)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    video = synthetic_clips(8, 6, 9)
    eps = [torch.randn(8, 16, generator=torch.Generator().manual_seed(100 + k)) for k in range(2)]
    sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
    new_sd, _, _, ograds = O.train_step('disentangled', sd64, {}, video.double(), [e.double() for e in eps], 0.4, tcfg,
                                        3e-4, micro_batches=2)
    m = m.cuda().train()
    opt = FusedAdam(m.parameters(), lr=3e-4, weight_decay=1e-5)
    acc = None
    bn0 = None
    for r in range(2):
        sl = slice(4 * r, 4 * r + 4)
        if r == 1:   # rank 1 starts from the same buffers as rank 0 did
            bn_after_rank0 = {k: v.clone() for k, v in m.state_dict().items() if 'running' in k or 'num_b' in k}
            m.load_state_dict({**m.state_dict(), **{k: sd[k].cuda() for k in bn_after_rank0}})
        opt.zero_grad()
        with rng.inject_eps([e[sl] for e in eps]):
            out = m(video[sl].cuda())
        losses.disentangled_loss(out, video[sl].cuda(), beta=0.4, beta_content=0.4, beta_motion=0.4,
                                 recon_loss_type='bce', temporal_weight=0.05,
                                 independence_weight=0.02)['total_loss'].backward()
        g = [p.grad.clone() for p in m.parameters()]
        acc = g if acc is None else [a + b for a, b in zip(acc, g)]
    m.load_state_dict({**m.state_dict(), **bn_after_rank0})          # rank 0's buffers are the ones kept
    for p, a in zip(m.parameters(), acc):
        p.grad = a / 2
    opt.clip_grad_norm_(5.0, fold_into_step=True)
    opt.step()
    # the oracle's gradients are post-clip: the post-step parameters are what is compared
    for k, v in m.state_dict().items():
        ref = new_sd[k]
        if ref.is_floating_point():
            np.testing.assert_allclose(v.cpu().numpy(), ref.float().numpy(), rtol=2e-4, atol=5e-6, err_msg=k)


def test_graphed_step_equals_eager_steps():
    """Three replays of the captured step give the parameters of three eager steps (same batches, same
    injected-free path: eval of eps is disabled by using beta only through mu, so compare with eps = 0)."""
    import models
    from hipvae import losses, rng
    from hipvae.graph import GraphedTrainStep
    from hipvae.optim import FusedAdam
    cfg = dict(latent_dim=16, encoder_channels=[4, 8, 16, 32], decoder_channels=[32, 16, 8, 4], feature_dim=32,
               lstm_hidden=32, lstm_layers=2, bidirectional=True, sequence_length=6)
    batches = [synthetic_clips(4, 6, 70 + i).cuda() for i in range(3)]

    def loss_of(m, v):
        recon, z, mu, logvar = m(v)
        return losses.lstm_loss(recon, v, mu, logvar, beta=0.5, recon_loss_type='bce', free_bits=0.25,
                                use_free_bits=True)

    zero_eps = lambda t: torch.zeros_like(t)
    results = []
    orig = rng.draw_like
    rng.draw_like = zero_eps          # identical (zero) draws in both runs
    import models.encoders as enc
    try:
        for mode in ('eager', 'graph'):
            torch.manual_seed(11)
            m = models.LSTMVAE(**cfg).cuda().train()
            opt = FusedAdam(m.parameters(), lr=1e-3, weight_decay=1e-5)
            sched = torch.optim.lr_scheduler.ExponentialLR(opt, gamma=0.5)
            if mode == 'eager':
                for v in batches:
                    opt.zero_grad()
                    loss_of(m, v)['total_loss'].backward()
                    opt.clip_grad_norm_(5.0, fold_into_step=True)
                    opt.step()
                    sched.step()
            else:
                step = GraphedTrainStep(m, opt, loss_of, batches[0], gradient_clip=5.0)
                for v in batches:
                    out = step(v)
                    sched.step()
                assert torch.isfinite(out['total_loss'])
            sd = opt.state_dict()
            results.append(({k: v.clone() for k, v in m.state_dict().items()},
                            float(sd['state'][0]['step'])))
    finally:
        rng.draw_like = orig
    (a, sa), (b, sb) = results
    assert sa == sb == 3.0
    for k in a:
        if a[k].is_floating_point():
            # every kernel has a fixed reduction order; the one difference is Adam's bias correction,
            # computed on the device in fp32 under replay and on the host in fp64 otherwise.  A stale
            # learning rate or step count would show up at the 1e-3 (= lr) level
            d = (a[k] - b[k]).abs().max().item()
            assert d <= 1e-6 + 1e-5 * a[k].abs().max().item(), (k, d)
        else:
            assert torch.equal(a[k], b[k]), k


def test_trainer_with_graph_replay_matches_eager(tmp_path):
    """Trainer(use_graph=True): three epochs with KL annealing (beta changes per epoch and reaches the one
    captured step through device memory) give the history of the eager Trainer.  Normal draws are zero in both runs
    (a captured graph draws from torch's generator, an eager run from the same generator at different
    offsets, so equal noise needs a fixed value)."""
    from hipvae import rng
    from training.config import DataConfig, ExperimentConfig, ModelConfig, TrainingConfig
    from training.trainer import create_trainer
    info = json.load(open(os.path.join(GOLDEN, 'trainer_run.json')))
    raw = info['config']
    train = [(synthetic_clips(4, 6, 20 + i), [{}] * 4) for i in range(3)]
    val = [(synthetic_clips(4, 6, 40 + i), [{}] * 4) for i in range(2)]
    orig = rng.draw_like
    rng.draw_like = lambda t: torch.zeros_like(t)
    hists = []
    try:
        for use_graph in (False, True):
            cfg = ExperimentConfig(name=raw['name'])
            cfg.data, cfg.model, cfg.training = DataConfig(**raw['data']), ModelConfig(**raw['model']), \
                TrainingConfig(**raw['training'])
            cfg.training.device = 'cuda'
            cfg.training.checkpoint_dir = str(tmp_path / f'ck{int(use_graph)}')
            cfg.training.results_dir = str(tmp_path / f'res{int(use_graph)}')
            torch.manual_seed(5)
            tr = create_trainer(cfg, train, val)
            tr.use_graph = use_graph
            hists.append(tr.train())
            if use_graph:
                assert len(tr._graphs) == 1 and len(tr._val_graphs) == 1     # one capture serves every beta
                assert tr.graph_fallback_reason is None
            else:
                assert not tr._graphs and not tr._val_graphs
                sd = tr.optimizer.state_dict()
                assert float(sd['state'][0]['step']) == 9.0          # 3 epochs x 3 batches, warm-ups undone
    finally:
        rng.draw_like = orig
    eager, graph = hists
    assert sorted(eager) == sorted(graph)
    for k in eager:
        np.testing.assert_allclose(graph[k], eager[k], rtol=1e-6, err_msg=k)


@pytest.mark.parametrize('kind', ['disentangled', 'lstm'])
def test_graph_replay_of_the_full_width_models_matches_eager(kind):
    """The one-launch LSTM kernels (hidden size 256: encoder chains, single-layer and stacked decoder) only run at
    the models' real widths.  Four replays of the captured step against four eager steps from the same state with
    zero noise: same losses and parameters, and no workgroup of those kernels gave up a wait (under replay the
    zeroing of their arrival counters must be ordered in front of them)."""
    from hipvae import losses as L, ops, rng
    from hipvae.graph import GraphedTrainStep
    from hipvae.optim import FusedAdam
    from models import get_model
    orig = rng.draw_like
    rng.draw_like = lambda t: torch.zeros_like(t)
    try:
        results = []
        clips = [synthetic_clips(8, 20, 70 + i).cuda() for i in range(4)]
        for use_graph in (False, True):
            torch.manual_seed(11)
            if kind == 'disentangled':
                model = get_model('disentangled', sequence_length=20).cuda()
                loss_of = lambda m, v: L.disentangled_loss(m(v), v, beta=4.0, recon_loss_type='bce',
                                                           temporal_weight=0.05, independence_weight=0.02)
            else:
                model = get_model('lstm', sequence_length=20, lstm_layers=2, bidirectional=True).cuda()
                loss_of = lambda m, v: (lambda o: L.lstm_loss(o[0], v, o[2], o[3], beta=1.0, recon_loss_type='bce'))(m(v))
            opt = FusedAdam(model.parameters(), lr=1e-3)
            losses = []
            if use_graph:
                g = GraphedTrainStep(model, opt, loss_of, clips[0], gradient_clip=5.0)
                for v in clips:
                    losses.append(float(g(v)['total_loss']))
            else:
                for v in clips:
                    opt.zero_grad(set_to_none=True)
                    ld = loss_of(model, v)
                    ld['total_loss'].backward()
                    opt.clip_grad_norm_(5.0)
                    opt.step()
                    losses.append(float(ld['total_loss']))
            torch.cuda.synchronize()
            results.append((losses, [p.detach().clone() for p in model.parameters()]))
        assert ops.lstm_seq_timeouts() == 0
        (le, pe), (lg, pg) = results
        np.testing.assert_allclose(lg, le, rtol=2e-5)
        for a, b in zip(pe, pg):
            assert (a - b).abs().max().item() <= 1e-6 + 2e-4 * a.abs().max().item()
    finally:
        rng.draw_like = orig


def _tiny_trainer_config(tmp_path, tag, num_epochs):
    from training.config import DataConfig, ExperimentConfig, ModelConfig, TrainingConfig
    info = json.load(open(os.path.join(GOLDEN, 'trainer_run.json')))
    raw = info['config']
    cfg = ExperimentConfig(name=raw['name'])
    cfg.data, cfg.model, cfg.training = DataConfig(**raw['data']), ModelConfig(**raw['model']), \
        TrainingConfig(**raw['training'])
    cfg.training.device = 'cuda'
    cfg.training.checkpoint_dir = str(tmp_path / f'ck_{tag}')
    cfg.training.results_dir = str(tmp_path / f'res_{tag}')
    cfg.training.num_epochs = num_epochs
    return cfg


def test_graph_replay_with_ragged_last_batch_and_lr_schedule(tmp_path):
    """Constant beta, ExponentialLR, and an epoch whose last batch is smaller: the Trainer alternates between
    two captured graphs (one per batch shape) every epoch.  Both must keep advancing the one device-resident
    {step count, learning rate} pair and see every scheduler change; history and final parameters equal the
    eager Trainer's."""
    from hipvae import rng
    from training.trainer import create_trainer
    train = [(synthetic_clips(4, 6, 20), [{}] * 4), (synthetic_clips(4, 6, 21), [{}] * 4),
             (synthetic_clips(2, 6, 22), [{}] * 2)]
    val = [(synthetic_clips(4, 6, 40), [{}] * 4)]
    orig = rng.draw_like
    rng.draw_like = lambda t: torch.zeros_like(t)
    runs = []
    try:
        for use_graph in (False, True):
            cfg = _tiny_trainer_config(tmp_path, f'ragged{int(use_graph)}', 4)
            cfg.training.use_kl_annealing = False
            cfg.training.kl_start_epoch = 0
            cfg.training.lr_scheduler = 'exponential'
            cfg.training.lr_decay = 0.5
            cfg.training.lr_min = 1e-9
            cfg.training.learning_rate = 2e-3
            torch.manual_seed(5)
            tr = create_trainer(cfg, train, val)
            tr.use_graph = use_graph
            hist = tr.train()
            if use_graph:
                assert len(tr._graphs) == 2                         # one per batch shape, captured once each
                dyn = tr.optimizer._dyn
                assert all(g.opt._dyn is dyn for g in tr._graphs.values())
                assert float(dyn[0]) == 12.0                        # 4 epochs x 3 steps on the shared counter
                assert float(tr.optimizer.state_dict()['state'][0]['step']) == 12.0
            runs.append((hist, {k: v.detach().clone() for k, v in tr.model.state_dict().items()}))
    finally:
        rng.draw_like = orig
    (he, pe), (hg, pg) = runs
    assert he['learning_rate'][0] != he['learning_rate'][-1]
    # every kernel has a fixed reduction order and the replayed Adam step reads the same fp32 bias
    # corrections the eager one is handed, so the two runs agree bit for bit; a stale learning rate or
    # step count in one of the graphs would show at the 1e-3 level
    for k in he:
        assert hg[k] == he[k], (k, hg[k], he[k])
    for k in pe:
        assert torch.equal(pe[k], pg[k]), (k, (pe[k].float() - pg[k].float()).abs().max().item())


def test_load_checkpoint_between_graphed_steps(tmp_path):
    """A graphed step, load_checkpoint(), another graphed step: the second step must run on the loaded
    parameters and Adam state (graphs captured before the load are not replayed against stale storage)."""
    from hipvae import rng
    from training.trainer import create_trainer
    batch = synthetic_clips(4, 6, 33)
    train = [(batch, [{}] * 4)]
    orig = rng.draw_like
    rng.draw_like = lambda t: torch.zeros_like(t)
    try:
        finals = []
        for use_graph in (False, True):
            cfg = _tiny_trainer_config(tmp_path, f'reload{int(use_graph)}', 1)
            cfg.training.use_kl_annealing = False
            cfg.training.kl_start_epoch = 0
            torch.manual_seed(5)
            tr = create_trainer(cfg, train, train)
            tr.use_graph = use_graph
            tr.async_checkpoint = False
            tr.model.train()
            tr.train_step(batch.cuda(), 0.4)
            tr.train_step(batch.cuda(), 0.4)
            tr.save_checkpoint('mid.pt')                            # state after two steps
            tr.train_step(batch.cuda(), 0.4)
            tr.train_step(batch.cuda(), 0.4)
            assert tr.load_checkpoint('mid.pt') is True              # back to the state after two steps
            tr.model.train()
            tr.train_step(batch.cuda(), 0.4)                         # the third step, again
            sd = tr.optimizer.state_dict()
            assert float(sd['state'][0]['step']) == 3.0
            finals.append(({k: v.detach().clone() for k, v in tr.model.state_dict().items()},
                           [st['exp_avg'].detach().clone() for st in tr.optimizer.state.values()]))
    finally:
        rng.draw_like = orig
    (pe, me), (pg, mg) = finals
    for k in pe:
        assert torch.equal(pe[k], pg[k]), (k, (pe[k].float() - pg[k].float()).abs().max().item())
    for a, b in zip(me, mg):
        assert torch.equal(a, b)


def test_gradsync_on_rccl_single_rank_equals_plain_step():
    """The GPU side of GradSync (gradient hooks, side comm stream, RCCL AVG all-reduce, p.grad aliasing
    flat buckets, clip + fused Adam reading the aliases) on a one-rank RCCL group: averaging over one
    rank is the identity, so three steps must give exactly the parameters of three plain steps.  (The
    multi-rank arithmetic is covered by the two-rank gloo test; two ranks cannot share the one GPU.)"""
    import socket
    import torch.distributed as dist
    import models
    from hipvae import losses, rng
    from hipvae.optim import FusedAdam
    from hipvae.parallel import GradSync
    cfg = dict(content_dim=8, motion_dim=8, content_channels=[4, 8, 16, 32], motion_channels=[4, 8],
               decoder_channels=[32, 16, 8, 4], content_feature_dim=32, motion_feature_dim=32, lstm_hidden=16,
               lstm_layers=1, bidirectional=True, sequence_length=6)
    batches = [synthetic_clips(4, 6, 90 + i).cuda() for i in range(3)]

    def run(use_sync):
        torch.manual_seed(21)
        m = models.DisentangledVAE(**cfg).cuda().train()
        opt = FusedAdam(m.parameters(), lr=1e-3, weight_decay=1e-5)
        sync = GradSync(m, bucket_mb=0.02) if use_sync else None
        if sync is not None:
            assert len(sync.buckets) >= 3
        for v in batches:
            if sync is not None:
                sync.zero_grad()
            else:
                opt.zero_grad(set_to_none=True)
            out = m(v)
            losses.disentangled_loss(out, v, beta=0.5, recon_loss_type='bce', independence_weight=0.1,
                                     temporal_weight=0.05)['total_loss'].backward()
            if sync is not None:
                sync.finish()
            opt.clip_grad_norm_(1.0, fold_into_step=True)
            opt.step()
        if sync is not None:
            sync.remove()
        return {k: v.clone() for k, v in m.state_dict().items()}

    orig = rng.draw_like
    rng.draw_like = lambda t: torch.zeros_like(t)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    def run_graphed(parallel):
        """the captured step: one graph, or the data-parallel pair of graphs around one all-reduce"""
        from hipvae.graph import GraphedDataParallelStep, GraphedTrainStep
        torch.manual_seed(21)
        m = models.DisentangledVAE(**cfg).cuda().train()
        opt = FusedAdam(m.parameters(), lr=1e-3, weight_decay=1e-5)
        loss_of = lambda mm, v: losses.disentangled_loss(mm(v), v, beta=0.5, recon_loss_type='bce',
                                                         independence_weight=0.1, temporal_weight=0.05)
        step = (GraphedDataParallelStep if parallel else GraphedTrainStep)(m, opt, loss_of, batches[0], gradient_clip=1.0)
        if parallel:
            assert all(p.grad is not None and p.grad.data_ptr() == v.data_ptr() for p, v in zip(step.params, step.views))
        for v in batches:
            out = step(v)
        assert torch.isfinite(out['total_loss'])
        assert float(opt.state_dict()['state'][0]['step']) == 3.0
        return {k: v.clone() for k, v in m.state_dict().items()}

    try:
        plain = run(False)
        synced = run(True)
        graphed = run_graphed(False)
        graphed_dp = run_graphed(True)
    finally:
        dist.destroy_process_group()
        rng.draw_like = orig
    for k in plain:
        assert torch.equal(plain[k], synced[k]), (k, (plain[k].float() - synced[k].float()).abs().max().item())
        # one rank: the two-graph data-parallel step is the one-graph step, bit for bit
        assert torch.equal(graphed[k], graphed_dp[k]), (k, (graphed[k].float() - graphed_dp[k].float()).abs().max().item())


def test_dp_graph_step_in_bf16_at_batch_256_equals_the_one_graph_step():
    """The data-parallel bf16 configuration as one thing: the captured data-parallel step
    (three graphs around the RCCL all-reduces, here on a one-rank group, where averaging is the identity) with bf16
    MFMA operands, full-width DisentangledVAE, 256 clips.  Three steps give the parameters of the one-graph bf16
    step bit for bit (same kernels in the same order; only the graph boundaries and the flat gradient buffer
    differ)."""
    import socket
    import torch.distributed as dist
    import models
    from hipvae import losses, ops, rng
    from hipvae.graph import GraphedDataParallelStep, GraphedTrainStep
    from hipvae.optim import FusedAdam
    batches = [synthetic_clips(256, 20, 300 + i).cuda() for i in range(2)]
    loss_of = lambda mm, v: losses.disentangled_loss(mm(v), v, beta=4.0, beta_content=4.0, beta_motion=4.0,
                                                     recon_loss_type='bce', independence_weight=0.02,
                                                     temporal_weight=0.05)

    def run(parallel):
        torch.manual_seed(21)
        m = models.get_model('disentangled', sequence_length=20).cuda().train()
        opt = FusedAdam(m.parameters(), lr=1e-4, weight_decay=1e-5)
        step = (GraphedDataParallelStep if parallel else GraphedTrainStep)(m, opt, loss_of, batches[0], gradient_clip=5.0)
        out = None
        for i in range(3):
            out = step(batches[i % 2])
        assert torch.isfinite(out['total_loss'])
        return {k: v.clone() for k, v in m.state_dict().items()}, float(out['total_loss'])

    orig = rng.draw_like
    rng.draw_like = lambda t: torch.zeros_like(t)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    ops.set_precision('bf16')
    try:
        one, loss_one = run(False)
        dp, loss_dp = run(True)
        assert ops.lstm_seq_timeouts() == 0
    finally:
        ops.set_precision('fp32')
        dist.destroy_process_group()
        rng.draw_like = orig
    assert loss_one == loss_dp
    for k in one:
        assert torch.equal(one[k], dp[k]), (k, (one[k].float() - dp[k].float()).abs().max().item())


def test_trainer_says_why_a_step_is_not_replayed(tmp_path, capsys):
    """A loss function the Trainer does not know may hold beta as a launch constant: the step then runs as eager
    launches and the reason is printed once and kept."""
    from hipvae import losses
    from training.trainer import create_trainer
    cfg = _tiny_trainer_config(tmp_path, 'why', 1)
    train = [(synthetic_clips(4, 6, 20 + i), [{}] * 4) for i in range(2)]
    torch.manual_seed(5)
    tr = create_trainer(cfg, train, train[:1])
    assert tr.use_graph                                          # the default
    tr.loss_fn = lambda *a, **k: losses.baseline_loss(*a, **k)
    out = tr.train_epoch()
    assert np.isfinite(out['total_loss']) and not tr._graphs
    assert 'user-supplied loss' in tr.graph_fallback_reason
    assert capsys.readouterr().out.count('runs as eager launches') == 1


@pytest.mark.parametrize('flags', [[], ['--no-hip-graph', '--host-render'], ['--precision', 'bf16']])
def test_cli_end_to_end(tmp_path, flags):
    """`python -m training.train --config <json> --device cuda` (here: its main() in process) on a tiny
    configuration with digit images from a file: two epochs, checkpoints and history written, losses
    finite and decreasing.  Default flags = graph-replayed step and validation + GPU-rendered clips; then the
    host-driven forms (eager launches, clips rendered by the DataLoader), and bf16 mode."""
    import hipvae
    from tests.golden.make_golden import synthetic_digit_bank
    from training import train as train_cli
    from training.config import ExperimentConfig
    info = json.load(open(os.path.join(GOLDEN, 'trainer_run.json')))
    raw = json.loads(json.dumps(info['config']))
    raw['name'] = 'cli_run'
    raw['data'].update(train_size=24, val_size=8, test_size=8, batch_size=8, num_workers=0, sequence_length=6)
    raw['model']['sequence_length'] = 6
    raw['training'].update(num_epochs=2, checkpoint_dir=str(tmp_path / 'ck'), results_dir=str(tmp_path / 'res'),
                           save_every=1, learning_rate=1e-3)
    cfg_path = tmp_path / 'cfg.json'
    json.dump(raw, open(cfg_path, 'w'))
    ExperimentConfig.load(str(cfg_path))                         # the file is a valid reference-schema config
    bank_path = tmp_path / 'bank.npy'
    np.save(bank_path, (synthetic_digit_bank() * 255).astype(np.uint8))
    try:
        hist = train_cli.main(['--config', str(cfg_path), '--device', 'cuda', '--digit-bank', str(bank_path)] + flags)
    finally:
        hipvae.set_precision('fp32')
    assert len(hist['train_loss']) == 2 and all(np.isfinite(hist['train_loss'])) and all(np.isfinite(hist['val_loss']))
    assert hist['train_recon'][1] < hist['train_recon'][0]
    files = sorted(os.listdir(tmp_path / 'ck' / 'cli_run'))
    assert 'final_model.pt' in files and 'best_model.pt' in files and 'config.json' in files


def test_async_checkpoint_is_a_consistent_snapshot(tmp_path):
    """save_checkpoint() returns before the file is written; what lands on disk is the state at the call,
    even though parameters and optimiser state are overwritten right afterwards, and equals what the
    blocking writer produces."""
    from training.config import ExperimentConfig
    from training.trainer import create_trainer
    cfg = ExperimentConfig(name='ckpt')
    cfg.model.model_type = 'baseline'
    cfg.model.latent_dim, cfg.model.feature_dim = 16, 32
    cfg.model.encoder_channels, cfg.model.decoder_channels = [8, 16, 16, 32], [32, 16, 16, 8]
    cfg.model.sequence_length = cfg.data.sequence_length = 4
    cfg.training.device = 'cuda'
    cfg.training.checkpoint_dir, cfg.training.results_dir = str(tmp_path / 'ck'), str(tmp_path / 'res')
    clips = [(synthetic_clips(4, 4, 7), [{}] * 4)]
    torch.manual_seed(11)
    tr = create_trainer(cfg, clips, clips)
    tr.train_epoch()                                             # some optimiser state to save
    want = {k: v.detach().clone() for k, v in tr.model.state_dict().items()}
    want_opt = [tr.optimizer.state_dict()['state'][i]['exp_avg'].detach().clone() for i in range(3)]
    saved_p = [p.detach().clone() for p in tr.model.parameters()]
    saved_m = [st['exp_avg'].detach().clone() for st in tr.optimizer.state.values()]
    assert tr.async_checkpoint
    tr.save_checkpoint('a.pt')
    with torch.no_grad():                                        # training goes on: everything changes
        for p in tr.model.parameters():
            p.add_(1.0)
        for st in tr.optimizer.state.values():
            st['exp_avg'].mul_(3.0)
    tr.wait_for_checkpoint()
    ck_dir = os.path.join(cfg.training.checkpoint_dir, cfg.name)
    assert sorted(f for f in os.listdir(ck_dir) if f.endswith('.pt') or f.endswith('.tmp')) == ['a.pt']
    a = torch.load(os.path.join(ck_dir, 'a.pt'), weights_only=False)
    for k, v in want.items():
        assert torch.equal(a['model_state_dict'][k].cpu(), v.cpu()), k
    for i, v in enumerate(want_opt):
        assert torch.equal(a['optimizer_state_dict']['state'][i]['exp_avg'].cpu(), v.cpu())
    assert getattr(a['model_state_dict'], '_metadata', None) is not None
    # the blocking writer stores the same dictionary
    with torch.no_grad():
        for p, v in zip(tr.model.parameters(), saved_p):
            p.copy_(v)
        for st, v in zip(tr.optimizer.state.values(), saved_m):
            st['exp_avg'].copy_(v)
    tr.async_checkpoint = False
    tr.save_checkpoint('b.pt')
    b = torch.load(os.path.join(ck_dir, 'b.pt'), weights_only=False)
    assert sorted(a.keys()) == sorted(b.keys())
    for k in want:
        assert torch.equal(a['model_state_dict'][k].cpu(), b['model_state_dict'][k].cpu()), k
    assert a['history'] == b['history'] and a['config'] == b['config'] and a['epoch'] == b['epoch']
    # and a failed write is reported, not swallowed
    tr.async_checkpoint = True
    tr.checkpoint_dir = str(tmp_path / 'does' / 'not' / 'exist')
    tr.save_checkpoint('c.pt')
    with pytest.raises(RuntimeError, match='checkpoint failed'):
        tr.wait_for_checkpoint()


def test_device_prefetcher_hands_over_every_batch_in_order():
    """One-batch-ahead upload on a side stream: same batches, same order, device batches untouched."""
    from training.trainer import DevicePrefetcher
    host = [(torch.full((2, 3, 1, 8, 8), float(i)).pin_memory(), {'i': i}) for i in range(5)]
    got = list(DevicePrefetcher(host, torch.device('cuda')))
    assert len(got) == 5
    for i, (v, meta) in enumerate(got):
        assert v.is_cuda and meta == {'i': i}
        assert torch.equal(v.cpu(), host[i][0])
    on_device = [(torch.ones(1, 2, 1, 8, 8, device='cuda'), None)]
    out = list(DevicePrefetcher(on_device, torch.device('cuda')))
    assert out[0][0] is on_device[0][0]
    assert list(DevicePrefetcher([], torch.device('cuda'))) == []

def test_graphed_forward_replays_the_eval_forward():
    # the captured no-grad forward returns what the eager one returns, for every batch it is given
    import models
    from hipvae.graph import GraphedForward
    torch.manual_seed(3)
    m = models.get_model('disentangled', in_channels=1, content_dim=16, motion_dim=16, content_channels=[8, 16, 32, 64],
                         motion_channels=[8, 16], decoder_channels=[64, 32, 16, 8], content_feature_dim=64,
                         motion_feature_dim=64, lstm_hidden=128, lstm_layers=1, bidirectional=True,
                         sequence_length=6).cuda().eval()
    clips = [torch.rand(3, 6, 1, 64, 64, device='cuda') for _ in range(3)]
    g = GraphedForward(m, clips[0])
    for v in clips[::-1]:
        got = {k: t.clone() for k, t in g(v).items()}
        with torch.no_grad():
            want = m(v)
        for k in want:
            assert torch.equal(got[k], want[k]), k


def test_a_dead_graph_in_a_reference_cycle_does_not_abort_the_next_capture(monkeypatch):
    """Python's cycle collector may run at any allocation.  If it destroys an old captured step while a new one is
    being captured the runtime refuses and the process aborts (seen in bench.py: a whole run lost in its last
    section); every capture of hipvae.graph therefore collects first and keeps the collector quiet until it has
    ended.  Here: three dead steps in reference cycles, then a capture: none of them may still exist when the captured
    region runs, and the collector must be off inside it.  (tools/debug/gc_probe.py runs the collector inside the
    captured region: 'ok' with the guard, exit status 134 with `--unguarded`.)"""
    import gc
    import weakref
    import models
    from hipvae import losses
    import hipvae.graph as graph_module
    from hipvae.graph import GraphedTrainStep
    from hipvae.optim import FusedAdam
    monkeypatch.setattr(graph_module, '_KEPT', None)      # the session keeps graphs alive; here they must be able to die
    cfg = dict(content_dim=8, motion_dim=8, content_channels=[4, 8, 16, 32], motion_channels=[4, 8],
               decoder_channels=[32, 16, 8, 4], content_feature_dim=32, motion_feature_dim=32, lstm_hidden=16,
               lstm_layers=1, bidirectional=True, sequence_length=6)
    seen = []

    def make(dead_graphs=()):
        m = models.DisentangledVAE(**cfg).cuda().train()
        opt = FusedAdam(m.parameters(), lr=1e-3)
        v = synthetic_clips(4, 6, 3).cuda()

        def loss_of(mm, x):
            if torch.cuda.is_current_stream_capturing():
                seen.append((gc.isenabled(), [r() is not None for r in dead_graphs]))
            return losses.disentangled_loss(mm(x), x, beta=1.0)
        step = GraphedTrainStep(m, opt, loss_of, v)
        step(v)
        return step

    was = gc.isenabled()
    gc.disable()
    try:
        dead = []
        for _ in range(3):
            step = make()
            step.me = step                       # a cycle: only the collector can free it
            dead.append(weakref.ref(step))
            del step
        assert dead[-1]() is not None             # (the earlier ones went when the next step collected before capturing)
    finally:
        if was:
            gc.enable()
    make(dead)
    torch.cuda.synchronize()
    assert seen and all(enabled is False and not any(alive) for enabled, alive in seen), seen
    assert gc.isenabled() == was


def test_graphs_kept_alive_on_request():
    """hipvae.graph.keep_graphs_alive(): every captured graph is held until the process ends (or the switch goes
    off), so that no graph is destroyed while others are still to be built (see its docstring)."""
    import hipvae.graph as graph_module
    import models
    from hipvae.graph import GraphedForward
    before = graph_module._KEPT
    try:
        graph_module.keep_graphs_alive()
        m = models.BaselineVAE(latent_dim=8, encoder_channels=[4, 8, 16, 32], decoder_channels=[32, 16, 8, 4],
                               feature_dim=32, sequence_length=4).cuda().eval()
        v = synthetic_clips(2, 4, 5).cuda()
        g = GraphedForward(m, v)
        assert graph_module._KEPT == [g.graph]
        graph_module.keep_graphs_alive(False)
        assert graph_module._KEPT is None
    finally:
        graph_module._KEPT = before
