"""Host-side pieces of the drop-in surface, checked on CPU against what the reference itself produced
(tests/golden/*.json): configuration files, schedules, generator, state_dict layout, CLI plumbing."""
import hashlib
import json
import os

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), 'golden')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONFIGS = os.path.join(ROOT, 'disentangled-vae_amd', 'configs')


def test_config_roundtrip_and_strictness(tmp_path):
    from training.config import ExperimentConfig, ModelConfig, get_ablation_configs, get_disentangled_config
    for fn in os.listdir(CONFIGS):
        cfg = ExperimentConfig.load(os.path.join(CONFIGS, fn))
        raw = json.load(open(os.path.join(CONFIGS, fn)))
        assert cfg.to_dict() == {**raw, 'model': {**ModelConfig().__dict__, **raw['model']}}
        p = tmp_path / 'sub' / fn
        cfg.save(str(p))
        assert ExperimentConfig.load(str(p)).to_dict() == cfg.to_dict()
    # lstm_simple.json omits content_dim etc.: defaults apply
    cfg = ExperimentConfig.load(os.path.join(CONFIGS, 'lstm_simple.json'))
    assert cfg.model.content_dim == 128 and cfg.model.lstm_layers == 2
    bad = json.load(open(os.path.join(CONFIGS, 'lstm_simple.json')))
    bad['model']['not_a_field'] = 1
    (tmp_path / 'bad.json').write_text(json.dumps(bad))
    with pytest.raises(TypeError):
        ExperimentConfig.load(str(tmp_path / 'bad.json'))
    assert get_disentangled_config().training.independence_weight == 0.1
    assert sorted(get_ablation_configs()) == sorted(
        ['beta_1.0', 'beta_2.0', 'beta_4.0', 'beta_8.0', 'beta_16.0', 'dim_64', 'dim_128', 'dim_256', 'agg_mean',
         'agg_middle', 'bidir_True', 'bidir_False'])


def test_beta_and_lr_schedules_match_reference():
    from training.config import ExperimentConfig, TrainingConfig
    from training.trainer import beta_for_epoch
    sched = json.load(open(os.path.join(GOLDEN, 'schedules.json')))
    for fn, rec in sched.items():
        if fn.startswith('synthetic_'):
            t = TrainingConfig(**rec['cfg'])
            got = [beta_for_epoch(t, e) for e in range(len(rec['beta']))]
            assert got == rec['beta']
            continue
        cfg = ExperimentConfig.load(os.path.join(CONFIGS, fn))
        t = cfg.training
        assert [beta_for_epoch(t, e) for e in range(t.num_epochs)] == rec['beta']
        # the LR rule of Trainer.train: step the scheduler only while lr > lr_min
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([p], lr=t.learning_rate)
        sch = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=t.num_epochs, eta_min=t.lr_min) \
            if t.lr_scheduler == 'cosine' else torch.optim.lr_scheduler.ExponentialLR(opt, gamma=t.lr_decay)
        lrs = []
        for _ in range(t.num_epochs):
            opt.step()
            if opt.param_groups[0]['lr'] > t.lr_min:
                sch.step()
            lrs.append(opt.param_groups[0]['lr'])
        np.testing.assert_allclose(lrs, rec['lr'], rtol=1e-12)


def test_generator_is_bit_exact():
    from data.moving_mnist import MovingMNIST, get_moving_mnist_dataloaders
    from tests.golden.make_golden import synthetic_digit_bank
    bank = synthetic_digit_bank()
    info = json.load(open(os.path.join(GOLDEN, 'generator.json')))
    vids = np.load(os.path.join(GOLDEN, 'generator_videos.npz'))
    assert hashlib.sha1(bank.tobytes()).hexdigest() == info['bank_sha1']
    for it in info['items']:
        ds = MovingMNIST(num_videos=32, seed=it['seed'], deterministic=True, digit_bank=bank, **it['kwargs'])
        v, meta = ds[it['index']]
        assert v.dtype == torch.float32 and list(v.shape) == it['shape']
        assert hashlib.sha1(v.numpy().tobytes()).hexdigest() == it['sha1']
        assert np.array_equal(v.numpy(), vids[f"video_s{it['seed']}_i{it['index']}"].astype(np.float32) / 255.0)
        assert [int(x) for x in meta['labels']] == it['labels']
        assert [p.tolist() for p in meta['initial_positions']] == it['initial_positions']
        assert [p.tolist() for p in meta['initial_velocities']] == it['initial_velocities']
        assert [p.tolist() for p in meta['position_history'][-1]] == it['last_positions']
        assert [p.tolist() for p in meta['velocity_history'][-1]] == it['last_velocities']
        v2, _ = ds[it['index']]
        assert torch.equal(v, v2)                               # repeatable indexing
    # loaders: split seeds (+0, +1000, +2000), batch shape, metadata list
    tr, va, te = get_moving_mnist_dataloaders(batch_size=4, train_size=8, val_size=4, test_size=4, num_workers=0,
                                              seed=42, digit_bank=bank)
    assert (tr.dataset.seed, va.dataset.seed, te.dataset.seed) == (42, 1042, 2042)
    batch, meta = next(iter(va))
    assert batch.shape == (4, 20, 1, 64, 64) and len(meta) == 4 and 'position_history' in meta[0]
    ref = MovingMNIST(num_videos=4, seed=1042, deterministic=True, digit_bank=bank)
    assert torch.equal(batch[2], ref[2][0])
    assert float(batch.min()) >= 0 and float(batch.max()) <= 1


def test_simulate_is_the_host_generator_without_rendering():
    """MovingMNIST.simulate (what the GPU renderer is fed) makes the draws of generate_video: same digit
    images, same trajectory metadata, and its integer positions re-render to the same frames."""
    from data.moving_mnist import MovingMNIST
    from tests.golden.make_golden import synthetic_digit_bank
    bank = synthetic_digit_bank()
    for kwargs, seed in ((dict(sequence_length=20, frame_size=64, num_digits=2), 42),
                         (dict(sequence_length=7, frame_size=128, num_digits=3), 0),
                         (dict(sequence_length=40, frame_size=32, num_digits=1), 7)):
        ds = MovingMNIST(num_videos=16, seed=seed, deterministic=True, digit_bank=bank, **kwargs)
        for idx in (0, 3, 15):
            video, meta = ds[idx]
            rng = np.random.RandomState(seed + idx if seed else idx)
            didx, ipos, meta2 = ds.simulate(rng)
            assert didx.dtype == np.int32 and ipos.dtype == np.int32
            assert ipos.shape == (kwargs['sequence_length'], kwargs['num_digits'], 2)
            assert [int(x) for x in meta2['labels']] == [int(x) for x in meta['labels']]
            for a, b in zip(meta['position_history'], meta2['position_history']):
                assert all(np.array_equal(x, y) for x, y in zip(a, b))
            for t in range(kwargs['sequence_length']):
                frame = ds._render_frame([bank[i] for i in didx], [ipos[t, i].astype(np.float64)
                                                                   for i in range(kwargs['num_digits'])])
                assert np.array_equal(frame, video[t, 0].numpy())
        # the batched, vectorised form used by the device loader: identical draws and trajectories
        ids = [0, 3, 15, 7, 8]
        bidx, bpos, bmeta = ds.simulate_batch(ids, with_meta=True)
        for j, idx in enumerate(ids):
            didx, ipos, meta = ds.simulate(np.random.RandomState(seed + idx if seed else idx))
            assert np.array_equal(bidx[j], didx) and np.array_equal(bpos[j], ipos)
            assert [int(x) for x in bmeta[j]['labels']] == [int(x) for x in meta['labels']]
            for name in ('position_history', 'velocity_history'):
                for a, b in zip(meta[name], bmeta[j][name]):
                    assert all(np.array_equal(x, y) for x, y in zip(a, b)), name
        assert ds.simulate_batch(ids, with_meta=False)[2] == [{}] * len(ids)


def test_missing_mnist_is_a_clear_error(tmp_path):
    from data.moving_mnist import MovingMNIST
    try:
        import torchvision  # noqa: F401
        pytest.skip('torchvision present: it would try to download')
    except ImportError:
        pass
    with pytest.raises(FileNotFoundError):
        MovingMNIST(root=str(tmp_path), num_videos=2)


def test_state_dict_layout_and_parameter_counts():
    import models
    from training.config import ExperimentConfig
    from training.trainer import model_kwargs_from_config
    man = json.load(open(os.path.join(GOLDEN, 'manifests.json')))
    for fn, rec in man.items():
        if fn == 'config5_128x128x40':
            m = models.DisentangledVAE(content_channels=[64, 128, 256, 512, 512], motion_channels=[32, 64, 64],
                                       decoder_channels=[512, 512, 256, 128, 64], sequence_length=40)
        else:
            cfg = ExperimentConfig.load(os.path.join(CONFIGS, fn))
            m = models.get_model(cfg.model.model_type, **model_kwargs_from_config(cfg))
            assert m.model_type == rec['model_type']
            assert len(list(m.parameters())) == rec['n_trainable_tensors']
        sd = m.state_dict()
        assert list(sd.keys()) == list(rec['keys'].keys())
        assert {k: list(v.shape) for k, v in sd.items()} == rec['keys']
        assert sum(p.numel() for p in m.parameters() if p.requires_grad) == rec['n_params']
    assert man['disentangled_simple.json']['n_params'] == 12934625
    assert man['lstm_simple.json']['n_params'] == 10441857


def test_same_seed_gives_the_reference_initialisation():
    # parameter containers are created in the reference's order, so the default init matches draw for draw
    import models
    g = np.load(os.path.join(GOLDEN, 'trainer_run.npz'))
    torch.manual_seed(5)
    m = models.BaselineVAE(in_channels=1, sequence_length=6, latent_dim=16, encoder_channels=[4, 8, 16, 32],
                           decoder_channels=[32, 16, 8, 4], feature_dim=32)
    for k, v in m.state_dict().items():
        assert np.array_equal(v.numpy(), g['sd0/' + k]), k


def test_errors_match_the_reference_contract():
    import models
    with pytest.raises(ValueError):
        models.get_model('nope')
    with pytest.raises(ValueError):
        models.get_loss_fn('nope')
    from hipvae import losses
    with pytest.raises(ValueError):
        losses._kind('huber')
    # models run on the GPU only
    m = models.BaselineVAE(latent_dim=16, encoder_channels=[4, 8, 16, 32], decoder_channels=[32, 16, 8, 4],
                           feature_dim=32)
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 2, 1, 64, 64))


def test_cli_config_assembly(tmp_path):
    from training.train import config_from_args, parse_args
    a = parse_args(['--config', os.path.join(CONFIGS, 'disentangled_simple.json'), '--device', 'cuda'])
    cfg = config_from_args(a)
    assert cfg.training.device == 'cuda' and cfg.training.beta_vae == 0.4 and cfg.model.model_type == 'disentangled'
    a = parse_args(['--config', os.path.join(CONFIGS, 'disentangled_simple.json')])
    assert config_from_args(a).training.device == 'mps'      # the flag's default overrides the file, as upstream
    a = parse_args(['--model', 'lstm', '--beta', '2.0', '--epochs', '3', '--no-kl-annealing'])
    cfg = config_from_args(a)
    assert cfg.name == 'lstm_vae' and cfg.training.beta_vae == 2.0 and not cfg.training.use_kl_annealing
    assert cfg.model.model_type == 'lstm' and cfg.training.num_epochs == 3


def test_trainer_refuses_to_run_without_a_gpu(tmp_path):
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from training.config import get_baseline_config
    from training.trainer import create_trainer
    cfg = get_baseline_config('t')
    cfg.training.device = 'cuda'
    cfg.training.checkpoint_dir = str(tmp_path / 'c')
    cfg.training.results_dir = str(tmp_path / 'r')
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        create_trainer(cfg, [], [])
