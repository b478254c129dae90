"""Command-line entry point: `python -m training.train --config configs/<file>.json --device cuda`
(flags as in the reference, training/train.py:32-101).  Launched under torch.distributed.run it trains
data-parallel, one process per GPU."""
import argparse
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import numpy as np
import torch
import torch.distributed as dist

from training.config import ExperimentConfig, get_baseline_config, get_disentangled_config, get_lstm_config


def set_seed(seed: int):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def parse_args(argv=None):
    p = argparse.ArgumentParser(description='Train Video VAE models (MI355X build)')
    p.add_argument('--config', type=str, default=None, help='JSON config file (overrides the other flags)')
    p.add_argument('--model', type=str, default='disentangled', choices=['baseline', 'lstm', 'disentangled'])
    p.add_argument('--name', type=str, default=None)
    p.add_argument('--train-size', type=int, default=10000)
    p.add_argument('--val-size', type=int, default=1000)
    p.add_argument('--batch-size', type=int, default=32)
    p.add_argument('--num-workers', type=int, default=4)
    p.add_argument('--latent-dim', type=int, default=256)
    p.add_argument('--content-dim', type=int, default=128)
    p.add_argument('--motion-dim', type=int, default=128)
    p.add_argument('--lstm-hidden', type=int, default=256)
    p.add_argument('--epochs', type=int, default=100)
    p.add_argument('--lr', type=float, default=1e-4)
    p.add_argument('--beta', type=float, default=4.0)
    p.add_argument('--warmup', type=int, default=10)
    p.add_argument('--no-kl-annealing', action='store_true')
    p.add_argument('--independence-weight', type=float, default=0.1)
    p.add_argument('--content-agg', type=str, default='mean', choices=['mean', 'middle'])
    p.add_argument('--device', type=str, default='mps', choices=['mps', 'cuda', 'cpu'])
    p.add_argument('--seed', type=int, default=42)
    p.add_argument('--resume', action='store_true')
    p.add_argument('--save-every', type=int, default=10)
    p.add_argument('--visualize-every', type=int, default=5)
    # not in the reference: operand precision of the MFMA kernels (tensors and checkpoints stay fp32)
    p.add_argument('--precision', type=str, default='fp32', choices=['fp32', 'bf16', 'fp32_split'])
    # not in the reference.  The default run replays the training step and the validation forward from captured
    # hipGraphs and renders the Moving-MNIST frames on the GPU (bit-identical clips, no DataLoader workers, nothing
    # but digit indices and positions crosses PCIe); these flags select the slower host-driven forms.
    p.add_argument('--no-hip-graph', action='store_true', help='launch every kernel of a step from the host')
    p.add_argument('--host-render', action='store_true', help='render the clips in DataLoader worker processes')
    p.add_argument('--hip-graph', action='store_true', help='(default; kept for older command lines)')
    p.add_argument('--gpu-render', action='store_true', help='(default; kept for older command lines)')
    # not in the reference: digit images from a .npy file ([N,28,28], uint8 or float in [0,1]) instead of
    # MNIST, for machines without the dataset
    p.add_argument('--digit-bank', type=str, default=None)
    return p.parse_args(argv)


def config_from_args(args) -> ExperimentConfig:
    if args.config is not None:
        print(f"Loading config from: {args.config}")
        cfg = ExperimentConfig.load(args.config)
        if args.device:                      # always true: the flag wins over the file (reference :113-114)
            cfg.training.device = args.device
        return cfg
    cfg = {'baseline': get_baseline_config, 'lstm': get_lstm_config}.get(args.model, get_disentangled_config)()
    cfg.name = args.name if args.name is not None else f"{args.model}_vae"
    d, m, t = cfg.data, cfg.model, cfg.training
    d.train_size, d.val_size, d.batch_size, d.num_workers, d.seed = (args.train_size, args.val_size, args.batch_size,
                                                                     args.num_workers, args.seed)
    m.latent_dim, m.content_dim, m.motion_dim, m.lstm_hidden = (args.latent_dim, args.content_dim, args.motion_dim,
                                                                args.lstm_hidden)
    m.content_aggregation = args.content_agg
    t.num_epochs, t.learning_rate, t.beta_vae, t.warmup_epochs = args.epochs, args.lr, args.beta, args.warmup
    t.use_kl_annealing = not args.no_kl_annealing
    t.independence_weight = args.independence_weight
    t.device = args.device
    t.save_every, t.visualize_every = args.save_every, args.visualize_every
    return cfg


def main(argv=None):
    args = parse_args(argv)
    cfg = config_from_args(args)
    world = int(os.environ.get('WORLD_SIZE', 1))
    rank = int(os.environ.get('RANK', 0))
    if world > 1:
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)))
        dist.init_process_group('nccl')
    set_seed(cfg.data.seed)
    if args.precision != 'fp32':
        import hipvae
        hipvae.set_precision(args.precision)
    if rank == 0:
        print(f"Model type: {cfg.model.model_type}")
        print(f"Data: train {cfg.data.train_size}, val {cfg.data.val_size}, batch {cfg.data.batch_size}"
              f"{f' per GPU x {world}' if world > 1 else ''}")
        print(f"Training: {cfg.training.num_epochs} epochs, lr {cfg.training.learning_rate}, "
              f"beta {cfg.training.beta_vae}, KL annealing {cfg.training.use_kl_annealing}, "
              f"device {cfg.training.device}")
    from data import get_moving_mnist_dataloaders
    from training.trainer import create_trainer
    bank = None
    if args.digit_bank is not None:
        import numpy as np
        bank = np.load(args.digit_bank)
        bank = bank.astype(np.float32) / 255.0 if bank.dtype == np.uint8 else bank.astype(np.float32)
    train_loader, val_loader, _ = get_moving_mnist_dataloaders(
        digit_bank=bank,
        root=cfg.data.root, batch_size=cfg.data.batch_size, train_size=cfg.data.train_size,
        val_size=cfg.data.val_size, test_size=cfg.data.test_size, sequence_length=cfg.data.sequence_length,
        frame_size=cfg.data.frame_size, num_digits=cfg.data.num_digits, num_workers=cfg.data.num_workers,
        seed=cfg.data.seed, rank=rank, world_size=world,
        pin_memory=args.host_render,             # host batches are uploaded asynchronously, one batch ahead
        render_device=None if args.host_render else torch.device('cuda', torch.cuda.current_device()))
    trainer = create_trainer(cfg, train_loader, val_loader)
    if args.no_hip_graph:
        trainer.use_graph = False
    history = trainer.train(resume=args.resume)
    if rank == 0:
        print(f"Results saved to: {trainer.results_dir}")
        print(f"Checkpoints saved to: {trainer.checkpoint_dir}")
    if world > 1:
        dist.destroy_process_group()
    return history


if __name__ == '__main__':
    main()
