"""Where do the one-launch LSTM kernels run into their wait bound?  Prints the per-GPU counter after each phase."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'disentangled-vae_amd'))
from hipvae import ops, losses as L
from hipvae.optim import FusedAdam
from hipvae.graph import GraphedTrainStep
from models import get_model
from training.config import ExperimentConfig

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
cfg = ExperimentConfig.load(os.path.join(ROOT, 'disentangled-vae_amd', 'configs', 'disentangled_simple.json'))
torch.manual_seed(0)
print('start', ops.lstm_seq_timeouts(), flush=True)
m = cfg.model
model = get_model('disentangled', content_dim=m.content_dim, motion_dim=m.motion_dim, content_channels=m.encoder_channels,
                  motion_channels=m.motion_channels, decoder_channels=m.decoder_channels, content_feature_dim=m.feature_dim,
                  motion_feature_dim=m.feature_dim, lstm_hidden=m.lstm_hidden, lstm_layers=m.lstm_layers,
                  bidirectional=m.bidirectional, sequence_length=20).cuda()
opt = FusedAdam(model.parameters(), lr=1e-3)
x = (torch.rand(B, 20, 1, 64, 64) > 0.9).float().cuda() * torch.rand(B, 20, 1, 64, 64).cuda()


def loss_fn(out, tgt):
    return L.disentangled_loss(out, tgt, beta=4.0, recon_loss_type='bce', temporal_weight=0.05, independence_weight=0.02)


for i in range(3):
    opt.zero_grad()
    out = model(x)
    print(' fwd', i, ops.lstm_seq_timeouts(), flush=True)
    l = loss_fn(out, x)
    l['total_loss'].backward()
    print(' bwd', i, ops.lstm_seq_timeouts(), flush=True)
    opt.step()
t0 = time.time()
g = GraphedTrainStep(model, opt, lambda mdl, v: loss_fn(mdl(v), v), x, gradient_clip=5.0)
print('captured', ops.lstm_seq_timeouts(), time.time() - t0, flush=True)
for i in range(5):
    t0 = time.time()
    g(x)
    torch.cuda.synchronize()
    print(' replay', i, ops.lstm_seq_timeouts(), round((time.time() - t0) * 1e3, 2), 'ms', flush=True)
