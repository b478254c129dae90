"""Diagnostic: where the DisentangledVAE gradient of the HIP path departs from the fp64 oracle at a given
batch size.  Prints per-tensor errors and the errors of the gradients flowing into the latents."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'disentangled-vae_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np
import torch
from oracle import vae_oracle as O
import models
from hipvae import rng
from test_models_gpu import FULL, SETTINGS, loss_of, rel

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iw = float(sys.argv[2]) if len(sys.argv) > 2 else 0.02
kind, T, size = 'disentangled', 20, 64
torch.manual_seed(0)
m = models.get_model(kind, in_channels=1, **FULL[kind])
sd = {k: v.clone() for k, v in m.state_dict().items()}
gen = torch.Generator().manual_seed(1)
video = (torch.rand(B, T, 1, size, size, generator=gen) > 0.9).float() * torch.rand(B, T, 1, size, size, generator=gen)
eps = [torch.randn(B, 128, generator=torch.Generator().manual_seed(100 + k)) for k in range(2)]
tcfg = dict(SETTINGS['tcfg'][kind], beta_content=4.0, beta_motion=4.0, independence_weight=iw)
LAT = ('z_content', 'z_motion', 'mu_content', 'logvar_content', 'mu_motion', 'logvar_motion', 'recon')


def oracle(dtype):
    sdx = {k: (v.to(dtype) if v.is_floating_point() else v) for k, v in sd.items()}
    keys = O.trainable_keys(sdx)
    leaves = {k: sdx[k].clone().requires_grad_(True) for k in keys}
    work = dict(sdx)
    work.update(leaves)
    out = O.model_forward(kind, work, video.to(dtype), [e.to(dtype) for e in eps], True, lstm_layers=1, new_stats={})
    for k in LAT:
        out[k].retain_grad()
    loss = O.model_loss(kind, out, video.to(dtype), 4.0, tcfg)
    loss['total_loss'].backward()
    return keys, {k: leaves[k].grad for k in keys}, {k: out[k].grad for k in LAT}, out


keys, g64, l64, o64 = oracle(torch.float64)
_, g32, l32, _ = oracle(torch.float32)
m = m.cuda().train()
with rng.inject_eps(eps):
    out = m(video.cuda())
for k in LAT:
    out[k].retain_grad()
losses = loss_of(kind, out, video.cuda(), tcfg, 4.0)
losses['total_loss'].backward()
named = dict(m.named_parameters())
print(f'B={B} iw={iw}')
print('gradients reaching the latents / reconstruction (hip err, fp32-cpu err):')
for k in LAT:
    print(f'  {k:16s} {rel(out[k].grad.cpu().numpy(), l64[k].numpy()):.2e}  {rel(l32[k].numpy(), l64[k].numpy()):.2e}'
          f'   fwd {rel(out[k].detach().cpu().numpy(), o64[k].detach().numpy()):.2e}')
print('parameters:')
for k in keys:
    e, own = rel(named[k].grad.cpu().numpy(), g64[k].numpy()), rel(g32[k].numpy(), g64[k].numpy())
    print(f'  {k:55s} {e:.2e}  {own:.2e} {"  <-- " if e > max(1e-4, 2 * own) else ""}')

# ---- the content encoder's feature gradient: dfeat = dmu W_mu + dlv W_lv
print('content feature gradient:')
sd64 = {k: v.double() for k, v in sd.items() if v.is_floating_point()}
Wm, Wl = sd64['content_encoder.vae_head.fc_mu.weight'], sd64['content_encoder.vae_head.fc_logvar.weight']
ref_total = l64['mu_content'] @ Wm + l64['logvar_content'] @ Wl
hip_dmu, hip_dlv = out['mu_content'].grad.double().cpu(), out['logvar_content'].grad.double().cpu()
ref_from_hip_inputs = hip_dmu @ Wm + hip_dlv @ Wl
m.zero_grad()
feats = {}
orig_fwd = m.content_encoder.cnn_encoder.forward
def fwd(x):
    f = orig_fwd(x)
    f.register_hook(lambda g: feats.__setitem__('dfeat', g.detach().clone()))
    feats['feat'] = f.detach().clone()
    return f
m.content_encoder.cnn_encoder.forward = fwd
with rng.inject_eps(eps):
    out2 = m(video.cuda())
loss_of(kind, out2, video.cuda(), tcfg, 4.0)['total_loss'].backward()
d = feats['dfeat'].double().cpu()
print(f'  hip dfeat vs fp64 oracle            {rel(d.numpy(), ref_total.numpy()):.2e}')
print(f'  hip dfeat vs fp64 of its own inputs {rel(d.numpy(), ref_from_hip_inputs.numpy()):.2e}')
print(f'  fp64(hip inputs) vs fp64 oracle     {rel(ref_from_hip_inputs.numpy(), ref_total.numpy()):.2e}')
print(f'  |dmu W_mu| {float((hip_dmu @ Wm).norm()):.3e} |dlv W_lv| {float((hip_dlv @ Wl).norm()):.3e} |sum| {float(ref_total.norm()):.3e}')
from hipvae import ops
g = torch.Generator().manual_seed(3)
for (M, N, K) in ((256, 512, 128), (32, 512, 128)):
    a, b = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g)
    got = ops.gemm_nn(a.cuda(), b.cuda())
    print(f'  gemm_nn {M}x{N}x{K}: {rel(got.cpu().numpy(), (a.double() @ b.double()).numpy()):.2e}')

# ---- the content CNN's backward in isolation: fed the fp64 oracle's feature gradient (rounded to fp32)
m.zero_grad()
with rng.inject_eps(eps):
    out3 = m(video.cuda())
f3 = feats['feat']
cparams = {k: p for k, p in m.named_parameters() if k.startswith('content_encoder.cnn_encoder')}
# rebuild a graph handle to the feature: run the CNN alone on the mean frame
from hipvae import ops as _ops
mean, _ = _ops.split_mean_diff(video.cuda().contiguous())
m.content_encoder.cnn_encoder.forward = orig_fwd
feat = m.content_encoder.cnn_encoder(mean)
gs = torch.autograd.grad(feat, list(cparams.values()), grad_outputs=ref_total.float().cuda())
print('content CNN backward alone, fed the oracle feature gradient:')
for (k, p), g_ in zip(cparams.items(), gs):
    print(f'  {k:55s} {rel(g_.cpu().numpy(), g64[k].numpy()):.2e}')
gs = torch.autograd.grad(m.content_encoder.cnn_encoder(mean), list(cparams.values()), grad_outputs=d.float().cuda())
print('content CNN backward alone, fed the HIP feature gradient:')
for (k, p), g_ in zip(cparams.items(), gs):
    print(f'  {k:55s} {rel(g_.cpu().numpy(), g64[k].numpy()):.2e}')
dd = (d - ref_total)
print('error of HIP dfeat: norm', float(dd.norm()), 'norm of its batch mean x sqrt(B)', float(dd.mean(0).norm()) * B ** 0.5,
      '| signal: norm', float(ref_total.norm()), 'batch-mean part', float(ref_total.mean(0).norm()) * B ** 0.5)
d32 = (l32['mu_content'].double() @ Wm + l32['logvar_content'].double() @ Wl) - ref_total
print('error of fp32-CPU dfeat: norm', float(d32.norm()), 'norm of its batch mean x sqrt(B)', float(d32.mean(0).norm()) * B ** 0.5)

# ---- the bias/activation backward alone
sd64all = {k: (v.double() if v.is_floating_point() else v) for k, v in sd.items()}
mean64 = video.double().mean(1)
feat64 = O.cnn_encoder(mean64, sd64all, 'content_encoder.cnn_encoder', True, {})
slope = torch.where(feat64 > 0, torch.ones_like(feat64), torch.full_like(feat64, 0.2))
ref_bias = (ref_total * slope).sum(0)
print('bias/activation backward alone:')
print(f'  my fp64 formula vs oracle autograd: {rel(ref_bias.numpy(), g64["content_encoder.cnn_encoder.fc.0.bias"].numpy()):.2e}')
fh = feats['feat']
print(f'  feat hip vs fp64: {rel(fh.cpu().numpy(), feat64.numpy()):.2e}; sign mismatches: '
      f'{int(((fh.cpu() > 0) != (feat64 > 0)).sum())} of {fh.numel()}; |feat64| < 1e-5: {int((feat64.abs() < 1e-5).sum())}')
dpre = ref_total.float().cuda().clone()
db = _ops.bias_act_bwd(dpre, fh.contiguous(), _ops.ACT_LRELU, True)
print(f'  hip bias_act_bwd: dbias {rel(db.cpu().numpy(), ref_bias.numpy()):.2e}, dpre {rel(dpre.cpu().numpy(), (ref_total * slope).numpy()):.2e}')
