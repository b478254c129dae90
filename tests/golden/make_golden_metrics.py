"""Golden values for the evaluation metrics, produced by the reference's own functions on CPU.

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_metrics.py

The reference module is loaded by file path; scikit-image, which it imports for its host-side SSIM and
which this image lacks, is replaced by an inert double (SSIM is not generated).  Inputs are small
seeded tensors stored next to the expected values.
"""
import importlib.util
import os
import sys
import types

import numpy as np
import torch

sys.dont_write_bytecode = True
REF = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    sk = types.ModuleType('skimage')
    skm = types.ModuleType('skimage.metrics')
    skm.structural_similarity = lambda *a, **k: float('nan')          # never called here
    sk.metrics = skm
    sys.modules['skimage'] = sk
    sys.modules['skimage.metrics'] = skm
    spec = importlib.util.spec_from_file_location('ref_metrics', os.path.join(REF, 'evaluation', 'metrics.py'))
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)

    g = torch.Generator().manual_seed(77)
    arrays, expect = {}, {}
    for name, (B, T, S) in {'small': (2, 5, 16), 'one_frame': (3, 1, 8), 'clip': (1, 20, 64)}.items():
        target = (torch.rand(B, T, 1, S, S, generator=g) > 0.9).float() * torch.rand(B, T, 1, S, S, generator=g)
        recon = torch.sigmoid(torch.randn(B, T, 1, S, S, generator=g) * 2)
        arrays[f'{name}/recon'] = recon.numpy()
        arrays[f'{name}/target'] = target.numpy()
        expect[name] = [ref.mse(recon, target), ref.psnr(recon, target), ref.temporal_consistency(recon)]
    same = torch.rand(2, 3, 1, 8, 8, generator=g)
    arrays['identical/recon'] = same.numpy()
    arrays['identical/target'] = same.numpy()
    expect['identical'] = [ref.mse(same, same), ref.psnr(same, same), ref.temporal_consistency(same)]
    arrays['half_range/recon'] = arrays['small/recon']
    arrays['half_range/target'] = arrays['small/target']
    r, t = torch.from_numpy(arrays['small/recon']), torch.from_numpy(arrays['small/target'])
    expect['half_range'] = [ref.mse(r, t), ref.psnr(r, t, max_val=0.5), ref.temporal_consistency(r)]
    for k, v in expect.items():
        arrays[f'{k}/expect'] = np.asarray(v, dtype=np.float64)
        print(k, v)
    np.savez_compressed(os.path.join(HERE, 'metrics.npz'), **arrays)


if __name__ == '__main__':
    main()
