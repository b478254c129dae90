"""The C-ABI library loads on a GPU-less host and exports every symbol include/dvae_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, 'include', 'dvae_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(dvae_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from hipvae import _lib
    names = declared_symbols()
    assert len(names) > 30
    for n in names:
        assert hasattr(_lib.lib, n), f'{n} declared in include/dvae_hip.h but not exported'
    # and the binding table covers the header one to one
    assert sorted(_lib.SIGNATURES) == names


def test_version_and_error_string():
    from hipvae import _lib
    assert _lib.lib.dvae_version() >= 1
    assert isinstance(_lib.lib.dvae_last_error(), (bytes, type(None)))


def test_argument_validation_happens_before_any_launch():
    # shape checks are host-side: they must reject bad arguments without touching a GPU
    from hipvae import _lib
    rc = _lib.lib.dvae_gemm_nt_f32(None, 6, None, 6, None, 0, None, 8, 4, 8, 6, None, None)
    assert rc != 0
    assert b'multiples of 4' in _lib.lib.dvae_last_error()
    rc = _lib.lib.dvae_conv4s2_fwd_f32(None, None, None, 0, None, None, None, 2, 48, 48, 4, 8, None)
    assert rc != 0 and b'powers of two' in _lib.lib.dvae_last_error()


def test_ops_refuse_cpu_tensors():
    import pytest
    import torch
    from hipvae import ops
    with pytest.raises(RuntimeError):
        ops.split_mean_diff(torch.zeros(2, 3, 1, 64, 64))
