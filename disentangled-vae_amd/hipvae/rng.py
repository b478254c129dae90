"""Source of the standard-normal draws used by reparameterisation.

Default: torch.randn_like on the tensor's device (device Philox stream).  Parity tests and
reproducible runs inject prepared draws instead, consumed in call order (content before motion
for the disentangled model, as in the reference's encode()).
"""
from contextlib import contextmanager
from typing import List, Optional

import torch

_injected: Optional[List[torch.Tensor]] = None


def draw_like(t: torch.Tensor) -> torch.Tensor:
    global _injected
    if _injected is not None:
        if not _injected:
            raise RuntimeError("injected eps list exhausted")
        e = _injected.pop(0)
        if e.shape != t.shape:
            raise RuntimeError(f"injected eps has shape {tuple(e.shape)}, expected {tuple(t.shape)}")
        return e.to(device=t.device, dtype=t.dtype)
    return torch.randn_like(t)


@contextmanager
def inject_eps(eps_list):
    global _injected
    prev = _injected
    _injected = list(eps_list)
    try:
        yield
    finally:
        _injected = prev
