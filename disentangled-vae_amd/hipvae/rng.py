"""Source of the standard-normal draws used by reparameterisation.

Default: torch.randn_like on the tensor's device (device Philox stream).  Parity tests and
reproducible runs inject prepared draws instead, consumed in call order (content before motion
for the disentangled model, as in the reference's encode()).
"""
from contextlib import contextmanager
from typing import List, Optional

import torch

_injected: Optional[List[torch.Tensor]] = None
_slots: Optional[List[torch.Tensor]] = None       # collects the draw buffers of a step run under draw_slots()
_prepared: Optional[List[torch.Tensor]] = None    # buffers to hand out, in call order, instead of new ones


def injecting() -> bool:
    return _injected is not None


@contextmanager
def draw_slots(prepared: Optional[List[torch.Tensor]] = None):
    """For a step that is recorded into a hipGraph in a run with injected draws.  Every draw_like() inside the
    block returns a buffer and notes it in the list this yields: a new zero-filled one (a rehearsal of the step,
    which tells how many draws of which shapes it makes), or the next of `prepared`.  The buffers a captured step
    reads must be `prepared` ones allocated BEFORE capture began: memory allocated during capture is shared with
    tensors of the same graph whose lifetimes ended earlier, so it is overwritten by every replay before the
    draw is read.  The owner of the graph copies the next injected draws into them before each replay
    (fill_slots)."""
    global _slots, _prepared
    prev = (_slots, _prepared)
    _slots, _prepared = [], (list(prepared) if prepared is not None else None)
    try:
        yield _slots
        if _prepared:
            raise RuntimeError(f"{len(_prepared)} prepared draw buffer(s) were not used by the step")
    finally:
        _slots, _prepared = prev


def fill_slots(slots: List[torch.Tensor]) -> None:
    """The next injected draws, in call order, into the buffers a captured step reads."""
    for buf in slots:
        buf.copy_(draw_like(buf), non_blocking=True)


def draw_like(t: torch.Tensor) -> torch.Tensor:
    global _injected
    if _slots is not None:
        if _prepared is None:
            buf = torch.zeros_like(t)
        else:
            if not _prepared:
                raise RuntimeError("the step draws more often than its rehearsal did")
            buf = _prepared.pop(0)
            if buf.shape != t.shape or buf.device != t.device or buf.dtype != t.dtype:
                raise RuntimeError(f"prepared draw buffer {tuple(buf.shape)} does not fit a draw of {tuple(t.shape)}")
        _slots.append(buf)
        return buf
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
