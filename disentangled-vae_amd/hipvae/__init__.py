"""MI355X (gfx950) kernels and autograd nodes for the video-VAE hot path.

Importing this package loads libdvae_hip.so; if it has not been built the import fails loudly.
"""
from . import _lib  # noqa: F401  (raises ImportError when the shared library is missing)
from . import ops, functional, losses, rng  # noqa: F401
from .ops import set_precision, get_precision  # noqa: F401

__all__ = ['ops', 'functional', 'losses', 'rng', 'set_precision', 'get_precision']
