import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'disentangled-vae_amd')
for p in (ROOT, PKG, os.path.join(ROOT, 'tests')):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')

# A test session captures and drops hundreds of graphs in one process: they stay alive until it ends (a few GiB of
# pools), because it is the destruction of graphs that can leave the ROCm 7.0 runtime in a state in which a later
# hipGraphLaunch faults (hipvae.graph.keep_graphs_alive).  DVAE_KEEP_GRAPHS=0 in the environment switches it off.
os.environ.setdefault('DVAE_KEEP_GRAPHS', '1')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


def pytest_sessionfinish(session, exitstatus):
    # how much device memory the session held at its peak (graphs kept alive hold their pools), on stderr
    try:
        import torch
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            print(f'\n[tests] peak device memory reserved: {torch.cuda.max_memory_reserved() / 2 ** 30:.1f} GiB',
                  file=sys.stderr)
    except Exception:                                               # noqa: BLE001
        pass
