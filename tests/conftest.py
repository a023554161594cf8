import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dfa-nerf_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return dict(np.load(os.path.join(GOLDEN, name + ".npz")))
    return load


@pytest.fixture(scope="session")
def scene():
    from dfanerf import synth
    return synth.bench_scene(0, n_frames=8)


@pytest.fixture(scope="session")
def states():
    from dfanerf import synth
    return synth.synth_all_states(0)


@pytest.fixture(scope="session")
def latents():
    from dfanerf import synth
    return synth.synth_latents(0)


def free_port():
    """a TCP port nobody listens on right now (torch.distributed rendezvous of the multi-process tests)"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port
