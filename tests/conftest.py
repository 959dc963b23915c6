import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "dwc-gan_amd")
for p in (PKG, REPO):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """Tests that start rank processes (torch.distributed.run children) go FIRST: on the GPU boxes a process must not start
    other programs once it has initialised the GPU, and nothing before them in the session has touched it yet."""
    spawning = [it for it in items if "test_two_rank_rccl_training_step" in it.nodeid]
    if spawning:
        items[:] = spawning + [it for it in items if it not in spawning]


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
