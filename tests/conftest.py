"""pytest configuration: `gpu` marker, repo-root import path, fixture locations."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# Several "devices" of a multi-device context may be ONE GPU on this pool (tests/test_gpu_partition.py): the ranks' persistent launches then come from several
# streams of one process, and the HIP runtime multiplexes a process's streams onto 4 hardware queues by default -- two launches that wait for each other on one
# queue never overlap (the context notices: hand-off time-out -> element form).  Real devices have queues of their own; for the shared-GPU tests the runtime
# is asked for more queues, before anything initialises it.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure).  Built on demand with the committed Makefile."""
    from oracle import oracle as o

    o.build()
    return o


@pytest.fixture(scope="session")
def mesh_loader(oracle):
    cache = {}

    def load(name):
        if name not in cache:
            cache[name] = oracle.load_mesh(os.path.join(GOLDEN, "mesh", name))
        return cache[name]

    return load
