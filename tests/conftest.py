"""pytest configuration: registers the `gpu` marker and puts the repo root on sys.path."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The engine's default solves the 5 x 5 coarsest grid of W- / F-cycles directly (MultigridEngine(coarse_direct=None) ->
# "auto"): within 1e-12 of the reference's iterates, not bit-identical.  The suite's many bit-for-bit comparisons between
# engine paths (fused legs vs one launch per operator, decomposed vs single domain, kernels vs oracle) are statements about
# the reference-faithful iteration, so the suite pins it; tests of the default pass coarse_direct="auto" explicitly.
os.environ.setdefault("MG_COARSE_DIRECT", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    """A machine with the AMD compute device node is a GPU box: there the gpu-marked tests RUN (and fail
    loudly if the runtime is broken) instead of being skipped on a flaky availability probe."""
    return os.path.exists("/dev/kfd")


def pytest_collection_modifyitems(config, items):
    """A gpu-marked test on a machine without a GPU is an error of selection, not a skip:
    the driver selects with -m; anything else that lands here without a device is skipped
    loudly so a CPU-only `pytest tests/` stays green."""
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible (gpu-marked tests run on the MI355X box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_ops():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "ops.npz"))


@pytest.fixture(scope="session")
def golden_solves():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "solves.npz"))


@pytest.fixture(scope="session")
def golden_large():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "large_1025.npz"))


@pytest.fixture(scope="session")
def golden_large4097():
    import numpy as np
    return np.load(os.path.join(GOLDEN, "large_4097.npz"))
