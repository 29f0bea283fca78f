import os
import sys

import pytest

# PyTorch ships its own copy of the HIP runtime. When a test shares a process between torch and
# librpw_hip.so (device-resident batches), torch must be loaded FIRST so that both resolve to one
# runtime; the other order leaves torch with "No HIP GPUs are available".
try:
    import torch  # noqa: F401
except Exception:  # pragma: no cover
    torch = None

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu() -> bool:
    # counting devices does not initialise the GPU runtime on this image
    try:
        import torch
        return torch.cuda.device_count() > 0
    except Exception:
        return os.path.exists("/dev/kfd")


@pytest.fixture(scope="session")
def has_gpu():
    return _has_gpu()


@pytest.fixture(scope="session")
def engine_factory():
    """Creates rpw engines; the HIP library must exist -- no fallback."""
    import rpw_amd
    if not os.path.exists(rpw_amd.LIB_PATH):          # a snapshot without built artefacts: compile in-tree (hipcc), then load
        rpw_amd.build()

    made = []

    def make(config=None, **kw):
        e = rpw_amd.Engine(config or rpw_amd.PatchworkConfig(), **kw)
        made.append(e)
        return e

    yield make
    for e in made:
        e.close()
