import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run by the driver with -m gpu)")


def _gpu_available() -> bool:
    try:
        import torch

        return bool(torch.cuda.is_available())
    except Exception:
        return False


@pytest.fixture(scope="session")
def gpu():
    if not _gpu_available():
        pytest.skip("no GPU visible")
    return 0


@pytest.fixture(scope="session", autouse=True)
def _native_libraries_built():
    """A fresh checkout has no .so files (they are git-ignored): build them once (hipcc cross-compiles without a GPU,
    ~30 s; gcc for the oracle).  On the GPU box the prebuilt in-tree libraries travel with the snapshot."""
    from parcels_amd import _hip

    if not os.path.exists(_hip.LIB_PATH):
        _hip.build_library()
    from oracle import c_oracle

    c_oracle.build()
