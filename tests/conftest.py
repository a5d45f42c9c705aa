import sys
from pathlib import Path

import pytest

REPO = Path(__file__).resolve().parents[1]
if str(REPO) not in sys.path:
    sys.path.insert(0, str(REPO))

GOLDEN = REPO / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a HIP device (run on the MI355X box with -m gpu)")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def hip_lib():
    """The built C-ABI library (built on demand; hipcc cross-compiles without a GPU)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("_parrot_build", REPO / "lit-parrot_amd" / "_build.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.build()
    from lit_parrot_amd import _hip

    return _hip.load()
