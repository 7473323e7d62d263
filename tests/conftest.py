import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def _ensure_built():
    """The native library and the oracle are git-ignored build products: build them when a fresh
    checkout runs the suite (hipcc cross-compiles gfx950 without a GPU).  A failed build is not
    hidden: the imports below then fail loudly."""
    lib = os.path.join(ROOT, "uzkge_amd", "libuzkge_gpu.so")
    orc = os.path.join(ROOT, "oracle", "liboracle_bn254.so")
    if not (os.path.exists(lib) and os.path.exists(orc)):
        import __graft_entry__
        __graft_entry__.build()


_ensure_built()


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu() -> bool:
    try:
        from uzkge_amd import backend
        return backend.device_count() > 0
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session")
def gpu():
    """Initialised backend (device 0)."""
    from uzkge_amd import backend
    backend.init(0)
    yield backend
