import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """A fresh checkout has no built libraries (they are git-ignored): build them once, exactly as
    __graft_entry__.build() does, so that `pytest tests -m "not gpu"` is self-sufficient wherever hipcc / gcc exist."""
    lib = os.path.join(ROOT, "dream_amd", "libdream_hip.so")
    oracle_lib = os.path.join(ROOT, "oracle", "libpeaks_c.so")
    if os.path.exists(lib) and os.path.exists(oracle_lib):
        return
    import __graft_entry__ as entry
    if not os.path.exists(lib):
        entry.build_hip()
    if not os.path.exists(oracle_lib):
        entry.build_oracle()


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
