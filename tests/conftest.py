import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
sys.path.insert(0, os.path.join(REPO, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` through gpurun)")


@pytest.fixture(scope="session")
def built():
    """Both libraries + the oracle are built once per session (hipcc cross-compiles gfx950 without a GPU)."""
    import llama_box_amd as L
    import harness

    if not (os.path.exists(L.BACKEND_SO) and os.path.exists(L.HOST_SO)):
        L.build()
    if not os.path.exists(harness.ORACLE_SO):
        harness.build_oracle()
    return L


@pytest.fixture(scope="session")
def H(built):
    return built.host()


@pytest.fixture(scope="session")
def backend(built):
    """The MI355X backend through its C-ABI. Fails loudly (never skips, never falls back) when it cannot load."""
    be = built.Backend(0)
    yield be
    be.close()


_LOG = os.path.join(REPO, "gpurun_out", "parity_log.txt")


@pytest.fixture(scope="session")
def plog():
    os.makedirs(os.path.dirname(_LOG), exist_ok=True)
    f = open(_LOG, "a")

    def log(msg):
        f.write(msg + "\n")
        f.flush()

    yield log
    f.close()
