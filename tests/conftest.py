import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    import oracle_lib
    return oracle_lib.load_oracle()


@pytest.fixture(scope="session")
def ref():
    """The reference's own dbscan.cpp / dbscan1d.cpp / kc.cpp (oracle/_ref); skipped when not built."""
    import oracle_lib
    lib = oracle_lib.load_ref()
    if lib is None:
        pytest.skip("oracle/_ref/libcsvref.so not built (needs /root/reference at build time)")
    return lib


@pytest.fixture(scope="session")
def ctx():
    """A GPU context; GPU tests must fail (not skip) when the HIP library or the device is missing."""
    import contextsv_amd as cs
    c = cs.Context(0)
    yield c
    c.close()
