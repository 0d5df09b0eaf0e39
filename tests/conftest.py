import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure). Built on demand with g++."""
    from oracle import oracle_py
    oracle_py.lib()
    return oracle_py


@pytest.fixture(scope="session")
def engine_mod():
    """The engine binding; building is hipcc-only so it also works without a GPU."""
    import __graft_entry__ as ge
    ge.build_engine_only()
    from alphazero_rs_amd import engine
    return engine


@pytest.fixture(scope="session")
def engine(engine_mod):
    """One az_engine on cuda:0 for the whole GPU session, with the stub, hash and conv nets registered."""
    e = engine_mod.Engine(device=0, max_batch=8192, net_channels=512)
    e.net_set_kind(0, engine_mod.NET_STUB, 0)
    e.net_set_kind(10, engine_mod.NET_HASH, 1234)
    e.net_set_kind(11, engine_mod.NET_HASH, 1234)
    yield e
    e.close()
