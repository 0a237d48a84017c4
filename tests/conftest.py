import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as graft  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def pkg():
    return graft.load_package()


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.load()
    return oracle


_scene_cache = {}


@pytest.fixture(scope="session")
def scene_factory(pkg):
    def get(name, bands=None):
        key = (name, bands)
        if key not in _scene_cache:
            _scene_cache[key] = pkg.scenes.by_name(name, bands)
        return _scene_cache[key]
    return get
