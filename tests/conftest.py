import os
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
for p in (HERE, ROOT):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(HERE, "golden")
REF_DATA = os.path.join(GOLDEN, "ref_data")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "spawns: starts GPU-using child processes; must run before this "
                                       "process initialises the GPU itself")


def pytest_collection_modifyitems(config, items):
    # Child processes may only be started (fork + exec) while this process has not yet
    # initialised HIP, so tests that spawn GPU workers go first.
    def spawns(item):
        return item.get_closest_marker("spawns") is not None or "test_host_cli" in item.nodeid

    items.sort(key=lambda it: 0 if spawns(it) else 1)


@pytest.fixture(scope="session")
def oracle_lib():
    import oracle_py

    oracle_py.build()
    return oracle_py.lib()
