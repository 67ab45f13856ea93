import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import __graft_entry__ as entry  # noqa: E402


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long-running CPU check")


def pytest_collection_modifyitems(config, items):
    """Collection order under `pytest -x`: the GPU parity tests proper first (every single-GPU BASELINE config is exercised by
    tests/test_gpu_parity.py), the other parity files in name order, the bench.py harness contract last."""
    def rank(item):
        name = os.path.basename(str(item.fspath))
        if name == "test_gpu_parity.py":
            return 0
        if name.startswith("test_zz_"):
            return 2
        return 1
    items.sort(key=rank)  # stable: the order inside a class of files is unchanged


@pytest.fixture(scope="session")
def pkg():
    return entry.load_package()


@pytest.fixture(scope="session")
def synth(pkg):
    import importlib
    return importlib.import_module("amos_slam_amd.synth")


@pytest.fixture(scope="session")
def ob():
    import oracle_binding
    oracle_binding.lib()
    return oracle_binding


@pytest.fixture(scope="session")
def gpu_lib(pkg):
    """The HIP library on a box with a GPU; GPU tests fail (not skip) if it cannot be used."""
    if pkg.device_count() < 1:
        pytest.fail("no HIP device visible")
    return pkg
