"""pytest configuration: markers, paths, shared fixtures.

`-m "not gpu"` : oracle vs golden vectors, host logic, ABI symbol checks, lane emulator (CPU only).
`-m gpu`       : parity tests proper; they call the HIP path through the C ABI on cuda:0.
"""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "aac.js_amd", "python"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """A GPU test that has not returned after ten minutes never will (the whole suite takes one): fail it with the stacks on
    stderr (pytest-timeout, where installed) instead of sitting there until whoever started the run gives up."""
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("gpu") and not item.get_closest_marker("timeout"):
            item.add_marker(pytest.mark.timeout(600))


@pytest.fixture(scope="session")
def golden():
    from golden_io import load_golden
    return load_golden()


@pytest.fixture(scope="session")
def oracle():
    """ctypes handle on oracle/liboracle.so, built on demand (gcc only, no GPU)."""
    import orc
    return orc.load()


@pytest.fixture(scope="session")
def engine_lib():
    """The product C-ABI library; GPU tests must fail loudly when it is missing."""
    import aacgpu
    return aacgpu.load_library()
