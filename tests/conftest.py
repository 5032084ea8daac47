"""pytest configuration: registers the `gpu` marker and shared fixtures.

`-m "not gpu"` : oracle pins, host logic, C-ABI symbol checks, kernel logic under the CPU
                 wave emulator (tests/emu) -- runs without a GPU.
`-m gpu`       : parity tests proper; they call the HIP path through the C-ABI (libk4lz4.so).
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """`-m gpu` tests need the device: on a box without one they are skipped, not errored (a plain `pytest` run on a CPU
    box stays green); the product itself still fails loudly without a device (tests/test_host_logic.py)."""
    gpu_items = [it for it in items if it.get_closest_marker("gpu")]
    if not gpu_items:
        return
    try:
        from k4os.compression.lz4_amd import _native
        have = _native.load_library().k4lz4_device_count() > 0
    except Exception:
        have = False
    if not have:
        skip = pytest.mark.skip(reason="no gfx950 device visible (k4lz4_device_count() == 0)")
        for it in gpu_items:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    from oracle_lib import Oracle
    return Oracle()


@pytest.fixture(scope="session")
def syslz4():
    from oracle_lib import SystemLZ4
    s = SystemLZ4()
    if not s.available:
        pytest.skip("liblz4.so.1 not present")
    return s
