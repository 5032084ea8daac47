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
