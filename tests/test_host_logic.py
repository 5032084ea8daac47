"""CPU-side checks of the boundary: the C ABI library loads and exports every symbol the header
declares, host arithmetic (bounds, envelope header), argument validation of the host mirror, the
byte-balanced multi-GPU split, and loud failure without a device.  No compute calls."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import k4os.compression.lz4_amd as pkg
from k4os.compression.lz4_amd import LZ4Codec, LZ4Level, LZ4Pickler, _native, corpus
from k4os.compression.lz4_amd.sharding import byte_balanced_ranges

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from k4os.compression.lz4_amd import build
    build.build_native()
    return _native.load_library()


def test_library_exports_every_declared_symbol(lib):
    header = open(os.path.join(ROOT, "include", "k4lz4.h")).read()
    declared = set(re.findall(r"K4LZ4_API\s+[\w\s\*]+?\b(k4lz4_\w+)\s*\(", header))
    assert len(declared) >= 20
    assert declared == set(_native.SYMBOLS), declared ^ set(_native.SYMBOLS)
    raw = C.CDLL(_native.LIB_PATH)
    for name in declared:
        assert getattr(raw, name) is not None
    assert lib.k4lz4_version() == 100


def test_compress_bound_matches_reference_formula(lib, oracle):
    # LL.tools.cs:38-40
    for n in (0, 1, 254, 255, 256, 4096, 65536, 0x7E000000, 0x7E000001, -1):
        assert lib.k4lz4_compress_bound(n) == oracle.compress_bound(n)
    assert LZ4Codec.MaximumOutputSize(65536) == 65809 and LZ4Codec.MaximumOutputSize(4096) == 4128


def test_level_enum_values():
    # LZ4Level.cs:6-39
    assert [int(l) for l in LZ4Level] == [0, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12]


def test_unpickle_size_matches_oracle_header(lib, oracle):
    rng = np.random.default_rng(1)
    cases = [oracle.pickle(corpus.lorem(n)) for n in (1, 10, 300, 70000)] + [oracle.pickle(corpus.random_bytes(500, 1))]
    for _ in range(300):
        n = int(rng.integers(1, 8))
        cases.append(rng.integers(0, 256, n, dtype=np.uint8).tobytes())
    assert lib.k4lz4_unpickle_size(None, 0) == 0
    for p in cases:
        a = np.frombuffer(p, np.uint8)
        rc, off, rl, comp = oracle.unpickle_header(p)
        want = rl if (rc == 0 and rl >= 0) else -1
        assert lib.k4lz4_unpickle_size(a.ctypes.data, a.size) == want
    assert lib.k4lz4_pickle_bound(0) == 0 and lib.k4lz4_pickle_bound(100) == 105


def test_argument_validation_mirrors_reference():
    # Internal/Extensions.cs:37-52: null -> ArgumentNullException, bad ranges -> ArgumentException
    with pytest.raises(TypeError):
        LZ4Codec.Encode(None, bytearray(10))
    with pytest.raises(TypeError):
        LZ4Codec.Decode(b"abc", None)
    with pytest.raises(ValueError):
        LZ4Codec.Encode(b"abcdef", 2, 10, bytearray(10), 0, 10)
    with pytest.raises(ValueError):
        LZ4Codec.Decode(b"abcdef", 0, 6, bytearray(10), 5, 6)
    with pytest.raises(ValueError):
        LZ4Codec.Encode(b"abcdef", bytes(10))   # read-only target
    # empty source short-circuits before touching the device (LZ4Codec.cs:45-46,:108-109)
    assert LZ4Codec.Encode(b"", bytearray(10)) == 0
    assert LZ4Codec.Decode(b"", bytearray(10)) == 0
    with pytest.raises(TypeError):
        LZ4Pickler.Pickle(b"abc", None)


def test_no_device_fails_loudly(lib):
    if lib.k4lz4_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(_native.NativeLibraryError):
        _native.Context()
    with pytest.raises(_native.NativeLibraryError):
        LZ4Codec.Encode(b"some bytes to compress, some bytes", bytearray(100))
    with pytest.raises(_native.NativeLibraryError):
        LZ4Pickler.Pickle(b"hello hello hello hello")


def test_product_never_imports_oracle():
    pkg_dir = os.path.join(ROOT, "k4os")
    for dp, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                text = open(os.path.join(dp, f)).read()
                assert "oracle_lib" not in text and "k4o_" not in text and "libk4lz4_oracle" not in text, f


@pytest.mark.parametrize("world", [1, 2, 3, 4, 8])
def test_byte_balanced_ranges(world):
    rng = np.random.default_rng(world)
    lens = np.exp(rng.uniform(np.log(1024), np.log(4 << 20), 5000)).astype(np.int64)
    r = byte_balanced_ranges(lens, world)
    assert r[0][0] == 0 and r[-1][1] == lens.size
    for (a, b), (c, d) in zip(r[:-1], r[1:]):
        assert b == c and a <= b
    per = np.array([lens[a:b].sum() for a, b in r], dtype=np.float64)
    assert per.max() <= lens.sum() / world + lens.max()
    assert per.min() >= lens.sum() / world - lens.max()
    # degenerate inputs
    assert byte_balanced_ranges([], 4) == [(0, 0)] * 4
    assert byte_balanced_ranges([5], 3)[-1][1] == 1
    eq = byte_balanced_ranges([65536] * 4096, 8)
    assert all(b - a == 512 for a, b in eq)


def test_enforce32_switch_round_trips():
    from k4os.compression.lz4_amd import LZ4Codec
    assert LZ4Codec.Enforce32 is False
    LZ4Codec.Enforce32 = True
    assert LZ4Codec.Enforce32 is True
    LZ4Codec.Enforce32 = False
    assert LZ4Codec.Enforce32 is False


def test_one_hip_runtime_whatever_the_import_order():
    """libk4lz4.so and torch must bind the same libamdhip64: with two copies mapped only the first sees the GPU."""
    import subprocess, sys
    code = (
        "from k4os.compression.lz4_amd import _native; _native.load_library(); import torch;"
        "print(len({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l}))"
    )
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=str(__import__('pathlib').Path(__file__).resolve().parents[1]))
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().splitlines()[-1] == "1"


def test_recommended_min_batch_is_the_documented_crossover():
    """INTEGRATION.md "Crossover": floor of a device call x what the host sustains / block size; no device needed"""
    from k4os.compression.lz4_amd import LZ4Codec
    assert LZ4Codec.RecommendedMinBatch(0, 65536, 32.0) == 1049          # 2.0 ms x 32 GiB/s / 64 KiB
    assert LZ4Codec.RecommendedMinBatch(1, 65536, 35.0) == 316           # 0.55 ms x 35 GiB/s / 64 KiB
    assert LZ4Codec.RecommendedMinBatch(0, 65536, 0.8) == 27             # one host thread
    assert LZ4Codec.RecommendedMinBatch(0, 65536) == LZ4Codec.RecommendedMinBatch(0, 65536, 32.0)
    assert LZ4Codec.RecommendedMinBatch(0, 4096, 32.0) == LZ4Codec.RecommendedMinBatch(0, 65536, 32.0)   # the floor scales with the block
    assert LZ4Codec.RecommendedMinBatch(2, 65536, 2.2) > 100
    with pytest.raises(ValueError):
        LZ4Codec.RecommendedMinBatch(3, 65536)
