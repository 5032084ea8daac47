"""The reference's own known-answer rows for the block encoders (Tests/ChecksumBlockTests.cs:13-172, copied to
tests/golden/checksum_block_rows.json by tests/tools/extract_checksum_rows.py): whole Silesia files as one block, levels
0 / 3 / 9 / 10 / 12, for the 64-bit engine and -- architecture 4 -- the 32-bit one (LZ4Codec.Enforce32).  They were
generated from native lz4 1.9.2 builds (playground/SharedSources/app.cpp:79-141).

The corpus is not in the repository or the image: these tests run only when K4LZ4_CORPUS_DIR points at a directory
holding the Silesia files (dickens, mozilla, mr, ...), and are skipped otherwise.  With a corpus they pin
  * the oracle (all rows; the optimal-parser rows only when K4LZ4_GOLDEN_SLOW=1, they take minutes on a CPU thread),
  * the GPU path (rows of levels 0 and 3 at both architectures; marked gpu)."""
import base64
import json
import os
import zlib

import numpy as np
import pytest

GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "checksum_block_rows.json")
ROWS = json.load(open(GOLDEN))["rows"]
CORPUS = os.environ.get("K4LZ4_CORPUS_DIR")


def _load(row):
    if not CORPUS:
        pytest.skip("K4LZ4_CORPUS_DIR not set: the Silesia corpus is not available")
    path = os.path.join(CORPUS, row["file"])
    if not os.path.exists(path):
        pytest.skip(f"{path} not found")
    with open(path, "rb") as f:
        f.seek(row["index"])
        data = f.read(row["length"])
    assert len(data) == row["length"]
    return np.frombuffer(data, np.uint8)


def _check(row, comp: bytes):
    head = base64.b64decode(row["first_bytes_base64"])
    assert comp[:len(head)] == head
    assert len(comp) == row["compressed_length"]
    assert zlib.adler32(comp) == row["adler32"]          # TestHelpers/Tools.cs:14-44 is the standard Adler-32


def _id(r):
    return f"a{r['architecture']}-{r['file']}-L{r['level']}"


def test_rows_are_complete():
    """12 Silesia files x (fast + levels 3, 9, 10, 12) x two architectures"""
    assert len(ROWS) == 120
    assert {r["architecture"] for r in ROWS} == {4, 8} and {r["level"] for r in ROWS} == {0, 3, 9, 10, 12}
    for f in {r["file"] for r in ROWS}:
        hc32 = {(r["level"], r["compressed_length"], r["adler32"]) for r in ROWS if r["file"] == f and r["architecture"] == 4 and r["level"] >= 3}
        hc64 = {(r["level"], r["compressed_length"], r["adler32"]) for r in ROWS if r["file"] == f and r["architecture"] == 8 and r["level"] >= 3}
        assert hc32 == hc64          # the HC encoders do not depend on the architecture; the fast encoder does
    fast32 = {r["file"]: r["compressed_length"] for r in ROWS if r["architecture"] == 4 and r["level"] == 0}
    fast64 = {r["file"]: r["compressed_length"] for r in ROWS if r["architecture"] == 8 and r["level"] == 0}
    assert any(fast32[f] != fast64[f] for f in fast32)


@pytest.mark.parametrize("row", ROWS, ids=[_id(r) for r in ROWS])
def test_oracle_reproduces_reference_goldens(oracle, row):
    if row["level"] >= 10 and os.environ.get("K4LZ4_GOLDEN_SLOW") != "1":
        pytest.skip("optimal-parser rows take minutes per file on one CPU thread: set K4LZ4_GOLDEN_SLOW=1")
    data = _load(row)
    if row["level"] == 0:
        n, dst = oracle.compress_fast_x32(data) if row["architecture"] == 4 else oracle.compress_fast(data)
    else:
        n, dst = oracle.compress_hc(data, row["level"])
    assert n > 0
    _check(row, dst[:n].tobytes())


@pytest.mark.gpu
@pytest.mark.parametrize("row", [r for r in ROWS if r["level"] in (0, 3)], ids=[_id(r) for r in ROWS if r["level"] in (0, 3)])
def test_gpu_reproduces_reference_goldens(row):
    from k4os.compression.lz4_amd import LZ4Codec, LZ4Level
    data = _load(row)
    target = np.zeros(LZ4Codec.MaximumOutputSize(data.size), np.uint8)
    try:
        LZ4Codec.Enforce32 = row["architecture"] == 4
        n = LZ4Codec.Encode(data, target, LZ4Level(row["level"]))
    finally:
        LZ4Codec.Enforce32 = False
    assert n > 0
    _check(row, target[:n].tobytes())
    out = np.zeros(data.size, np.uint8)
    assert LZ4Codec.Decode(target[:n].copy(), out) == data.size and out.tobytes() == data.tobytes()
