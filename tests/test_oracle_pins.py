"""Pins the CPU oracle (oracle/k4lz4_oracle.c) before anything is compared against it.

Sources of truth:
  * assets/issue64/input.dat -> output.dat of the reference repo (copied to tests/golden/),
    decoded exactly as src/K4os.Compression.LZ4.Tests/Issue64.cs:16-55 does;
  * probe values recorded at survey time from liblz4 1.9.3 == line-by-line LL64 restatement;
  * byte equality with the system liblz4.so.1 on the reference's reproducible fixtures
    (BlockRoundtripTests.cs:45-98: quick fox, repeated bytes, Lorem lengths).
"""
import os
import struct

import numpy as np
import pytest

from k4os.compression.lz4_amd import corpus

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")


def _issue64_records():
    raw = open(os.path.join(GOLDEN, "issue64_input.bin"), "rb").read()
    pos = 20
    recs = []
    while raw[pos:pos + 4] == b"bv41":
        u, c = struct.unpack_from("<II", raw, pos + 4)
        recs.append((u, raw[pos + 12:pos + 12 + c]))
        pos += 12 + c
    assert raw[pos:pos + 4] == b"bv4$"
    return recs


def test_issue64_golden_decode(oracle):
    want = open(os.path.join(GOLDEN, "issue64_output.bin"), "rb").read()
    recs = _issue64_records()
    assert recs[0][0] == 65536 and len(recs[0][1]) == 14505
    out = bytearray()
    prev = np.zeros(0, np.uint8)
    for u, payload in recs:
        src = np.frombuffer(payload, np.uint8)
        if prev.size == 0:
            n, dst = oracle.decompress_safe(src, u)
        else:
            n, dst = oracle.decompress_using_dict(src, u, prev)
        assert n == u
        prev = dst[:n].copy()
        out += prev.tobytes()
    assert bytes(out) == want[:len(out)]
    assert oracle.adler32(recs[0][1]) == 0x1228b5d5
    assert oracle.adler32(want[:65536]) == 0x5848d976


PROBES_FAST = [(1000, 428, 0x5b689c08), (4096, 440, 0x2f5aa7cc), (65536, 681, 0x0d0097ed),
               (0x172a5, 812, None), (0x123456, 5118, 0x208fd79e)]


@pytest.mark.parametrize("n,size,adler", PROBES_FAST)
def test_survey_probe_values_fast(oracle, n, size, adler):
    ret, dst = oracle.compress_fast(corpus.lorem(n))
    assert ret == size
    if adler is not None:
        assert oracle.adler32(dst[:ret]) == adler


def test_survey_probe_small(oracle):
    ret, dst = oracle.compress_fast(np.frombuffer(corpus.QUICK_FOX, np.uint8))
    assert ret == 45 and oracle.adler32(dst[:ret]) == 0x8ade10e6
    ret, dst = oracle.compress_fast(corpus.repeated(0, 1000))
    assert dst[:ret].tobytes() == bytes.fromhex("1f000100ffffffd2500000000000")
    ret, dst = oracle.compress_fast(corpus.repeated(0xAA, 65536))
    assert ret == 267 and dst[:6].tobytes() == bytes.fromhex("1faa0100ffff")


def _fixtures():
    yield "fox", np.frombuffer(corpus.QUICK_FOX, np.uint8)
    for n in (1, 12, 13, 14, 64, 1000, 4096, 0x7FFF, 0xFFFF, 65536, 65546, 65547, 0x172a5, 300000):
        yield f"lorem{n}", corpus.lorem(n)
    for n in (1, 13, 15, 17, 33, 67, 1000, 65536):
        yield f"rep{n}", corpus.repeated(0xAA, n)
    for n in (100, 5000, 70000, 200000):
        yield f"rand{n}", corpus.random_bytes(n, n)
    for i, name in enumerate(corpus.SILESIA_NAMES):
        yield f"cls-{name}", corpus.class_bytes(name, 65536 if i % 2 else 150000, 7)


@pytest.mark.parametrize("name,data", list(_fixtures()), ids=[n for n, _ in _fixtures()])
def test_fast_encode_equals_liblz4_and_roundtrips(oracle, syslz4, name, data):
    bound = oracle.compress_bound(data.size)
    ret, dst = oracle.compress_fast(data)
    ret2, dst2 = syslz4.compress_fast(data, bound)
    assert ret == ret2 and dst[:ret].tobytes() == dst2[:ret2].tobytes()
    assert (dst[ret:] == 0xCD).all()
    # decode: exact capacity and oversize capacity (BlockRoundtripTests.cs:45-61)
    for cap in (data.size, 2 * data.size + 64):
        n, out = oracle.decompress_safe(dst[:ret], cap)
        assert n == data.size and out[:n].tobytes() == data.tobytes()
        assert (out[n:] == 0xCD).all()
        n2, out2 = syslz4.decompress_safe(dst[:ret], cap)
        assert n2 == n
    # limitedOutput: cap == size succeeds with identical bytes, cap == size-1 fails
    r3, d3 = oracle.compress_fast(data, cap=ret)
    assert r3 == ret and d3[:ret].tobytes() == dst[:ret].tobytes()
    if ret > 1:
        r4, _ = oracle.compress_fast(data, cap=ret - 1)
        assert r4 == 0


def test_decode_malformed_matches_reference_rules(oracle, syslz4):
    """Mutated streams: the oracle must never write outside dst and must agree with liblz4 on
    accept/reject except for the one documented divergence class."""
    rng = np.random.default_rng(5)
    data = corpus.class_bytes("dickens", 3000, 3)
    ret, dst = oracle.compress_fast(data)
    good = dst[:ret].copy()
    diverge = 0
    for t in range(1500):
        bad = good.copy()
        kind = t % 3
        if kind == 0:
            bad = bad[:rng.integers(1, ret)]
        elif kind == 1:
            for _ in range(int(rng.integers(1, 4))):
                bad[rng.integers(0, ret)] = rng.integers(0, 256)
        else:
            bad = np.concatenate([bad, rng.integers(0, 256, size=int(rng.integers(1, 9)), dtype=np.uint8)])
        cap = data.size + int(rng.integers(-20, 21))
        n, out = oracle.decompress_safe(bad, cap)
        n2, out2 = syslz4.decompress_safe(bad, cap)
        if (n < 0) != (n2 < 0):
            diverge += 1
            continue
        if n >= 0:
            assert n == n2 and out[:n].tobytes() == out2[:n].tobytes()
    assert diverge <= 5


def _hc_fixtures():
    yield "fox", np.frombuffer(corpus.QUICK_FOX, np.uint8)
    for n in (1, 12, 13, 14, 64, 1000, 4096, 65536, 0x123456):
        yield f"lorem{n}", corpus.lorem(n)
    for n in (13, 33, 1000, 65536):
        yield f"rep{n}", corpus.repeated(0xAA, n)
    yield "rand5000", corpus.random_bytes(5000, 5)
    yield "pattern", np.tile(np.frombuffer(b"abcdabcdabcdabcd" * 4 + b"xyz", np.uint8), 700)
    yield "zeros+noise", np.concatenate([np.zeros(30000, np.uint8), corpus.random_bytes(100, 1), np.zeros(30000, np.uint8)])
    for i, name in enumerate(corpus.SILESIA_NAMES):
        yield f"cls-{name}", corpus.class_bytes(name, 65536 if i % 2 else 100000, 7)


PROBES_HC3 = [(1000, 425, 0x4a7d9a5a), (4096, 437, None), (65536, 678, 0x5301963f), (0x123456, 5099, None)]


@pytest.mark.parametrize("n,size,adler", PROBES_HC3)
def test_survey_probe_values_hc3(oracle, n, size, adler):
    ret, dst = oracle.compress_hc(corpus.lorem(n), 3)
    assert ret == size
    if adler is not None:
        assert oracle.adler32(dst[:ret]) == adler


@pytest.mark.parametrize("level", [3, 4, 6, 8, 9, 10, 11, 12])
@pytest.mark.parametrize("name,data", list(_hc_fixtures()), ids=[n for n, _ in _hc_fixtures()])
def test_hc_encode_equals_liblz4_and_roundtrips(oracle, syslz4, name, data, level):
    """LZ4_compress_HC levels 3..9 (hash chain; level 9 adds pattern analysis) and 10..12 (optimal parser with
    chain swap): the restatement of LL64.high.cs against liblz4 1.9.3 -- the reference's own HC goldens need the
    Silesia corpus."""
    if level >= 10 and data.size > 200000:
        data = data[:200000]
    bound = oracle.compress_bound(data.size)
    ret, dst = oracle.compress_hc(data, level)
    ret2, dst2 = syslz4.compress_hc(data, bound, level)
    assert ret == ret2 and dst[:ret].tobytes() == dst2[:ret2].tobytes()
    assert (dst[ret:] == 0xCD).all()
    n, out = oracle.decompress_safe(dst[:ret], data.size)
    assert n == data.size and out[:n].tobytes() == data.tobytes()
    r3, d3 = oracle.compress_hc(data, level, cap=ret)
    assert r3 == ret and d3[:ret].tobytes() == dst[:ret].tobytes()
    if ret > 1:
        assert oracle.compress_hc(data, level, cap=ret - 1)[0] == 0


# ---- the data the GPU path is graded on -------------------------------------------------
# The fixtures above are a few dozen inputs; the bench batch, the configs[3] messages and the configs[4] blocks are what
# the kernels are compared with the oracle on, so the oracle itself is compared with liblz4 on exactly those bytes.

def _pool_map(fn, items):
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=os.cpu_count() or 4) as ex:      # ctypes releases the GIL
        return list(ex.map(fn, items))


def test_fast_oracle_equals_liblz4_on_every_block_of_the_bench_batch(oracle, syslz4):
    """all 4096 blocks of bench.py's batch (BASELINE.json configs[1], seed 2 = rank 0's), L00_FAST"""
    blocks = corpus.silesia_like_blocks(4096, 65536, seed=2)
    bound = oracle.compress_bound(65536)

    def one(i):
        r, d = oracle.compress_fast(blocks[i])
        r2, d2 = syslz4.compress_fast(blocks[i], bound)
        return r == r2 and d[:r].tobytes() == d2[:r2].tobytes()
    bad = [i for i, ok in enumerate(_pool_map(one, range(blocks.shape[0]))) if not ok]
    assert not bad, bad[:10]


def test_hc3_oracle_equals_liblz4_on_the_unique_blocks_of_the_bench_batch(oracle, syslz4):
    """configs[4]: L03_HC of the same data -- the first 384 blocks hold every class's unique material"""
    blocks = corpus.silesia_like_blocks(4096, 65536, seed=2)[:384]
    bound = oracle.compress_bound(65536)

    def one(i):
        r, d = oracle.compress_hc(blocks[i], 3)
        r2, d2 = syslz4.compress_hc(blocks[i], bound, 3)
        return r == r2 and d[:r].tobytes() == d2[:r2].tobytes()
    bad = [i for i, ok in enumerate(_pool_map(one, range(blocks.shape[0]))) if not ok]
    assert not bad, bad[:10]


def test_fast_oracle_equals_liblz4_on_configs3_messages(oracle, syslz4):
    """200 messages of the configs[3] batch (1 KiB .. 4 MiB, random / text alternating), the 20 longest of the first
    20 000 among them: everything from 64 KiB + 11 bytes on takes the byU32 table and hash5 (LL64.fast.cs:526-544)"""
    lens = corpus.config4_lengths()
    first = lens[:20000]
    pick = sorted(set(range(0, 20000, 111)) | set(int(i) for i in np.argsort(first)[-20:]))
    assert len(pick) >= 200 and int(first[pick].max()) > (3 << 20)

    def one(i):
        data, off, ln = corpus.config4_share(lens, i, i + 1)
        msg = data[:int(ln[0])]
        r, d = oracle.compress_fast(msg)
        r2, d2 = syslz4.compress_fast(msg, oracle.compress_bound(msg.size))
        return r == r2 and d[:r].tobytes() == d2[:r2].tobytes()
    bad = [i for i, ok in zip(pick, _pool_map(one, pick)) if not ok]
    assert not bad, bad[:10]


@pytest.mark.parametrize("level", [9, 10, 12])
def test_hc_high_levels_oracle_equals_liblz4_on_bench_blocks(oracle, syslz4, level):
    """216 blocks of the bench batch (18 of every class) at the levels the reference holds goldens for besides 3
    (ChecksumBlockTests.cs:137-172): pattern analysis (9) and the optimal parser (10, 12)"""
    blocks = corpus.silesia_like_blocks(4096, 65536, seed=2)[:216]
    bound = oracle.compress_bound(65536)

    def one(i):
        r, d = oracle.compress_hc(blocks[i], level)
        r2, d2 = syslz4.compress_hc(blocks[i], bound, level)
        return r == r2 and d[:r].tobytes() == d2[:r2].tobytes()
    bad = [i for i, ok in enumerate(_pool_map(one, range(blocks.shape[0]))) if not ok]
    assert not bad, bad[:10]
