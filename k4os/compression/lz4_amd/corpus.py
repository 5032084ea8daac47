"""Deterministic synthetic block inputs for tests and bench.py.

The reference's own fixtures:
  * ``lorem(n)``      -- TestHelpers/Lorem.cs:9-16 text tiled to n bytes
                         (BlockRoundtripTests.cs:86-98, PicklingTests.cs:11-50)
  * ``repeated(b,n)`` -- BlockRoundtripTests.cs:70-84
  * Silesia corpus    -- not available offline; ``silesia_like_blocks`` builds the 12-class
                         "Silesia-like" mix SURVEY.md section 8d / BASELINE.md config 2 describe.
                         Set K4LZ4_CORPUS_DIR to a directory holding the real corpus files to
                         use them instead.

Everything is numpy-vectorised (a 256 MiB batch is generated in seconds) and depends only on
the integer seed, so the oracle and the GPU path always see the same bytes.
"""
from __future__ import annotations

import os
from typing import List, Sequence, Tuple

import numpy as np

LOREM_TEXT = (
    "Lorem ipsum dolor sit amet, consectetur adipiscing elit, "
    "sed do eiusmod tempor incididunt ut labore et dolore magna aliqua. "
    "Ut enim ad minim veniam, quis nostrud exercitation ullamco laboris "
    "nisi ut aliquip ex ea commodo consequat. Duis aute irure dolor in reprehenderit "
    "in voluptate velit esse cillum dolore eu fugiat nulla pariatur. "
    "Excepteur sint occaecat cupidatat non proident, sunt in culpa qui officia "
    "deserunt mollit anim id est laborum. "
).encode("utf-8")

QUICK_FOX = b"The quick brown fox jumps over the lazy dog"

SILESIA_NAMES = (
    "dickens", "mozilla", "mr", "nci", "ooffice", "osdb",
    "reymont", "samba", "sao", "webster", "xml", "x-ray",
)  # TestHelpers/Tools.cs:105-109


def lorem(n: int) -> np.ndarray:
    base = np.frombuffer(LOREM_TEXT, dtype=np.uint8)
    reps = -(-max(n, 1) // base.size)
    return np.tile(base, reps)[:n].copy()


def repeated(value: int, n: int) -> np.ndarray:
    return np.full(n, value & 0xFF, dtype=np.uint8)


def random_bytes(n: int, seed: int) -> np.ndarray:
    return np.random.default_rng(seed).integers(0, 256, size=n, dtype=np.uint8)


# ---------------------------------------------------------------------------------------------
# class generators (each returns exactly n bytes)
# ---------------------------------------------------------------------------------------------

def _word_text(n: int, rng: np.random.Generator, vocab_size: int, zipf_a: float,
               alphabet: bytes, noise: float, mean_len: float = 5.0) -> np.ndarray:
    """Zipf-distributed words over a random vocabulary, with `noise` of the words replaced by
    fresh random tokens (keeps the text from being *too* compressible)."""
    alpha = np.frombuffer(alphabet, dtype=np.uint8)
    wlen = np.clip(rng.poisson(mean_len, vocab_size), 1, 14).astype(np.int64)
    vstart = np.concatenate(([0], np.cumsum(wlen + 1)))[:-1]
    vflat = alpha[rng.integers(0, alpha.size, size=int((wlen + 1).sum()))]
    vflat[vstart + wlen] = 0x20  # trailing space
    seps = np.frombuffer(b". , ; \n", dtype=np.uint8)
    nwords = int(n / (mean_len + 1) * 1.4) + 16
    ranks = np.minimum(rng.zipf(zipf_a, nwords) - 1, vocab_size - 1)
    lens = wlen[ranks] + 1
    starts = np.concatenate(([0], np.cumsum(lens)))[:-1]
    total = int(lens.sum())
    word_of = np.repeat(np.arange(nwords), lens)
    within = np.arange(total) - starts[word_of]
    out = vflat[vstart[ranks][word_of] + within]
    if noise > 0:
        m = rng.random(total) < noise
        out[m] = alpha[rng.integers(0, alpha.size, size=int(m.sum()))]
    punct = rng.random(nwords) < 0.08
    out[(starts + lens - 1)[punct]] = seps[rng.integers(0, seps.size, size=int(punct.sum()))]
    if total < n:                       # short draw (small n, long-tailed word lengths): wrap around
        out = np.resize(out, n)
    return out[:n].copy()


_LOWER = b"etaoinshrdlcumwfgypbvkjxqz"
_MIXED = b"etaoinshrdlcumwfgypbvkETAOINSHRDL0123456789_-/"


def _english(n, rng):
    return _word_text(n, rng, 6000, 1.25, _LOWER, 0.045)


def _polish(n, rng):
    return _word_text(n, rng, 12000, 1.35, _LOWER + bytes([0xB1, 0xE6, 0xEA, 0xB3, 0xF3, 0xB6]), 0.02, 6.0)


def _dictionary(n, rng):
    return _word_text(n, rng, 20000, 1.3, _LOWER + b"()[];:", 0.03, 5.5)


def _records(n: int, rng, templates: Sequence[bytes], field_alphabet: bytes, field_len: int,
             vary: float) -> np.ndarray:
    """Markup / record streams: fixed templates with short variable fields."""
    alpha = np.frombuffer(field_alphabet, dtype=np.uint8)
    parts: List[np.ndarray] = []
    size = 0
    pool = [np.frombuffer(t, dtype=np.uint8) for t in templates]
    fields = alpha[rng.integers(0, alpha.size, size=(64, field_len))]
    while size < n:
        t = pool[int(rng.integers(0, len(pool)))]
        if rng.random() < vary:
            f = alpha[rng.integers(0, alpha.size, size=field_len)]
        else:
            f = fields[int(rng.integers(0, 64))]
        parts.append(t)
        parts.append(f)
        size += t.size + f.size
    return np.concatenate(parts)[:n].copy()


def _xml(n, rng):
    t = [b'<entry id="', b'">\n  <title>', b'</title>\n  <author><name>', b'</name></author>\n  <value unit="kg">',
         b'</value>\n</entry>\n', b'  <link rel="alternate" type="text/html" href="http://example.org/item/',
         b'"/>\n  <updated>2003-12-13T18:30:02Z</updated>\n  <summary>']
    return _records(n, rng, t, b"abcdefghijklmnopqrstuvwxyz0123456789 ", 9, 0.55)


def _nci(n, rng):
    t = [b"  0  0  0  0  0  0  0  0  0  0999 V2000\n", b"    0.0000    0.0000    0.0000 C   0  0  0  0  0  0  0  0  0  0  0  0\n",
         b"  1  2  1  0  0  0  0\n", b"M  END\n> <NSC>\n", b"\n$$$$\n", b"   -", b"    0.0000 O   0  0\n"]
    return _records(n, rng, t, b"0123456789.", 6, 0.25)


def _source_tar(n, rng):
    t = [b"static int ", b"(struct connection_struct *conn, const char *", b")\n{\n\tint ret = -1;\n\tif (!",
         b") {\n\t\tDEBUG(3, (\"", b" failed\\n\"));\n\t\treturn ret;\n\t}\n", b"\treturn smb_", b"#include \"includes.h\"\n",
         b"/* ", b" */\n", b";\n\t", b" = talloc_strdup(ctx, "]
    return _records(n, rng, t, b"abcdefghijklmnopqrstuvwxyz_", 7, 0.8)


def _binary_x86(n, rng):
    """x86-ish opcode soup: skewed byte distribution + short repeated idioms + random imm32."""
    ops = np.array([0x8B, 0x89, 0xE8, 0xFF, 0x83, 0x0F, 0x48, 0x4C, 0x85, 0xC3, 0x55, 0x5D, 0x74, 0x75, 0xEB, 0x00],
                   dtype=np.uint8)
    w = np.array([14, 12, 9, 8, 8, 7, 7, 5, 5, 4, 4, 4, 4, 3, 3, 3], dtype=np.float64)
    out = ops[rng.choice(ops.size, size=n, p=w / w.sum())]
    m = rng.random(n) < 0.42
    out[m] = rng.integers(0, 256, size=int(m.sum()), dtype=np.uint8)
    idiom = np.frombuffer(bytes.fromhex("554889e54883ec20897dec488b45f8c9c3"), dtype=np.uint8)
    for pos in rng.integers(0, max(1, n - idiom.size), size=n // 220):
        out[pos:pos + idiom.size] = idiom
    return out


def _mixed_binary(n, rng):
    out = np.empty(n, dtype=np.uint8)
    pos = 0
    kinds = (_binary_x86, _source_tar, _english, lambda k, r: random_bytes(k, int(r.integers(1 << 30))),
             lambda k, r: np.zeros(k, dtype=np.uint8))
    while pos < n:
        k = int(min(n - pos, rng.integers(2048, 24576)))
        out[pos:pos + k] = kinds[int(rng.integers(0, len(kinds)))](k, rng)
        pos += k
    return out


def _db_records(n, rng):
    rec = 128
    rows = -(-n // rec)
    out = np.zeros((rows, rec), dtype=np.uint8)
    ids = np.arange(rows, dtype=np.uint32) + int(rng.integers(1 << 20))
    out[:, 0:4] = ids.view(np.uint8).reshape(rows, 4)
    out[:, 4:12] = rng.integers(0x30, 0x3A, size=(rows, 8), dtype=np.uint8)
    names = rng.integers(0x41, 0x5B, size=(97, 20), dtype=np.uint8)
    out[:, 12:32] = names[rng.integers(0, 97, size=rows)]
    out[:, 32:48] = rng.integers(0, 256, size=(rows, 16), dtype=np.uint8)
    out[:, 48:56] = rng.integers(0, 4, size=(rows, 8), dtype=np.uint8)
    cities = rng.integers(0x61, 0x7B, size=(23, 24), dtype=np.uint8)
    out[:, 56:80] = cities[rng.integers(0, 23, size=rows)]
    out[:, 80:96] = rng.integers(0, 256, size=(rows, 16), dtype=np.uint8)
    out[:, 96:128] = 0x20
    return out.reshape(-1)[:n].copy()


def _float_table(n, rng):
    k = -(-n // 4)
    vals = (rng.standard_normal(k) * 40.0 + 180.0).astype(np.float32)
    return vals.view(np.uint8)[:n].copy()


def _image12(n, rng):
    k = -(-n // 2)
    base = (np.sin(np.arange(k) / 97.0) * 300 + 2048).astype(np.int32)
    noise = rng.integers(-160, 160, size=k)
    return np.clip(base + noise, 0, 4095).astype("<u2").view(np.uint8)[:n].copy()


def _image_smooth(n, rng):
    k = -(-n // 2)
    walk = np.cumsum(rng.integers(-3, 4, size=k))
    vals = np.clip(900 + walk % 700, 0, 4095).astype("<u2")
    bg = rng.random(k // 256 + 1) < 0.35
    vals[np.repeat(bg, 256)[:k]] = 0
    return vals.view(np.uint8)[:n].copy()


_CLASS_GENERATORS = {
    "dickens": _english, "mozilla": _mixed_binary, "mr": _image_smooth, "nci": _nci,
    "ooffice": _binary_x86, "osdb": _db_records, "reymont": _polish, "samba": _source_tar,
    "sao": _float_table, "webster": _dictionary, "xml": _xml, "x-ray": _image12,
}


def class_bytes(name: str, n: int, seed: int) -> np.ndarray:
    """n bytes of the synthetic stand-in for Silesia file `name` (or of the real file when
    K4LZ4_CORPUS_DIR provides it)."""
    corpus_dir = os.environ.get("K4LZ4_CORPUS_DIR")
    if corpus_dir:
        path = os.path.join(corpus_dir, name)
        if os.path.exists(path):
            data = np.fromfile(path, dtype=np.uint8)
            reps = -(-n // max(1, data.size))
            return np.tile(data, reps)[:n].copy()
    rng = np.random.default_rng(seed * 1000 + SILESIA_NAMES.index(name))
    out = _CLASS_GENERATORS[name](n, rng)
    assert out.dtype == np.uint8 and out.size == n
    return out


def silesia_like_blocks(n_blocks: int, block_size: int, seed: int = 2,
                        unique_bytes_per_class: int = 1 << 21) -> np.ndarray:
    """[n_blocks, block_size] uint8: the 12 classes interleaved block by block ("corpus tiled",
    BASELINE.json configs[1]).  Each class has `unique_bytes_per_class` of generated material;
    later tiles of the same material get 8 bytes per block overwritten with the block index so
    no two blocks are identical."""
    ncls = len(SILESIA_NAMES)
    per_class = -(-n_blocks // ncls)
    uniq_blocks = max(1, min(per_class, unique_bytes_per_class // block_size))
    out = np.empty((n_blocks, block_size), dtype=np.uint8)
    for ci, name in enumerate(SILESIA_NAMES):
        base = class_bytes(name, uniq_blocks * block_size, seed).reshape(uniq_blocks, block_size)
        idx = np.arange(ci, n_blocks, ncls)
        if idx.size == 0:
            continue
        out[idx] = base[np.arange(idx.size) % uniq_blocks]
        if block_size >= 32:
            tag = idx.astype("<u8").view(np.uint8).reshape(-1, 8)
            tiles = np.arange(idx.size) // uniq_blocks
            late = tiles > 0
            out[idx[late], 16:24] = tag[late]
    return out


def silesia_blocks(n_blocks: int, block_size: int, corpus_dir: str) -> np.ndarray | None:
    """[n_blocks, block_size] of the REAL corpus ("if K4LZ4_CORPUS_DIR is set, use real Silesia cut into
    consecutive 64 KiB blocks"): the twelve files in name order, whole blocks only, tiled when the corpus is shorter than the
    batch.  None when a file is missing."""
    parts = []
    for name in SILESIA_NAMES:
        path = os.path.join(corpus_dir, name)
        if not os.path.exists(path):
            return None
        data = np.fromfile(path, dtype=np.uint8)
        k = data.size // block_size
        if k:
            parts.append(data[:k * block_size].reshape(k, block_size))
    if not parts:
        return None
    allb = np.concatenate(parts)
    return allb[np.arange(n_blocks) % allb.shape[0]].copy()


def variable_messages(n_msgs: int, seed: int = 4, lo: int = 1024, hi: int = 4 << 20,
                      budget_bytes: int | None = None) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """BASELINE.json configs[3]: message lengths log-uniform in [lo, hi], content alternating
    random / text-like.  Returns (packed bytes, uint64 offsets, int32 lengths)."""
    rng = np.random.default_rng(seed)
    lens = np.exp(rng.uniform(np.log(lo), np.log(hi), size=n_msgs)).astype(np.int64)
    if budget_bytes is not None:
        keep = np.cumsum(lens) <= budget_bytes
        keep[0] = True
        lens = lens[keep]
    offs = np.concatenate(([0], np.cumsum(lens)))
    total = int(offs[-1])
    text = _english(min(total, 8 << 20) + 16, np.random.default_rng(seed + 1))
    rnd = random_bytes(min(total, 8 << 20) + 16, seed + 2)
    out = np.empty(total, dtype=np.uint8)
    for i, (o, ln) in enumerate(zip(offs[:-1], lens)):
        src = rnd if (i & 1) == 0 else text
        start = int((i * 7919) % max(1, src.size - ln)) if ln < src.size else 0
        chunk = src[start:start + ln]
        if chunk.size < ln:
            chunk = np.tile(src, -(-int(ln) // src.size))[:ln]
        out[o:o + ln] = chunk
    return out, offs[:-1].astype(np.uint64), lens.astype(np.int32)


def config4_lengths(n_msgs: int = 100000, seed: int = 4, lo: int = 1024, hi: int = 4 << 20) -> np.ndarray:
    """BASELINE.json configs[3]: the message lengths of the whole batch (log-uniform in [lo, hi]); every rank computes the
    same vector and takes its byte-balanced range of it (sharding.byte_balanced_ranges)."""
    rng = np.random.default_rng(seed)
    return np.exp(rng.uniform(np.log(lo), np.log(hi), size=n_msgs)).astype(np.int64)


def config4_share(lens_all: np.ndarray, lo: int, hi: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    """Messages [lo, hi) of the configs[3] batch: content alternates random / text-like by GLOBAL message index, cut out of
    8 MiB of material at an index-dependent position, so any split of the batch into ranges produces the same bytes.
    Returns (packed bytes, uint64 offsets, int32 lengths)."""
    lens = lens_all[lo:hi].astype(np.int32)
    off = np.concatenate(([0], np.cumsum(lens.astype(np.int64))))[:-1].astype(np.uint64)
    total = int(lens.astype(np.int64).sum())
    text = class_bytes("dickens", 8 << 20, 5)
    rnd = random_bytes(8 << 20, 6)
    data = np.empty(max(total, 1), np.uint8)
    for i in range(lens.size):
        srcbuf = rnd if ((lo + i) & 1) == 0 else text
        st = ((lo + i) * 7919) % (srcbuf.size - int(lens[i]))
        data[int(off[i]):int(off[i]) + int(lens[i])] = srcbuf[st:st + int(lens[i])]
    return data[:total] if total else data[:0], off, lens
