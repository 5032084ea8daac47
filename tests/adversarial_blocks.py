"""Blocks built to hit particular corners of the fast encoder (used by the emulator tests and the GPU parity tests)."""
import numpy as np
from k4os.compression.lz4_amd import corpus


def dense_four_byte_matches(n_words, total, seed):
    """a block that is nearly all 4-byte matches with no literals between them: `n_words` words of four bytes with distinct first
    bytes, once each, then word after word such that no pair of neighbours has stood together before (the match cannot grow into the
    next word) -- the most sequences per byte the format allows, short of the (65546 - 6) / 4 the records are sized for"""
    rng = np.random.default_rng(seed)
    words = np.zeros((n_words, 4), np.uint8)
    words[:, 0] = rng.permutation(256)[:n_words]
    words[:, 1:] = rng.integers(0, 256, (n_words, 3))
    out, cnt = [words[i] for i in range(n_words)], n_words * 4
    for s in range(1, n_words):
        x = 0
        for _ in range(n_words):
            if cnt + 4 > total - 12:
                break
            out.append(words[x]); cnt += 4
            x = (x + s) % n_words
    buf = np.concatenate(out)
    return np.concatenate([buf, rng.integers(0, 256, total - buf.size, dtype=np.uint8)]).astype(np.uint8)


def search_limit_at_block_end():
    """The search's 66th probe at the end of a block (LL64.fast.cs:156-172).  After a match the search probes 66 positions in a row
    and then every second one; the step a probe is LEFT with is the one worked out an iteration earlier, so probe number 65 (the
    last of the contiguous ones) is already checked against mflimitPlusOne with a step of two (:170-172): where it lies at
    U - 12 exactly, the reference does not make it and the block ends in 78 literals, although a match starts there.  Blocks: text,
    a repeated phrase (a match that ends where the noise begins), `noise` unmatchable bytes, then `tail` bytes copied from the
    text -- for every noise length 62 .. 70 and tail 8 .. 18, so that each of the last probes falls on either side of the limit."""
    rng = np.random.default_rng(97)
    text = corpus.lorem(900)
    blocks = []
    for noise in range(62, 71):
        for tail in range(8, 19):
            filler = rng.integers(128, 256, noise, dtype=np.uint8)          # (lorem is ASCII: these bytes match nothing)
            filler[0] = 0xFF
            blocks.append(np.concatenate([text, text[100:140], filler, text[300:300 + tail]]).astype(np.uint8))
    return blocks
