"""Hand-made stream sets shared by the emulator tests and the GPU tests (test infrastructure)."""
import numpy as np

from k4os.compression.lz4_amd import corpus


def long_field_streams(oracle, rng):
    """streams whose literal-length and match-length fields are runs of 255 (LL.tools.cs:165-193), whole and cut / damaged inside and
    right behind those runs: (stream, capacity, bytes_defined) triples -- a damaged byte may turn an offset into 0, which the
    reference does not reject and whose output is whatever the target held (only accept / reject and the return
    value are compared there)"""
    datas = [corpus.random_bytes(4096, 1), corpus.random_bytes(70000, 2), corpus.repeated(9, 70000), corpus.repeated(3, 300),
             np.concatenate([corpus.random_bytes(700, 3), corpus.repeated(1, 5000), corpus.random_bytes(16 + 255 * 3, 4), corpus.repeated(2, 19 + 255 * 2)]),
             np.concatenate([corpus.lorem(3000), corpus.random_bytes(15 + 255, 5), corpus.lorem(400)])]
    out = []
    for data in datas:
        good = np.frombuffer(oracle.encode(data), np.uint8)
        out.append((good, data.size, True))
        out.append((good, data.size + 40, True))
        runs = np.flatnonzero(good == 255)
        spots = sorted(set([1, 2, 16, 17, 18, 19, good.size - 1, good.size - 2] + [int(x) for x in runs[:3]] + [int(x) + 1 for x in runs[-3:]] +
                           [int(rng.integers(1, good.size)) for _ in range(12)]))
        for cut in spots:
            if 0 < cut < good.size:
                out.append((good[:cut].copy(), data.size, True))
                bad = good.copy(); bad[cut] = 255
                out.append((bad, data.size, True))
                bad = good.copy(); bad[cut] = 0
                out.append((bad, data.size + 7, False))
        out.append((np.concatenate([good, np.full(70, 255, np.uint8)]), data.size, True))
    # a field of nothing but 255s up to the end of the stream, and one that runs into the last five bytes
    out.append((np.concatenate([[0xF0], np.full(300, 255, np.uint8)]).astype(np.uint8), 100000, True))
    out.append((np.concatenate([[0x1F, 65, 1, 0], np.full(40, 255, np.uint8), [7, 0x50, 1, 2, 3, 4, 5]]).astype(np.uint8), 20000, True))
    return out


