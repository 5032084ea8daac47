#!/usr/bin/env python
"""Extracts the known-answer rows of the reference's Tests/ChecksumBlockTests.cs (generated there from native lz4 1.9.2
builds, 32- and 64-bit: playground/SharedSources/app.cpp:79-141) into tests/golden/checksum_block_rows.json.
Each row: architecture (4 = LZ4Codec.Enforce32, 8 = default), Silesia file name, chunk index and length, LZ4Level,
expected compressed length, Adler32 of the compressed bytes, and their first 60 bytes (base64).
The Silesia corpus itself is not in the repository; tests/test_reference_goldens.py uses the rows when
K4LZ4_CORPUS_DIR points at it.  Usage (in the build container only): tests/tools/extract_checksum_rows.py"""
import json, os, re, sys
SRC = "/root/reference/src/K4os.Compression.LZ4.Tests/ChecksumBlockTests.cs"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "golden", "checksum_block_rows.json")
rows = []
pat = re.compile(r'\[InlineData\((\d+),\s*"\.corpus/([^"]+)",\s*(\d+),\s*(\d+),\s*(\d+),\s*(\d+),\s*(0x[0-9a-fA-F]+),\s*"([^"]+)"\)\]')
for line in open(SRC, encoding="utf-8-sig"):
    m = pat.search(line)
    if m:
        a, f, idx, ln, lvl, clen, adler, b64 = m.groups()
        rows.append({"architecture": int(a), "file": f, "index": int(idx), "length": int(ln), "level": int(lvl),
                     "compressed_length": int(clen), "adler32": int(adler, 16), "first_bytes_base64": b64})
json.dump({"source": "src/K4os.Compression.LZ4.Tests/ChecksumBlockTests.cs", "rows": rows}, open(OUT, "w"), indent=0)
print(len(rows), "rows ->", OUT)
