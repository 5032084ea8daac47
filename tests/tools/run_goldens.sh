#!/bin/bash
# The reference's own known-answer rows (Tests/ChecksumBlockTests.cs:14-50,:125-172; tests/golden/checksum_block_rows.json)
# against the oracle (120 rows; the 48 optimal-parser rows only with K4LZ4_GOLDEN_SLOW=1) and, on a box with a GPU, against
# libk4lz4.so (48 rows: levels 0 and 3, 64- and 32-bit engine).  Needs the Silesia corpus (tests/tools/fetch_silesia.md).
# Usage: tests/tools/run_goldens.sh <corpus-dir>      -> profiles/goldens_<date>.log
set -u
DIR=${1:?usage: tests/tools/run_goldens.sh <directory holding dickens, mozilla, mr, ...>}
ROOT=$(cd "$(dirname "$0")/../.." && pwd)
LOG=$ROOT/profiles/goldens_$(date +%Y%m%d).log
cd "$ROOT"
{
  echo "# corpus: $DIR"; ls -l "$DIR" | head -20
  echo "# oracle rows"
  K4LZ4_CORPUS_DIR="$DIR" python -m pytest tests/test_reference_goldens.py -q -m "not gpu" -rs 2>&1 | tail -40
  echo "# GPU rows"
  K4LZ4_CORPUS_DIR="$DIR" python -m pytest tests/test_reference_goldens.py -q -m gpu -rs 2>&1 | tail -40
} | tee "$LOG"
echo "written: $LOG"
