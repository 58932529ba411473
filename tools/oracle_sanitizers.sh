#!/bin/bash
# The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer on 60 random fuzz scenes (CPU only; GPU sanitizers
# are not available on this pool).  usage: tools/oracle_sanitizers.sh
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
gcc -O1 -g -fPIC -shared -fopenmp -ffp-contract=off -fno-math-errno -mfma -msse4.1 -fsanitize=address,undefined \
    -fno-sanitize-recover=undefined "$ROOT/oracle/mrtx_oracle.c" -o /tmp/liborc_asan.so -lm
cat > /tmp/asan_run.py <<PY
import sys, itertools
sys.path.insert(0, "$ROOT"); sys.path.insert(0, "$ROOT/tests")
from oracle import orc
real = orc.C.CDLL
orc.C.CDLL = lambda path, *a, **k: real("/tmp/liborc_asan.so" if "liborc_" in str(path) else path, *a, **k)
import fuzz_cases
from common import render_oracle
n = 0
for c in itertools.islice(fuzz_cases.cases(3), 60):
    desc, dem, col, bg, s, flags, tile, blocks, extra = c
    render_oracle(s, dem, col, bg, blocks=blocks, capsules=extra["capsules"]); n += 1
print("sanitised oracle rendered", n, "fuzz cases cleanly")
PY
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0 python3 /tmp/asan_run.py
