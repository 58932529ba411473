#!/usr/bin/env python3
"""Randomised bit-exact parity sweep, HIP path vs the CPU oracle:  python tools/fuzz_parity.py [n_cases] [seed]
(the case generator is tests/fuzz_cases.py; FUZZ_ONLY=<k> renders case k alone).  Test infrastructure."""
import itertools
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MOONRT_PATH_QUEUE_MIN", "0")   # these small frames go through the path queue too (as under pytest)
import fuzz_cases   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
only = os.environ.get("FUZZ_ONLY")
t0 = time.time()
hits = 0
for k, c in enumerate(itertools.islice(fuzz_cases.cases(seed), n)):
    if only is not None and k != int(only):
        continue
    try:
        st = fuzz_cases.check_case(c)
    except AssertionError as e:
        print("FAIL", e)
        raise SystemExit(1)
    hits += st["primary_hits"]
    if k % 10 == 0 or only is not None:
        print("ok", c[0], " hits", st["primary_hits"], flush=True)
print(f"{n} random cases bit-exact in {time.time() - t0:.1f} s (seed {seed}); {hits} primary hits in total")
