#!/usr/bin/env python3
"""Hunt for host-side corruption of the fuzz cases' INPUT arrays (tests/fuzz_cases.py::InputGuard): run cases of several seeds, do
not stop at an InputChanged, log every occurrence with all differing words, and count them.  python tools/guard_hunt.py [cases] [seed0] [seeds]
Test infrastructure."""
import itertools
import os
import sys
import time

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MOONRT_PATH_QUEUE_MIN", "0")
import numpy as np   # noqa: E402
import fuzz_cases   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 7
n_seeds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
changed = parity = ok = 0
sizes = {}
t0 = time.time()
for seed in range(seed0, seed0 + n_seeds):
    for k, c in enumerate(itertools.islice(fuzz_cases.cases(seed), n)):
        inputs = {"dem": c[1], "colour": c[2], "environment": c[3], "capsules": c[8]["capsules"]}
        snap = {k_: (None if v is None else v.copy()) for k_, v in inputs.items()}
        try:
            fuzz_cases.check_case(c)
            ok += 1
        except fuzz_cases.InputChanged as e:
            changed += 1
            print("INPUT CHANGED:", e, flush=True)
            for k_, v in inputs.items():        # restore, so that the generator's later cases are not affected
                if v is not None and not np.array_equal(v, snap[k_]):
                    sizes[v.nbytes] = sizes.get(v.nbytes, 0) + 1
                    v[...] = snap[k_]
        except AssertionError as e:
            parity += 1
            print("PARITY FAILURE:", str(e)[:600], flush=True)
print(f"pooling {os.environ.get('MOONRT_POOL_STREAMS', 'default(on)')}: {ok} cases clean, {changed} with a changed input (array sizes {sizes}), "
      f"{parity} parity failures, {time.time() - t0:.0f} s")
