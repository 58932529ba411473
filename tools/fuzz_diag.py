#!/usr/bin/env python3
"""Show where one fuzz case differs: tools/fuzz_diag.py <seed> <case> -- pixels, radiance and hit records of both sides.
Test infrastructure."""
import itertools, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MOONRT_PATH_QUEUE_MIN", "0")
import numpy as np
import fuzz_cases
from common import render_hip, render_oracle
from moonrtx_amd import _lib

seed, k = int(sys.argv[1]), int(sys.argv[2])
desc, dem, col, bg, s, flags, tile, blocks, extra = next(itertools.islice(fuzz_cases.cases(seed), k, k + 1))
print(desc)
print("eye", s.eye, "target", s.target, "up", s.up, "centre", s.center, "radius", s.radius)
if os.environ.get("FUZZ_FLAGS"): flags = int(os.environ["FUZZ_FLAGS"])
caps = extra["capsules"]
if os.environ.get("NO_CAPS"): caps = None
lin_o, hits_o, st_o = render_oracle(s, dem, col, bg, blocks=blocks, capsules=caps)
lin_h, hits_h, st_h, _ = render_hip(s, dem, col, bg, blocks=blocks, tile=tile, flags=flags, capsules=caps)
bad = np.argwhere((lin_o.view(np.uint32) != lin_h.view(np.uint32)).any(axis=2) | (hits_o.view(np.uint32) != hits_h.view(np.uint32)).any(axis=2))
print(len(bad), "pixels differ")
for (y, x) in bad[:12]:
    print(f"({y},{x}) tile ({x // tile[0]},{y // tile[1]}): hip lin {lin_h[y, x]} hit {hits_h[y, x]} | oracle lin {lin_o[y, x]} hit {hits_o[y, x]}")
print({k_: (st_h[k_], st_o[k_]) for k_ in st_o if k_ in st_h and st_h[k_] != st_o[k_]})
