#!/usr/bin/env python3
"""Hunt for the round-1 miscompute of the heavily spilled in-wave path kernels.

Round 1: with a 72-VGPR cap the instantiation render_kernel<64, STATS, !WIDE, in-wave paths, OVERLAY>
(~290 spilled VGPRs) miscomputed one sample; the counting variants were then given a 2-waves launch bound and the case
was not kept.  This script brings the configuration back -- build with
    tools/build_variant.sh spill -DMRTX_BOUNCE_STATS_WAVES=7 -DMRTX_MIN_WAVES_BOUNCE=7
(or `make -C moonrtx_amd/csrc spilltest` -> moonrtx_amd/libmoonrt_spilltest.so, which the GPU suite uses)
-- and runs every fuzz case through exactly that instantiation (S = 64, counters on, paths in the wave, overlay tubes,
32-bit DEM offsets) against the oracle, printing every mismatch with its first differing pixel.
usage: MOONRT_LIB=ab/libmoonrt_spill.so python tools/spill_repro.py [n_cases] [seed]"""
import itertools, os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fuzz_cases
from common import render_hip, render_oracle
from moonrtx_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
bad = 0
for k, c in enumerate(itertools.islice(fuzz_cases.cases(seed), n)):
    desc, dem, col, bg, s, flags, tile, blocks, extra = c
    s.spp_per_launch = 64
    if s.path_seg_max < 2:
        s.path_seg_min, s.path_seg_max = 2, 4
    caps = extra["capsules"]
    if caps is None:                                  # force the OVERLAY instantiation: one tube outside the sphere
        ctr = np.asarray(s.center, float)
        caps = np.zeros((1, 12), np.float32)
        caps[0, 0:3] = ctr + np.array([0.0, 0.0, 1.1]) * s.radius; caps[0, 4:7] = ctr + np.array([0.4, 0.0, 1.05]) * s.radius
        caps[0, 3] = 0.01 * s.radius; caps[0, 8:11] = (0.9, 0.5, 0.1)
    fl = (_lib.F_COUNT_STATS | _lib.F_INWAVE_PATHS | (flags & (_lib.F_NO_SKIP | _lib.F_NO_CULL | _lib.F_NO_SORT))) & ~_lib.F_FORCE_WIDE
    lin_o, hits_o, st_o = render_oracle(s, dem, col, bg, blocks=(1,), capsules=caps)
    lin_h, hits_h, st_h, _ = render_hip(s, dem, col, bg, blocks=(1,), tile=tile, flags=fl, capsules=caps)
    same = (lin_h.view(np.uint32) == lin_o.view(np.uint32)).all() and (hits_h.view(np.uint32) == hits_o.view(np.uint32)).all()
    cnt_ok = all(st_h[key] == st_o[key] for key in st_o)
    if not (same and cnt_ok):
        bad += 1
        d = np.argwhere(lin_h.view(np.uint32) != lin_o.view(np.uint32))
        print("MISMATCH", desc, "first px", None if len(d) == 0 else tuple(d[0]), "n", len(d),
              "counters", {key: (st_h[key], st_o[key]) for key in st_o if st_h[key] != st_o[key]}, flush=True)
    elif k % 25 == 0:
        print("ok", k, flush=True)
print(f"{n} cases, {bad} mismatches")
sys.exit(1 if bad else 0)
