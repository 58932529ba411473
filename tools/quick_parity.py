#!/usr/bin/env python3
"""S = 64 parity check for an A/B build (MOONRT_LIB=ab/libmoonrt_<x>.so, possibly built with -DMRTX_DEV_ONLY_S64): a small
(2, 4) frame with colour map through the path queue, production and counting kernels, and a direct-light frame, against
the oracle bit for bit.  usage: MOONRT_LIB=... python tools/quick_parity.py"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MOONRT_PATH_QUEUE_MIN", "0")
import numpy as np
import synth_np
from common import render_hip, render_oracle, assert_bit_equal, STAT_KEYS
from moonrtx_amd.scene import named_scene

dem = synth_np.dem(360, 720, seed=5, craters=60)
col = synth_np.colour_map(180, 360)
bg = np.random.default_rng(5).integers(0, 255, (48, 96, 4), dtype=np.uint8)
for name, seg, env, vfov in (("S1", (2, 4), None, None), ("S3", (2, 4), bg, 12.0), ("S1", (1, 1), None, None), ("S2", (1, 3), None, 1.0)):
    s = named_scene(name, 96, 64, spp_per_launch=64)
    s.path_seg_min, s.path_seg_max = seg
    if vfov:
        s.vfov_deg = vfov
    lin_o, hits_o, st_o = render_oracle(s, dem, col, env)
    for flags in (1, 0):
        lin, hits, st, _ = render_hip(s, dem, col, env, flags=flags)
        assert_bit_equal(lin, lin_o, f"{name} {seg} flags {flags} radiance")
        assert_bit_equal(hits, hits_o, f"{name} {seg} flags {flags} hits")
        if flags:
            assert {k: st[k] for k in STAT_KEYS} == {k: st_o[k] for k in STAT_KEYS}, (st, st_o)
    print("ok", name, seg, "env" if env is not None else "", flush=True)
print("quick parity: all bit-exact")
