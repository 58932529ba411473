#!/usr/bin/env python3
"""Parity at the extremes the random sweep does not draw: minimal DEMs / maps / frames.  Test infrastructure."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("MOONRT_PATH_QUEUE_MIN", "0")   # these small frames go through the path queue too (as under pytest)
import numpy as np
from common import STAT_KEYS, assert_bit_equal, render_hip, render_oracle
from moonrtx_amd.scene import named_scene

rng = np.random.default_rng(99)
n = 0
for (h, w) in ((2, 2), (2, 3), (3, 5), (5, 4), (7, 64), (64, 7)):
    dem = (0.97 + 0.03 * rng.random((h, w))).astype(np.float32); dem.flat[rng.integers(0, dem.size)] = 1.0
    for (W, H) in ((1, 1), (1, 7), (9, 1), (33, 17)):
        for S in (1, 4, 64):
            for seg in ((1, 1), (2, 3)):
                s = named_scene("S1", W, H, spp_per_launch=S, libration=(float(rng.uniform(-180, 180)), float(rng.uniform(-80, 80))))
                s.path_seg_min, s.path_seg_max = seg
                if W * H < 10:
                    s.vfov_deg = 1.0
                col = rng.integers(0, 256, (2, 2, 4), dtype=np.uint8) if rng.random() < 0.5 else None
                bg = rng.integers(0, 256, (1, 1, 4), dtype=np.uint8) if rng.random() < 0.5 else None
                lin_h, hits_h, st_h, _ = render_hip(s, dem, col, bg, tile=(16, 16))
                lin_o, hits_o, st_o = render_oracle(s, dem, col, bg)
                what = f"dem {h}x{w} frame {W}x{H} S={S} seg={seg}"
                assert_bit_equal(lin_h, lin_o, what); assert_bit_equal(hits_h, hits_o, what + " hits")
                assert {k: st_h[k] for k in STAT_KEYS} == {k: st_o[k] for k in STAT_KEYS}, what
                n += 1
print(f"{n} extreme cases bit-exact")
