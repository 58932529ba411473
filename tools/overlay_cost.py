#!/usr/bin/env python3
"""Frame time at cfg3 with and without the reference's grid overlay (tests/golden/moon_grid_graphs.npz: 3900 capsules)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import numpy as np
from bench import WORKLOADS
from moonrtx_amd import overlays
from moonrtx_amd.renderer import MoonRT, synth_ldem, synth_color, dem_from_ldem
from moonrtx_amd.scene import named_scene

W, H, spp, dem_h, dem_w, col_shape = WORKLOADS["cfg3"]
src = synth_ldem(dem_h, dem_w, device=0)
dem_buf, _ = dem_from_ldem(src, dem_h, dem_w, 1, device=0)
src.free()
col = synth_color(col_shape[0], col_shape[1], device=0)
scene = named_scene("S1", W, H, spp_per_launch=64)
g = np.load(os.path.join(ROOT, "tests", "golden", "moon_grid_graphs.npz"))
R = np.asarray(scene.rotation, float)
caps = np.concatenate([overlays.graph_to_capsules(g["lines_pos"] @ R.T, g["lines_edges"], 0.006, [0.5] * 3),
                       overlays.graph_to_capsules(g["labels_pos"] @ R.T, g["labels_edges"], 0.012, [0.5] * 3)])
rt = MoonRT(W, H, device=0)
rt.bind_dem(dem_buf, dem_h, dem_w); rt.bind_color(col, col_shape[0], col_shape[1])
rt.apply_scene(scene); rt.set_params(flags=0)
for name, c in (("no overlay", None), ("grid lines + labels (3900 capsules)", caps)):
    rt.set_capsules(c)
    rt.reset(); rt.render(1)
    t = []
    for _ in range(5):
        rt.reset(); t.append(rt.render(1)["kernel_ms"])
    print(f"{name}: {min(t):.3f} ms")
