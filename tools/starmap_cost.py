#!/usr/bin/env python3
"""cfg3 with and without the reference's environment (moon_renderer.py:604-607: a 16384x8192 star map bound as
"TextureEnvironment"): every tile is dispatched (the sky is no longer black), every path that leaves the Moon adds light."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from moonrtx_amd.renderer import MoonRT, synth_ldem, synth_color, dem_from_ldem
from moonrtx_amd.scene import named_scene

W, H, spp, dem_h, dem_w, col_shape = bench.WORKLOADS["cfg3"]
src = synth_ldem(dem_h, dem_w); dem, _ = dem_from_ldem(src, dem_h, dem_w, 1); src.free()
col = synth_color(*col_shape)
stars = bench.synth_starmap(8192, 16384)
for seg, S in (((2, 4), 64), ((1, 1), 64), ((2, 4), 1)):
    for bg in (None, stars):
        scene = named_scene("S1", W, H, spp_per_launch=S)
        scene.path_seg_min, scene.path_seg_max = seg
        rt = MoonRT(W, H); rt.bind_dem(dem, dem_h, dem_w); rt.bind_color(col, *col_shape)
        rt.apply_scene(scene); rt.set_params(flags=int(os.environ.get("FLAGS", "0")))
        rt.upload_background(bg)
        rt.render(1)
        t = []
        for _ in range(3):
            rt.reset(); st = rt.render(1); t.append((st["kernel_ms"], st["primary_ms"], st["paths_ms"]))
        k, p, q = min(t)
        print(f"{S:2d} spp per launch, path_seg_range {seg}, {'star map 16384x8192' if bg is not None else 'no environment  '}: {k:7.3f} ms (render {p:.3f} + paths {q:.3f})", flush=True)
        rt.close()
