#!/usr/bin/env python3
"""Throughput of one launch at 3840x2160 for spp_per_launch = 1 .. 64 (a wave = 64/S pixels x S samples), cfg3 inputs,
path_seg_range PATH_SEG (default 2,4), paths behind the queue and inside the wave."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import WORKLOADS
from moonrtx_amd.renderer import MoonRT, synth_ldem, synth_color, dem_from_ldem
from moonrtx_amd.scene import named_scene

W, H, spp, dem_h, dem_w, col_shape = WORKLOADS["cfg3"]
src = synth_ldem(dem_h, dem_w, device=0)
dem_buf, _ = dem_from_ldem(src, dem_h, dem_w, 1, device=0)
src.free()
col = synth_color(col_shape[0], col_shape[1], device=0)
import time
seg = tuple(int(t) for t in os.environ.get("PATH_SEG", "2,4").split(","))
for S in (1, 2, 4, 8, 16, 32, 64):
    for flags in (0, 32):                       # 32 = MRTX_F_INWAVE_PATHS
        scene = named_scene("S1", W, H, spp_per_launch=S)
        scene.path_seg_min, scene.path_seg_max = seg
        rt = MoonRT(W, H, device=0)
        rt.bind_dem(dem_buf, dem_h, dem_w); rt.bind_color(col, col_shape[0], col_shape[1])
        rt.apply_scene(scene); rt.set_params(flags=flags)
        rt.reset(); rt.render(1)
        t, wall = [], []
        for _ in range(5):
            rt.reset()
            t0 = time.perf_counter(); st = rt.render(1); wall.append(time.perf_counter() - t0); t.append(st["kernel_ms"])
        ms = min(t)
        print(f"S={S:2d} {'in-wave' if flags else 'queue  '}: {ms:7.3f} ms kernels, {min(wall) * 1e3:7.3f} ms wall per launch, {W * H * S / ms / 1e3:9.0f} Msamples/s", flush=True)
        rt.close()
