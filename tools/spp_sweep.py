#!/usr/bin/env python3
"""Throughput of one launch at 3840x2160 for spp_per_launch = 1 .. 64 (a wave = 64/S pixels x S samples), cfg3 inputs."""
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
for S in (1, 2, 4, 8, 16, 32, 64):
    scene = named_scene("S1", W, H, spp_per_launch=S)
    rt = MoonRT(W, H, device=0)
    rt.bind_dem(dem_buf, dem_h, dem_w); rt.bind_color(col, col_shape[0], col_shape[1])
    rt.apply_scene(scene); rt.set_params(flags=0)
    rt.reset(); rt.render(1)
    t = []
    for _ in range(5):
        rt.reset(); t.append(rt.render(1)["kernel_ms"])
    ms = min(t)
    print(f"S={S:2d}: {ms:7.3f} ms per launch, {W * H * S / ms / 1e3:9.0f} Msamples/s")
    rt.close()
