#!/usr/bin/env python3
"""Sweep the tuning knobs of the deferred path stage (MOONRT_PATH_REFILL / _HITMIN / _WAVES) on one GPU:
cfg3 inputs are built once, every setting gets its own context.  usage: tools/path_sweep.py [workload] [settings...]
a setting is refill,segmin,raremin,waves (waves 0 = occupancy default)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from moonrtx_amd import build, _lib
from moonrtx_amd.renderer import MoonRT, synth_ldem, synth_color, dem_from_ldem
from moonrtx_amd.scene import named_scene

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
settings = [tuple(int(t) for t in a.split(",")) for a in sys.argv[2:]] or [(8, 24, 16, 0)]
if not os.environ.get("MOONRT_LIB"):
    build.build_native()
W, H, spp, dem_h, dem_w, col_shape = bench.WORKLOADS[wl]
S = min(spp, 64)
dem_h //= int(os.environ.get('DEM_SCALE', '1')); dem_w //= int(os.environ.get('DEM_SCALE', '1'))
src = synth_ldem(dem_h, dem_w); dem, _ = dem_from_ldem(src, dem_h, dem_w, 1); src.free()
col = synth_color(*col_shape) if col_shape else None
scene = named_scene(os.environ.get("SCENE", "S1"), W, H, spp_per_launch=S)
scene.path_seg_min, scene.path_seg_max = 2, 4
for (refill, segmin, rare, waves) in settings:
    os.environ["MOONRT_PATH_REFILL"] = str(refill); os.environ["MOONRT_PATH_SEGMIN"] = str(segmin); os.environ["MOONRT_PATH_HITMIN"] = str(rare)
    if waves: os.environ["MOONRT_PATH_WAVES"] = str(waves)
    else: os.environ.pop("MOONRT_PATH_WAVES", None)
    rt = MoonRT(W, H)
    rt.bind_dem(dem, dem_h, dem_w)
    if col is not None: rt.bind_color(col, *col_shape)
    rt.apply_scene(scene); rt.set_params(flags=int(os.environ.get("FLAGS", "0")))
    rt.render(1)
    acc = [0.0, 0.0]
    for _ in range(3):
        rt.reset(); st = rt.render(spp // S)
        acc[0] += st["primary_ms"] / 3; acc[1] += st["paths_ms"] / 3
    print(f"refill {refill:2d} seg {segmin:2d} rare {rare:2d} waves {waves:5d}: primary {acc[0]:7.3f} ms  paths {acc[1]:7.3f} ms  total {acc[0]+acc[1]:7.3f}", flush=True)
    rt.close()
