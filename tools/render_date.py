#!/usr/bin/env python3
"""Render the Moon for a date and an observer, headless: ephemeris -> scene -> HIP render -> PNG.

  python tools/render_date.py --time 2025-03-07T19:30:00+01:00 --lat 52.2 --lon 21.0 --out gpurun_out/moon.png
(the reference's `--time/--lat/--lon` drive, main.py; synthetic LOLA-like DEM unless --elevation-file is given)."""
import argparse, os, sys, time
from datetime import datetime
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from moonrtx_amd import ephemeris
from moonrtx_amd.renderer import MoonRT, synth_ldem, synth_color, dem_from_ldem

ap = argparse.ArgumentParser()
ap.add_argument("--time", required=True, help="ISO 8601 with UTC offset")
ap.add_argument("--lat", type=float, required=True)
ap.add_argument("--lon", type=float, required=True)
ap.add_argument("--elevation-m", type=float, default=0.0)
ap.add_argument("--parallactic", action="store_true", help="equatorial mount: celestial north up")
ap.add_argument("--size", type=int, nargs=2, default=(1024, 1024))
ap.add_argument("--spp", type=int, default=64)
ap.add_argument("--downscale", type=int, default=8)
ap.add_argument("--elevation-file", default=None)
ap.add_argument("--out", default="gpurun_out/moon.png")
a = ap.parse_args()

ephemeris.init(ephemeris.Observer(a.lat, a.lon, a.elevation_m))
eph = ephemeris.calculate_moon_ephemeris(datetime.fromisoformat(a.time), a.parallactic)
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in eph._asdict().items() if k != "rotation_matrix"})
W, H = a.size
scene = ephemeris.scene_from_ephemeris(eph, W, H, spp_per_launch=min(64, a.spp))
scene.max_spp = a.spp
scene.path_seg_min, scene.path_seg_max = 2, 4     # what MoonRenderer.init_renderer sets (moon_renderer.py:583)
if a.elevation_file:
    from moonrtx_amd.ingest import load_elevation_data
    dem, _ = load_elevation_data(a.elevation_file, a.downscale, device=0)     # host float32 (h, w), as the reference returns it
    dh, dw = dem.shape
    dem_buf = None
else:
    dh, dw = 46080 // a.downscale, 92160 // a.downscale
    src = synth_ldem(dh, dw, device=0)
    dem_buf, _ = dem_from_ldem(src, dh, dw, 1, device=0)
    src.free()
col = synth_color(1024, 2048, device=0)
rt = MoonRT(W, H, device=0)
if dem_buf is None:
    rt.upload_dem(dem)
else:
    rt.bind_dem(dem_buf, dh, dw)
rt.bind_color(col, 1024, 2048)
rt.apply_scene(scene)
rt.reset()
t0 = time.perf_counter()
st = rt.render(max(1, a.spp // scene.spp_per_launch))
print(f"rendered {W}x{H} x {a.spp} spp in {st['kernel_ms']:.2f} ms (kernel), {time.perf_counter() - t0:.3f} s wall")
img = rt.read_rgba8()
os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
from PIL import Image
Image.fromarray(img[..., :3]).save(a.out)
lit = float((img[..., :3].max(axis=2) > 8).mean())
print(f"wrote {a.out}; lit fraction of the frame {lit:.3f}; expected illuminated fraction of the disc {(1 + np.cos(np.radians(eph.phase_angle))) / 2:.3f}")
