#!/usr/bin/env python3
"""Time-lapse video of the Moon, headless: the reference's video export (renderer_video.py:149-340) driven through the facade --
encoder_create / encoder_start, one converged accumulation cycle per frame, the accum-done callback stepping the time.

  python tools/render_timelapse.py --time 2025-03-01T18:00:00+00:00 --lat 52.2 --lon 21.0 --frames 60 --step-min 360 \
         --size 1280 720 --out gpurun_out/lunation.avi
(ephemeris: moonrtx_amd/ephemeris.py; synthetic LOLA-like DEM; Motion-JPEG AVI, moonrtx_amd/video.py)."""
import argparse, os, sys, threading, time
from datetime import datetime, timedelta
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from moonrtx_amd import ephemeris
from moonrtx_amd.renderer import synth_ldem, synth_color, dem_from_ldem
from moonrtx_amd.tkoptix import TkOptiX

ap = argparse.ArgumentParser()
ap.add_argument("--time", required=True, help="ISO 8601 with UTC offset: the first frame")
ap.add_argument("--lat", type=float, required=True)
ap.add_argument("--lon", type=float, required=True)
ap.add_argument("--frames", type=int, default=30)
ap.add_argument("--step-min", type=float, default=60.0, help="minutes of real time between frames")
ap.add_argument("--fps", type=float, default=30.0)
ap.add_argument("--bitrate", type=float, default=16.0, help="Mbit/s (a per-frame byte budget for the JPEG quality)")
ap.add_argument("--size", type=int, nargs=2, default=(1280, 720))
ap.add_argument("--downscale", type=int, default=8)
ap.add_argument("--out", default="gpurun_out/timelapse.avi")
a = ap.parse_args()

W, H = a.size
ephemeris.init(ephemeris.Observer(a.lat, a.lon, 0.0))
t0 = datetime.fromisoformat(a.time)
dh, dw = 46080 // a.downscale, 92160 // a.downscale
src = synth_ldem(dh, dw, device=0)
dem_buf, _ = dem_from_ldem(src, dh, dw, 1, device=0)
src.free()
col = synth_color(1024, 2048, device=0)

rt = TkOptiX(width=W, height=H)
rt.bind_device_inputs(dem_buf, dh, dw, col, (1024, 2048))


def show(when):
    s = ephemeris.scene_from_ephemeris(ephemeris.calculate_moon_ephemeris(when, False), W, H, spp_per_launch=64)
    s.path_seg_min, s.path_seg_max = 2, 4          # moon_renderer.py:583
    rt.apply_scene_desc(s)


show(t0)
os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
rt.encoder_create(fps=a.fps, bitrate=a.bitrate)
rt.encoder_start(a.out, a.frames)
done = threading.Event()
wall = time.perf_counter()


def accum_done(r):                                  # render thread, padlock held -- renderer_video.py:276
    k = r.encoded_frames()
    if k < a.frames and r.encoder_is_open():
        show(t0 + timedelta(minutes=a.step_min * k))
        r.refresh_scene()
    else:
        done.set()


rt.set_accum_done_cb(accum_done)
rt.start()
while not done.wait(10.0):
    print(f"{rt.encoded_frames()} of {a.frames} frames", flush=True)
dt = time.perf_counter() - wall
rt.set_accum_done_cb(None)
rt.close()
print(f"wrote {rt.encoder_file}: {a.frames} frames of {W}x{H} in {dt:.2f} s ({a.frames / dt:.1f} frames/s rendered at 64 spp and encoded), "
      f"{os.path.getsize(rt.encoder_file) / 1e6:.1f} MB")
