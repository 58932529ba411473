#!/usr/bin/env python3
"""Section timers of render_kernel (a -DMRTX_PROF build): MOONRT_LIB=ab/libmoonrt_prof.so python tools/prof_sections.py

s_memtime deltas summed over all waves of one cfg3 launch; shares are of the summed per-wave trace time."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import WORKLOADS
from moonrtx_amd import _lib
from moonrtx_amd.renderer import MoonRT, synth_ldem, synth_color, dem_from_ldem
from moonrtx_amd.scene import named_scene

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
sc = sys.argv[2] if len(sys.argv) > 2 else "S1"
W, H, spp, dem_h, dem_w, col_shape = WORKLOADS[wl]
src = synth_ldem(dem_h, dem_w, device=0)
dem_buf, _ = dem_from_ldem(src, dem_h, dem_w, 1, device=0)
src.free()
col = synth_color(col_shape[0], col_shape[1], device=0)
scene = named_scene(sc, W, H, spp_per_launch=64)
if os.environ.get("ZOOM"):      # the zoomed terminator view (bench.py's also_zoomed): ZOOM=0.7
    from moonrtx_amd.scene import zoomed_on_terminator
    scene = zoomed_on_terminator(sc, W, H, vfov_deg=float(os.environ["ZOOM"]), spp_per_launch=64)
if os.environ.get("PATH_SEG"):
    scene.path_seg_min, scene.path_seg_max = [int(t) for t in os.environ["PATH_SEG"].split(",")]
rt = MoonRT(W, H, device=0)
rt.bind_dem(dem_buf, dem_h, dem_w)
rt.bind_color(col, col_shape[0], col_shape[1])
rt.apply_scene(scene)
rt.set_params(flags=0)
lib = _lib.load()
fn = lib.mrtx_prof_read
fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * 16)()
rt.reset(); rt.render(1)
fn(out, 1)
rt.reset(); st = rt.render(1)
fn(out, 1)
v = list(out)
names = ["trace total", "primary setup", "primary march", "bisect", "hit_vertex", "direct_light (incl. shadow march)",
         "seg_setup (all marches)", "step loops (all marches)"]
print(f"{wl} {sc}: kernel {st['kernel_ms']:.3f} ms, waves {v[10]}")
for i, n in enumerate(names):
    print(f"  {n:36s} {v[i]:16d} ticks  {v[i] / max(1, v[0]):.3f}")
print(f"  wave-level step iterations {v[11]} ({v[11] / max(1, v[10]):.1f} per wave), mean lanes evaluating {v[12] / max(1, v[11]):.1f}")
print(f"  primary segments {v[15]} of which empty for the whole wave {v[13]}; shadow segments {v[8] - v[15]}, empty {v[14]}")
if os.environ.get("TRIAL"):   # a -DMRTX_PROF -DMRTX_PROF_TRIAL build, PATH_SEG=2,4
    print(f"  TRIAL: step iterations {v[13]} ({v[13] / max(1, v[10]):.2f} per wave), mean lanes evaluating {v[14] / max(1, v[13]):.1f}; set-up + steps {v[15] / v[0]:.3f} of the trace")
if os.environ.get("FULLIV"):   # a -DMRTX_PROF -DMRTX_PROF_FULLIV build
    print(f"  FULLIV: lane-segments {v[9]}, of which without a skip interval from the max-mip {v[13]} ({v[13] / max(1, v[9]):.3f}), exact-fallback {v[14]} ({v[14] / max(1, v[9]):.4f}); "
          f"of the former: footprint over more than two cell ROWS {v[5] / max(1, v[9]):.4f}, over more than the tap's COLUMNS {v[6] / max(1, v[9]):.4f}, map edge {v[7] / max(1, v[9]):.4f} "
          f"(section timers 5-7 are overwritten in this build)")
if os.environ.get("SPREAD"):
    print(f"  SPREAD: waves with a hit {v[9]}; lanes' texel spread at the hit <= 8: {v[13] / max(1, v[9]):.3f}, <= 16: {v[14] / max(1, v[9]):.3f}, <= 28: {v[15] / max(1, v[9]):.3f}")
print(f"  wave-level segments {v[8]}, mean lanes alive in a segment {v[9] / max(1, v[8]):.1f}")
