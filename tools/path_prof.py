#!/usr/bin/env python3
"""Block statistics of path_kernel (a -DMRTX_PATH_PROF build): MOONRT_LIB=ab/libmoonrt_pprof.so python tools/path_prof.py [settings]
a setting is refill,segmin,raremin."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import WORKLOADS
from moonrtx_amd import _lib
from moonrtx_amd.renderer import MoonRT, synth_ldem, synth_color, dem_from_ldem
from moonrtx_amd.scene import named_scene

W, H, spp, dem_h, dem_w, col_shape = WORKLOADS["cfg3"]
src = synth_ldem(dem_h, dem_w); dem, _ = dem_from_ldem(src, dem_h, dem_w, 1); src.free()
col = synth_color(*col_shape)
scene = named_scene("S1", W, H, spp_per_launch=64); scene.path_seg_min, scene.path_seg_max = 2, 4
lib = _lib.load()
fn = lib.mrtx_pprof_read; fn.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
out = (C.c_ulonglong * 16)()
ft = lib.mrtx_pprof_times; ft.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
tms = (C.c_ulonglong * 16)()
settings = [tuple(int(t) for t in a.split(",")) for a in sys.argv[1:]] or [(32, 16, 16)]
for (refill, segmin, rare) in settings:
    os.environ.update(MOONRT_PATH_REFILL=str(refill), MOONRT_PATH_SEGMIN=str(segmin), MOONRT_PATH_HITMIN=str(rare))
    rt = MoonRT(W, H, rank=int(os.environ.get("RANK_", "0")), world=int(os.environ.get("WORLD_", "1"))); rt.bind_dem(dem, dem_h, dem_w); rt.bind_color(col, *col_shape); rt.apply_scene(scene); rt.set_params(flags=int(os.environ.get("FLAGS", "0")))
    if os.environ.get("STARMAP"):     # the reference's environment (moon_renderer.py:604-607); STARMAP=black: an all-zero one
        import bench
        sm = bench.synth_starmap(8192, 16384)
        if os.environ["STARMAP"] == "black": sm[:, :, :3] = 0
        rt.upload_background(sm)
    rt.render(1); fn(out, 1); ft(tms, 1)
    rt.reset(); st = rt.render(1); fn(out, 1); ft(tms, 1)
    v = list(out)
    print(f"refill {refill} seg {segmin} rare {rare}: paths {st['paths_ms']:.2f} ms; iterations/wave {v[0]/5120:.0f}")
    for i, n in enumerate(("refill", "set-up", "step", "rare")):
        ex, ln = v[1 + 2 * i], v[2 + 2 * i]
        print(f"    {n:7s} executions {ex:12d} ({ex / max(1, v[0]):.2f} per iteration)  lanes waiting {ln / max(1, ex):5.1f}")
    t = list(tms)
    print("    last wave of each label ends after " + ", ".join(f"{(t[i] - t[8]) / 100.0:.0f}" for i in range(8)) + " us (first wave start = 0)")
    if hasattr(lib, "mrtx_pprof_ends"):
        import numpy as np
        ends = (C.c_ulonglong * 5120)()
        lib.mrtx_pprof_ends.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
        lib.mrtx_pprof_ends(ends, 5120)
        e = (np.array(list(ends), np.float64) - t[8]) / 100.0
        e = np.sort(e[e > 0])
        print("    wave end times (us after the first wave's start): " + ", ".join(f"p{q}: {np.percentile(e, q):.0f}" for q in (1, 10, 50, 90, 99)) + f", last {e[-1]:.0f}; mean idle at the end {e[-1] - e.mean():.0f} us")
    tot = sum(v[9:14])
    if tot:
        names = ("bookkeeping + refill", "set-up", "step", "segment end + march over", "rare")
        print("    wave cycles: " + ", ".join(f"{n} {v[9 + i] / tot:.3f} ({v[9 + i] / max(1, v[(1, 3, 5, 0, 7)[i]]):.0f}/exec)" for i, n in enumerate(names)) + f"; {tot / 5120 / 1e6:.2f} M cycles per wave")
    rt.close()
