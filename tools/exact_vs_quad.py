#!/usr/bin/env python3
"""How much does the spec's per-segment quadratic of the texel coordinates move the RADIANCE?

The oracle renders crops of the cfg3 frame (3840x2160, DEM 23040x46080, colour map, 64 spp) twice: as specified, and with
every march step and bisection point evaluated exactly (orc.set_exact).  Reported per crop: samples whose radiance or hit
record differs at all, per-pixel L_inf of the 64-spp mean, and the mean absolute difference.  Test infrastructure (uses
the oracle); the DEM is synthesised on the GPU.  usage: tools/exact_vs_quad.py [out.json]"""
import json, os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
from moonrtx_amd.renderer import synth_ldem, synth_color, dem_from_ldem
from moonrtx_amd.scene import named_scene
from oracle import orc

W, H, DH, DW, CH, CW = 3840, 2160, 23040, 46080, 13680, 27360


def measure(dem, col, name, crop, seg=(1, 1), spp=64):
    x0, y0, w, h = crop
    reg = (x0, y0, x0 + w, y0 + h)
    per = {}
    for exact in (False, True):
        orc.set_exact(exact)
        s = named_scene(name, W, H, spp_per_launch=1)
        s.path_seg_min, s.path_seg_max = seg
        o = orc.Oracle(s, dem, col)
        frames = np.zeros((spp, h, w, 4), np.float32)
        hits0 = None
        for g in range(spp):                      # one sample per block: block g == global sample g of the 64-spp frame
            o.accum[:] = 0; o.blocks_done = g
            o.render(1, reg)
            frames[g] = o.accum[y0:y0 + h, x0:x0 + w]
            if g == 0:
                hits0 = o.hits[y0:y0 + h, x0:x0 + w].copy()
        per[exact] = (frames, hits0)
    orc.set_exact(False)
    a, b = per[False][0], per[True][0]
    diff_samples = int((a.view(np.uint32) != b.view(np.uint32)).any(-1).sum())
    mean_a, mean_b = a.mean(0, dtype=np.float64), b.mean(0, dtype=np.float64)
    d = np.abs(mean_a - mean_b)[..., :3]
    hit_a = a[..., 3] > 0
    return {"scene": name, "crop": list(crop), "path_seg": list(seg), "samples": int(a.shape[0] * w * h),
            "samples_on_moon": int(hit_a.sum()),
            "samples_that_differ": diff_samples, "fraction": diff_samples / float(a.shape[0] * w * h),
            "pixel_linf_64spp": float(d.max()), "pixel_mean_abs_64spp": float(d.mean()),
            "pixels_over_2^-10": int((d.max(-1) > 2.0 ** -10).sum()), "pixels": w * h,
            "hit_record_linf": float(np.abs(per[False][1] - per[True][1]).max())}


def main():
    t = time.time()
    src = synth_ldem(DH, DW); demb, _ = dem_from_ldem(src, DH, DW, 1); src.free()
    dem = demb.download(np.float32, (DH, DW)); demb.free()
    colb = synth_color(CH, CW); col = colb.download(np.uint8, (CH, CW, 4)); colb.free()
    orc.set_threads(min(os.cpu_count() or 1, 32))
    out = []
    for name, crop in (("S1", (1500, 1000, 96, 64)), ("S1", (2300, 400, 96, 64)), ("S1", (1900, 1900, 96, 64)),
                       ("S2", (1900, 1050, 96, 64)), ("S2", (2750, 700, 96, 64)), ("S3", (2300, 1500, 96, 64))):
        r = measure(dem, col, name, crop)
        out.append(r); print(json.dumps(r), flush=True)
    r = measure(dem, col, "S1", (1500, 1000, 96, 64), seg=(2, 4)); out.append(r); print(json.dumps(r), flush=True)
    print(f"total {time.time() - t:.0f} s", flush=True)
    if len(sys.argv) > 1:
        json.dump(out, open(sys.argv[1], "w"), indent=1)


if __name__ == "__main__":
    main()
