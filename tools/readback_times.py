#!/usr/bin/env python3
"""Host read-back cost of a finished 3840x2160 frame (what crosses PCIe if a caller wants pixels on the host)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from moonrtx_amd.renderer import MoonRT, synth_ldem, dem_from_ldem
from moonrtx_amd.scene import named_scene
W, H = 3840, 2160
src = synth_ldem(5760, 11520, device=0)
dem, _ = dem_from_ldem(src, 5760, 11520, 1, device=0)
rt = MoonRT(W, H, device=0)
rt.bind_dem(dem, 5760, 11520)
rt.apply_scene(named_scene("S1", W, H, spp_per_launch=64)); rt.set_params(flags=0)
rt.reset(); st = rt.render(1)
for name, fn, nbytes in (("read_rgba8 (resolve + 33 MB D2H)", rt.read_rgba8, W * H * 4), ("read_linear (resolve + 133 MB D2H)", rt.read_linear, W * H * 16),
                         ("read_hits (133 MB D2H)", rt.read_hits, W * H * 16)):
    fn()
    t = []
    for _ in range(5):
        t0 = time.perf_counter(); fn(); t.append(time.perf_counter() - t0)
    print(f"{name}: {min(t) * 1e3:.2f} ms = {nbytes / min(t) / 1e9:.1f} GB/s (pageable numpy destination)")
print(f"render kernel {st['kernel_ms']:.2f} ms")
