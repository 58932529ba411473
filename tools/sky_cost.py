"""How long does the render kernel take on frames with no Moon in view (pure dispatch + ray-setup cost)?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from moonrtx_amd.renderer import MoonRT, synth_ldem, dem_from_ldem
from moonrtx_amd.scene import named_scene
h, w = 2880, 5760
src = synth_ldem(h, w); dem, _ = dem_from_ldem(src, h, w, 1)
for name, tgt in (("moon in view", (0, 0, 0)), ("all sky", (0, -600, 0))):
    s = named_scene("S1", 3840, 2160, spp_per_launch=64)
    s.target = tgt
    rt = MoonRT(3840, 2160)
    rt.bind_dem(dem, h, w); rt.apply_scene(s); rt.set_params(flags=0)
    rt.render(1)
    ts = []
    for _ in range(5):
        rt.reset(); ts.append(rt.render(1)["kernel_ms"])
    print(name, "kernel ms:", [round(t, 3) for t in ts])
    rt.close()
