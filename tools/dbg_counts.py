import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import synth_np
from moonrtx_amd import _lib
from moonrtx_amd.renderer import MoonRT
from moonrtx_amd.scene import named_scene
dem = synth_np.dem(360, 720, seed=5, craters=60)
for name, spp in (("S1", 16), ("S3", 4)):
    s = named_scene(name, 120, 90, spp_per_launch=spp)
    for tag, flags in (("skip", 1), ("full", 1 | 4), ("full-nocull", 1 | 4 | 8), ("skip-nocull", 1 | 8)):
        rt = MoonRT(s.width, s.height)
        rt.upload_dem(dem); rt.apply_scene(s); rt.set_params(flags=flags)
        st = rt.render(1)
        print(name, tag, {k: st[k] for k in ("primary_rays", "primary_hits", "height_samples", "dem_fetches", "mip_fetches")})
        rt.close()
