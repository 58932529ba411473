"""Debug tool: statistics of the device-generated synthetic LDEM and a small rendered PNG."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from moonrtx_amd.renderer import MoonRT, synth_ldem, synth_color, dem_from_ldem
from moonrtx_amd.scene import named_scene

h, w = 2880, 5760
src = synth_ldem(h, w)
raw = src.download(np.int16, (h, w)).astype(np.float64) * 0.5 / 1000.0   # km
wts = np.cos(np.radians(90 - (np.arange(h) + 0.5) * 180 / h))[:, None] * np.ones((1, w))
mean = (raw * wts).sum() / wts.sum()
std = np.sqrt(((raw - mean) ** 2 * wts).sum() / wts.sum())
print("km: min %.2f max %.2f mean %.3f std %.3f" % (raw.min(), raw.max(), mean, std))
print("percentiles km", np.percentile(raw, [0.1, 1, 10, 50, 90, 99, 99.9]).round(2))
gy, gx = np.gradient(raw[h // 2 - 200:h // 2 + 200, :800] * 1000.0)
tex_m = 2 * np.pi * 1737400.0 / w
print("equatorial rms slope deg at texel scale %.0f m: %.2f" % (tex_m, np.degrees(np.arctan(np.sqrt((gx**2 + gy**2).mean()) / tex_m))))
dem, scale = dem_from_ldem(src, h, w, 1)
d = dem.download(np.float32, (h, w))
print("dem min %.5f mean %.5f max %.5f radius_scale %.5f" % (d.min(), (d * wts).sum() / wts.sum(), d.max(), scale))
col = synth_color(1368, 2736)
for name in ("S1", "S2", "S3"):
    s = named_scene(name, 960, 540, spp_per_launch=16)
    rt = MoonRT(960, 540)
    rt.bind_dem(dem, h, w); rt.bind_color(col, 1368, 2736); rt.apply_scene(s)
    st = rt.render(1)
    print(name, {k: st[k] for k in ("primary_hits", "shadow_rays", "height_samples", "kernel_ms")},
          "samples/hit %.1f" % (st["height_samples"] / st["primary_hits"]))
    img = rt.read_rgba8()
    from PIL import Image
    Image.fromarray(img[..., :3]).save(f"gpurun_out/synth_{name}.png")
    rt.close()
