"""Shared helpers for the parity tests: run the same scene through the HIP path and the oracle."""
import numpy as np

from oracle import orc
from moonrtx_amd.renderer import MoonRT

EXTRA_KEYS = ("dem_fetches", "mip_fetches", "kernel_ms", "primary_ms", "paths_ms", "launches",
              "camera_height_samples", "camera_dem_fetches", "camera_mip_fetches", "camera_colour_fetches", "camera_background_fetches")
STAT_KEYS = ("primary_rays", "primary_hits", "shadow_rays", "height_samples", "colour_fetches",
             "background_fetches", "bounce_rays", "bounce_sun_hits")


def render_hip(scene, dem, color=None, bg=None, blocks=(1,), rank=0, world=1, tile=(16, 16), flags=1, capsules=None):
    """tile: 16 x 16 is what the library itself uses on one or two GPUs (round 3's default); tests that exercise other tilings pass
    theirs."""
    rt = MoonRT(scene.width, scene.height, rank=rank, world=world, tile=tile)
    try:
        rt.upload_dem(dem)
        rt.upload_color(color)
        rt.upload_background(bg)
        rt.apply_scene(scene)
        rt.set_capsules(capsules)
        rt.set_params(flags=flags)
        stats = {k: 0 for k in STAT_KEYS + EXTRA_KEYS}
        for nb in blocks:
            st = rt.render(nb)
            for k in STAT_KEYS + EXTRA_KEYS:
                stats[k] += st[k]
        return rt.read_linear(), rt.read_hits(), stats, rt.read_rgba8()
    finally:
        rt.close()


def render_oracle(scene, dem, color=None, bg=None, blocks=(1,), region=None, capsules=None, rgba8=False):
    o = orc.Oracle(scene, dem, color, bg, capsules)
    for nb in blocks:
        st = o.render(nb, region)
    if rgba8:       # + the tone-mapped 8-bit frame (exposure / gamma of the scene), what render_hip returns fourth
        return o.linear(), o.hits.copy(), st, o.rgba8(scene.exposure, scene.gamma)
    return o.linear(), o.hits.copy(), st


def assert_bit_equal(a, b, what):
    a = np.ascontiguousarray(a); b = np.ascontiguousarray(b)
    same = a.view(np.uint32) == b.view(np.uint32)
    if not same.all():
        bad = np.argwhere(~same)
        diff = np.abs(a.astype(np.float64) - b.astype(np.float64))
        raise AssertionError(f"{what}: {len(bad)} of {a.size} values differ bitwise; max |diff| = {diff.max():.3e}; "
                             f"first at {tuple(bad[0])}: {a[tuple(bad[0])]!r} vs {b[tuple(bad[0])]!r}")
