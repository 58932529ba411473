"""BASELINE's full-size workload (cfg 3: 3840x2160, 64 spp, 23040x46080 DEM, 13680x27360 colour map) checked
through size-independent properties -- the oracle would take minutes there, so it is used on crops only."""
import os

import numpy as np
import pytest

from common import assert_bit_equal
from moonrtx_amd import _lib
from moonrtx_amd.renderer import MoonRT, DeviceBuffer, synth_ldem, synth_color, dem_from_ldem
from moonrtx_amd.scene import named_scene

pytestmark = pytest.mark.gpu

W, H, DEM_H, DEM_W, COL_H, COL_W = 3840, 2160, 23040, 46080, 13680, 27360


@pytest.fixture(scope="module")
def inputs(native_lib):
    src = synth_ldem(DEM_H, DEM_W)
    dem, scale = dem_from_ldem(src, DEM_H, DEM_W, 1)
    src.free()
    col = synth_color(COL_H, COL_W)
    yield dem, col, scale
    dem.free(); col.free()


def make(inputs, spp, **kw):
    dem, col, _ = inputs
    rt = MoonRT(W, H, **kw)
    rt.bind_dem(dem, DEM_H, DEM_W)
    rt.bind_color(col, COL_H, COL_W)
    rt.apply_scene(named_scene("S1", W, H, spp_per_launch=spp))
    return rt


def test_ingest_properties_at_full_size(inputs):
    """a1 at 1.06 G texels: peak exactly 1.0, LOLA-like range and radius scale (data_loader.py:236-242)."""
    dem, _, scale = inputs
    rows = dem.download(np.float32, (256, DEM_W))          # first 256 rows are enough for range sanity
    assert rows.max() <= 1.0 and rows.min() > 0.985
    assert 1.0055 < scale < 1.0065


def test_full_frame_is_deterministic_and_block_consistent(inputs):
    rt = make(inputs, 64)
    st1 = rt.render(1)
    a = rt.read_linear(); ha = rt.read_hits()
    rt.reset()
    st2 = rt.render(1)
    b = rt.read_linear()
    assert_bit_equal(a, b, "same frame twice")
    assert st1["height_samples"] == st2["height_samples"] > 5_000_000_000
    assert st1["primary_rays"] == W * H * 64 and st1["colour_fetches"] == st1["primary_hits"]
    # coverage channel == fraction of samples that hit; disk area ~ pi/4 * (0.9 H)^2 (moon_renderer.py:42)
    cov = float(a[..., 3].astype(np.float64).sum())
    assert abs(cov / (np.pi / 4 * (0.9 * H) ** 2) - 1.0) < 0.03
    assert abs(cov * 64 - st1["primary_hits"]) < 1.0
    # hit buffer: every covered pixel's hit lies in the displaced shell, at ~camera distance
    hd = ha[..., 3]
    r = np.linalg.norm(ha[..., :3], axis=-1)[hd > 0]
    assert 9.88 < r.min() and r.max() <= 10.0 + 1e-5 and 289.0 < hd[hd > 0].min() and hd.max() < 301.0
    # sky stays black, the lit limb carries light, night side is dark (zero ambient, moon_renderer.py:595)
    assert a[:40].max() == 0.0 and a[..., :3].max() > 0.15
    rt.close()


def test_16spp_blocks_accumulate_to_the_64spp_sample_set(inputs):
    """4 blocks of 16 spp visit the same (pixel, sample) set as 1 block of 64 spp: identical counters, and
    radiance equal up to float summation order."""
    rt64 = make(inputs, 64); s64 = rt64.render(1); a = rt64.read_linear(); rt64.close()
    rt16 = make(inputs, 16); s16 = rt16.render(4); b = rt16.read_linear(); rt16.close()
    for k in ("primary_rays", "primary_hits", "shadow_rays", "height_samples", "colour_fetches"):
        assert s64[k] == s16[k], k
    assert np.abs(a - b).max() < 2.0 ** -18


def test_sharded_full_frame_reassembles(inputs):
    """world = 2 on one GPU at full size: pack / unpack reproduces the single-context frame bit for bit."""
    rt = make(inputs, 16); rt.render(1); ref = rt.read_linear(); ref_h = rt.read_hits(); rt.close()
    r0 = make(inputs, 16, rank=0, world=2); r1 = make(inputs, 16, rank=1, world=2)
    r0.render(1); r1.render(1)
    buf = DeviceBuffer(r1.shard_bytes())
    r1.pack_shard(buf.ptr)
    r0.unpack_shard(1, buf.ptr)
    assert_bit_equal(r0.read_linear(), ref, "sharded radiance")
    assert_bit_equal(r0.read_hits(), ref_h, "sharded hits")
    r0.close(); r1.close(); buf.free()


def test_crop_of_the_full_frame_matches_the_oracle(inputs):
    """The oracle on a 96x64 crop across the terminator of the full-size frame, full-size DEM."""
    from oracle import orc
    dem_b, col_b, _ = inputs
    rt = make(inputs, 16)
    st = rt.render(1)
    lin = rt.read_linear(); hits = rt.read_hits()
    rt.close()
    dem = dem_b.download(np.float32, (DEM_H, DEM_W))
    col = col_b.download(np.uint8, (COL_H, COL_W, 4))
    o = orc.Oracle(named_scene("S1", W, H, spp_per_launch=16), dem, col)
    for (x0, y0) in ((1500, 1000), (2300, 400), (1900, 1900)):
        reg = (x0, y0, x0 + 96, y0 + 64)
        o.blocks_done = 0
        o.render(1, reg)
        got = lin[y0:y0 + 64, x0:x0 + 96]
        want = o.linear()[y0:y0 + 64, x0:x0 + 96]
        assert_bit_equal(got, want, f"crop at {x0},{y0}")
        assert_bit_equal(hits[y0:y0 + 64, x0:x0 + 96], o.hits[y0:y0 + 64, x0:x0 + 96], "crop hits")
    assert orc.quad_out_of_range() == 0


@pytest.fixture(scope="module")
def host_inputs(inputs):
    dem_b, col_b, _ = inputs
    return dem_b.download(np.float32, (DEM_H, DEM_W)), col_b.download(np.uint8, (COL_H, COL_W, 4))


# 64x48 crops of the cfg3 frame: across the terminator, the lit limb with sky beside it, the disc centre, the dark limb
# ... and a piece of pure sky (black without an environment map; with the star map: render_kernel<MODE 3> and its one-texel-per-pixel shortcut)
HEADLINE_CROPS = ((1500, 1000), (2860, 1060), (1888, 1056), (930, 1300), (96, 64))


def headline_scene():
    s = named_scene("S1", W, H, spp_per_launch=64)
    s.path_seg_min, s.path_seg_max = 2, 4                 # moon_renderer.py:583
    return s


def check_crops(rt, s, host_inputs, bg=None, what=""):
    from oracle import orc
    dem, col = host_inputs
    lin = rt.read_linear(); hits = rt.read_hits(); img = rt.read_rgba8(); img16 = rt.read_rgb16()
    o = orc.Oracle(s, dem, col, bg)
    lit = 0
    for (x0, y0) in HEADLINE_CROPS:
        reg = (x0, y0, x0 + 64, y0 + 48)
        o.reset()
        o.render(1, reg)
        sl = (slice(y0, y0 + 48), slice(x0, x0 + 64))
        assert_bit_equal(lin[sl], o.linear()[sl], f"{what} radiance, crop at {x0},{y0}")
        assert_bit_equal(hits[sl], o.hits[sl], f"{what} hits, crop at {x0},{y0}")
        assert np.array_equal(img[sl], o.rgba8(s.exposure, s.gamma)[sl]), f"{what} RGBA8, crop at {x0},{y0}"
        assert np.array_equal(img16[sl], o.rgb16(s.exposure, s.gamma)[sl]), f"{what} RGB16, crop at {x0},{y0}"
        lit += int((lin[sl][..., :3].max(-1) > 0.02).sum())
    assert lit > 3000 and orc.quad_out_of_range() == 0
    return lin


def test_headline_frame_crops_match_the_oracle(inputs, host_inputs, monkeypatch):
    """THE workload bench.py times -- cfg3, colour map, the reference's path_seg_range (2, 4) (moon_renderer.py:583), 64 spp,
    production kernels (flags = 0: no counters), WIDE addressing, record queue + path_kernel + resolve, default scheduling
    (no MOONRT_* test override) -- against the oracle on four crops: linear radiance, hit buffer, the 8-bit image the GUI
    shows and the 16-bit save_image data, all bit for bit."""
    monkeypatch.delenv("MOONRT_PATH_QUEUE_MIN", raising=False)
    monkeypatch.delenv("MOONRT_DEFAULT_FLAGS", raising=False)
    s = headline_scene()
    rt = make(inputs, 64)
    rt.apply_scene(s)
    rt.set_params(flags=0)
    st = rt.render(1)
    n_sub = max(1, int(os.environ.get("MOONRT_PATH_OVERLAP", "0") or 0))      # the overlapped path stage renders in sub-parts
    assert st["launches"] == 3 * n_sub and st["paths_ms"] > 0.0   # render_kernel<MODE 2> + path_kernel + resolve_paths_kernel
    check_crops(rt, s, host_inputs, what="headline")
    rt.close()


def test_headline_frame_with_the_star_map_crops_match_the_oracle(inputs, host_inputs, monkeypatch):
    """The same frame with the reference's default environment (a 16384x8192 star map, moon_renderer.py:604-607; synthetic
    here): sky tiles rendered by render_kernel<MODE 3>, escaping continuation rays look their texel up (PS_ESCAPED)."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    import bench
    monkeypatch.delenv("MOONRT_PATH_QUEUE_MIN", raising=False)
    monkeypatch.delenv("MOONRT_DEFAULT_FLAGS", raising=False)
    stars = bench.synth_starmap(8192, 16384)
    s = headline_scene()
    rt = make(inputs, 64)
    rt.upload_background(stars)
    rt.apply_scene(s)
    rt.set_params(flags=0)
    st = rt.render(1)
    n_sub = max(1, int(os.environ.get("MOONRT_PATH_OVERLAP", "0") or 0))
    assert st["launches"] == 3 * n_sub + 1                     # + the sky-only tiles' own launch
    lin = check_crops(rt, s, host_inputs, bg=stars, what="star map")
    assert lin[:40, :, :3].max() > 0.0                         # the sky is not black any more
    rt.close()


def test_production_kernels_equal_the_counting_kernels_at_full_size(inputs):
    """The kernels bench.py times carry no counters (flags = 0): separate template instantiations with their own
    register allocation.  At full size they must reproduce the counting kernels' frame -- which the crop test above pins
    to the oracle -- bit for bit, for the headline path and for the reference's path_seg_range (2, 4)."""
    for spp, seg in ((64, (1, 1)), (16, (1, 1)), (64, (2, 4))):
        frames = []
        for flags in (_lib.F_COUNT_STATS, 0):
            rt = make(inputs, spp)
            s = named_scene("S1", W, H, spp_per_launch=spp)
            s.path_seg_min, s.path_seg_max = seg
            rt.apply_scene(s)
            rt.set_params(flags=flags)
            rt.render(1)
            frames.append((rt.read_linear(), rt.read_hits()))
            rt.close()
        assert_bit_equal(frames[1][0], frames[0][0], f"production vs counting radiance, {spp} spp, path_seg {seg}")
        assert_bit_equal(frames[1][1], frames[0][1], f"production vs counting hits, {spp} spp, path_seg {seg}")


def _skip_vs_full(rt_factory, what):
    """One frame with every result-preserving skip (max-mip interval, medium-mip scans, horizon cut) and one with none of them
    (MRTX_F_NO_SKIP): radiance, hits and the spec counters must be the same bits over the WHOLE frame."""
    from common import STAT_KEYS
    out = {}
    for tag, flags in (("skip", _lib.F_COUNT_STATS), ("full", _lib.F_COUNT_STATS | _lib.F_NO_SKIP)):
        rt = rt_factory()
        rt.set_params(flags=flags)
        st = rt.render(1)
        out[tag] = (rt.read_linear(), rt.read_hits(), st)
        rt.close()
    assert_bit_equal(out["skip"][0], out["full"][0], f"{what}: skip vs full radiance")
    assert_bit_equal(out["skip"][1], out["full"][1], f"{what}: skip vs full hits")
    a, b = out["skip"][2], out["full"][2]
    assert {k: a[k] for k in STAT_KEYS} == {k: b[k] for k in STAT_KEYS}, (what, a, b)
    assert b["mip_fetches"] == 0 and 0 < a["dem_fetches"] < 0.6 * b["dem_fetches"]
    return a


def test_skips_are_result_preserving_over_the_whole_full_size_frame(inputs):
    """Every skip is a PROOF that a step lies above the surface: at cfg3's texel density -- 3.7 texels per
    march step times tan(incidence), where a segment's steps cross several medium-mip cells and grazing rays at the limb run through
    hundreds of them -- the whole 4K frame (8.3 M pixels x 4 samples, paths of 2-4 segments) must not differ in one bit from the
    frame marched without any of them.  Scenes S1 (phase 77 deg) and S3 (135 deg: grazing light, long shadows)."""
    for name in ("S1", "S3"):
        def factory():
            rt = make(inputs, 4)
            s = named_scene(name, W, H, spp_per_launch=4)
            s.path_seg_min, s.path_seg_max = 2, 4
            rt.apply_scene(s)
            return rt
        _skip_vs_full(factory, name)


def test_skips_are_result_preserving_on_a_spiked_dem(inputs, host_inputs):
    """... and on terrain built to defeat a careless bound: single-texel spikes 7 km tall and pits 7 km deep, three-texel mesas and
    cliffs along lines that are not aligned with any mip cell, so that neighbouring medium-mip cells differ by a dozen march steps
    and a test that looks at the wrong cell, or at a position a texel off, misses a hit."""
    dem = host_inputs[0].copy()
    dem[11::97, 7::89] += 0.004            # spikes (D is clipped to its maximum 1.0 below: the bounding sphere is R)
    dem[40::101, 33::83] -= 0.004          # pits
    for dr in range(3):
        for dc in range(3):
            dem[60 + dr::211, 15 + dc::197] += 0.003      # mesas
    dem[:, 5::1009] += 0.002               # meridional walls one texel wide
    dem[13::1013, :] += 0.002              # zonal walls
    np.clip(dem, 0.98, 1.0, out=dem)
    col_b = inputs[1]
    for name, vfov in (("S1", None), ("S1", 0.7)):
        def factory():
            rt = MoonRT(W, H)
            rt.upload_dem(dem)
            rt.bind_color(col_b, COL_H, COL_W)
            if vfov:
                from moonrtx_amd.scene import zoomed_on_terminator
                s = zoomed_on_terminator(name, W, H, vfov_deg=vfov, spp_per_launch=2)
            else:
                s = named_scene(name, W, H, spp_per_launch=2)
            s.path_seg_min, s.path_seg_max = 2, 4
            rt.apply_scene(s)
            return rt
        st = _skip_vs_full(factory, f"spiked DEM, {name}, vfov {vfov}")
        assert st["primary_hits"] > 1_000_000


def test_quadratic_texel_coordinates_error_budget_in_radiance(inputs):
    """The spec evaluates texel coordinates exactly at three anchors per 16-step segment and by a quadratic in between
    (DESIGN.md section 3.3).  Against exact evaluation at EVERY step (orc.set_exact) at cfg3's texel density, 64 spp:
    a few samples per thousand land on the other side of a hit / shadow decision (the two marches are equally valid
    discretisations that differ by <= 7e-3 texels), which moves single limb pixels by a few 1/64ths of a sample and
    the image by ~1e-5 on average.  Measured numbers for six crops: profiles/r02_exact_vs_quad.json."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import exact_vs_quad
    from oracle import orc
    dem_b, col_b, _ = inputs
    dem = dem_b.download(np.float32, (DEM_H, DEM_W))
    col = col_b.download(np.uint8, (COL_H, COL_W, 4))
    try:
        for name, crop in (("S1", (2300, 400, 64, 48)), ("S2", (2750, 700, 64, 48)), ("S3", (2300, 1500, 64, 48))):
            r = exact_vs_quad.measure(dem, col, name, crop)
            assert r["fraction"] < 0.01, r                      # samples that differ at all
            assert r["pixel_mean_abs_64spp"] < 5e-5, r
            assert r["pixel_linf_64spp"] < 4.0 / 64.0 * 0.3, r   # at most a few flipped samples of ~0.3 radiance in a pixel
    finally:
        orc.set_exact(False)
