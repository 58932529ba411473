"""Known-answer and independence checks of the C oracle (CPU only).

The reference has no tests; SURVEY.md section 4 lists the statements in its comments that can serve as known
answers.  The numpy ray-march (oracle/numpy_march.py) is the independent restatement of the model."""
import math

import numpy as np
import pytest

import synth_np
from oracle import orc, numpy_march
from moonrtx_amd import scene as sc


def smooth_dem(h=90, w=180):
    return np.ones((h, w), np.float32)


def test_lambert_calibration_on_a_smooth_sphere(oracle_lib):
    """moon_renderer.py:77-83 / SURVEY 8(a): brightness 100 maps a normally lit texel to its texture value:
    radiance = albedo * (b/100) * cos(theta_i)."""
    s = sc.make_scene(96, 96, phase_deg=0.0, bright_limb_deg=0.0, spp_per_launch=64, brightness=100.0, libration=(0, 0))
    o = orc.Oracle(s, smooth_dem())
    o.render(1)
    lin = o.linear()
    centre = lin[48, 48, :3]
    # sub-solar point, cos = 1; the light radius follows the Sun distance (moon_renderer.py:859), and
    # illumination scales with angular size squared
    expect = 75.0 / 255.0 * (s.light_radius / 100.0) ** 2
    assert np.allclose(centre, expect, rtol=2e-3)
    # Lambert falloff: pixel at sin(theta) = 0.5 of the disk radius -> cos(theta) = sqrt(0.75)
    disk_r_px = 0.9 * 96 / 2 * (math.asin(10 / 300.0) / math.atan(10 / 300.0))
    x = 48 + int(round(0.5 * disk_r_px))
    assert abs(lin[48, x, 0] / centre[0] - math.sqrt(0.75)) < 0.03


def test_disk_fills_ninety_percent_of_the_height(oracle_lib):
    """moon_renderer.py:42, :58-63."""
    s = sc.make_scene(64, 200, 0.0, 0.0, spp_per_launch=1, libration=(0, 0))
    o = orc.Oracle(s, smooth_dem())
    o.render(1)
    col = o.hits[:, 32, 3] > 0
    assert abs(col.sum() / 200.0 - 0.9) < 0.02


def test_illumination_depends_on_angular_size_not_light_distance(oracle_lib):
    """moon_renderer.py:77-83: scale (distance, radius) together => same image (up to float rounding)."""
    dem = synth_np.dem(90, 180, seed=3, craters=8)
    s1 = sc.make_scene(48, 48, 60.0, -90.0, spp_per_launch=16)
    s2 = sc.make_scene(48, 48, 60.0, -90.0, spp_per_launch=16)
    s2.light_pos = tuple(3.0 * t for t in s1.light_pos)
    s2.light_radius = 3.0 * s1.light_radius
    a = orc.Oracle(s1, dem); a.render(1)
    b = orc.Oracle(s2, dem); b.render(1)
    la, lb = a.linear()[..., :3], b.linear()[..., :3]
    assert la.max() > 0.05
    assert abs(la.mean() - lb.mean()) / la.mean() < 0.02


def test_cone_shadow_length(oracle_lib):
    """moon_renderer.py:96-98 (Piazzi Smyth: 18.1 km rendered vs 18.9 km geometric at 2.8 deg Sun): an
    isolated peak of height h throws a shadow whose tip lies where the grazing ray meets the (curved) ground:
    x tan(alt) - x^2 / (2 R) = h."""
    h, w = 3600, 7200                      # 1.5 km texels
    dem = synth_np.cone_dem(h, w, 0.0, 0.0, 2.5, 12.0)
    h_eff = (1.0 / float(dem[10, 10]) - 1.0) * 1737.4      # summit as the bilinear DEM really holds it
    alt = 5.0
    s = sc.make_scene(160, 160, phase_deg=90.0 - alt, bright_limb_deg=-90.0, spp_per_launch=16, libration=(0, 0),
                      brightness=100)
    s.vfov_deg = 0.09
    o = orc.Oracle(s, dem)
    o.render(1)
    row = o.linear()[80, :, 0]
    km_per_px = 2 * 300.0 * math.tan(math.radians(s.vfov_deg / 2)) / 160 * 173.74
    lit_level = np.median(row[row > 0])
    dark = np.flatnonzero(row < 0.25 * lit_level)
    assert dark.size > 20
    # the Sun is to the right (+X): the shadow runs left of the summit (px 80)
    length_km = (80 - dark.min()) * km_per_px
    ta = math.tan(math.radians(alt))
    geometric = 1737.4 * (ta - math.sqrt(ta * ta - 2 * h_eff / 1737.4))
    assert 2.0 < h_eff < 2.5 and abs(length_km - geometric) / geometric < 0.08, (length_km, geometric, h_eff)


def test_scene_epsilon_guard(oracle_lib):
    """moon_renderer.py:88-93: with the shadow origin lifted by scene_epsilon a smooth sunlit sphere must not
    shadow itself anywhere."""
    s = sc.make_scene(64, 64, 40.0, 20.0, spp_per_launch=4, libration=(0, 0))
    o = orc.Oracle(s, smooth_dem(180, 360))
    st = o.render(1)
    lin = o.linear()
    hitmask = o.hits[..., 3] > 0
    # every hit pixel whose normal faces the light carries light: count dark hit pixels on the day side
    Ld = np.array(s.light_pos) / np.linalg.norm(s.light_pos)
    n = o.hits[..., :3] / 10.0
    day = hitmask & ((n @ Ld) > 0.05)
    assert day.sum() > 500 and (lin[day, 0] > 0).all()


@pytest.mark.parametrize("name", ["S1", "S2"])
def test_numpy_march_agrees_statistically(oracle_lib, name):
    """Independent restatement (library trig, float64, own RNG) vs the C oracle: same image up to Monte-Carlo
    noise.  Catches an error shared by the oracle and the HIP kernels, which follow one arithmetic spec."""
    dem = synth_np.dem(180, 360, seed=4, craters=25)
    s = sc.named_scene(name, 40, 40, spp_per_launch=64)
    o = orc.Oracle(s, dem)
    o.render(1)
    a = o.linear()[..., :3]
    b = numpy_march.render(s, dem, spp=64)
    assert a.max() > 0.05
    assert abs(a.mean() - b.mean()) / a.mean() < 0.03
    # per-pixel: differences are MC noise at shadow edges; bulk agreement
    diff = np.abs(a - b)[..., 0]
    assert np.median(diff) < 0.01 and (diff < 0.08).mean() > 0.97


@pytest.mark.parametrize("name", ["S1", "S2", "S3"])
def test_numpy_march_agrees_per_pixel_with_the_spec_rng(oracle_lib, name):
    """The independent march (float64, library trig, EXACT texel coordinates at every step, its own normal and shading
    code) fed with the spec's counter-based uniforms and light-cone parameterisation traces the same rays as the C oracle
    up to rounding: every pixel agrees to 1e-5 in linear radiance -- a hundred times inside the north star's 2^-10 --
    unless one of its samples sits exactly on a hit / shadow decision (none on this terrain).  This is the check that the
    arithmetic spec (float32, polynomial atan, quadratic texel coordinates) computes the MODEL."""
    dem = synth_np.dem(360, 720, seed=5, craters=60)
    s = sc.named_scene(name, 96, 72, spp_per_launch=16)
    o = orc.Oracle(s, dem)
    o.render(1)
    a = o.linear()[..., :3].astype(np.float64)
    b = numpy_march.render(s, dem, spp=16, spec_rng=True)
    assert a.max() > 0.05
    d = np.abs(a - b).max(-1)
    assert d.max() < 1e-5, (d.max(), int((d > 1e-5).sum()))


def test_numpy_march_per_pixel_on_steep_terrain(oracle_lib):
    """Steep corrugated relief: a handful of samples land on the other side of a grazing hit / shadow decision in the two
    implementations (float32 vs float64, quadratic vs exact coordinates); everything else agrees to 1e-5."""
    dem = synth_np.corrugated_dem(360, 720)
    s = sc.named_scene("S1", 80, 60, spp_per_launch=16)
    o = orc.Oracle(s, dem)
    o.render(1)
    a = o.linear()[..., :3].astype(np.float64)
    b = numpy_march.render(s, dem, spp=16, spec_rng=True)
    d = np.abs(a - b).max(-1)
    assert (d < 1e-5).mean() > 0.99 and d.max() < 2.0 / 16.0 * a.max() and d.mean() < 2e-5


def test_dem_ingest_matches_numpy_semantics(oracle_lib):
    """data_loader.py:223-242: two-stage float32 block mean, scale, +1, /max -- the oracle's C restatement
    against numpy evaluating the same formula."""
    for d in (1, 2, 3, 5, 8):
        src = synth_np.ldem_source(12 * d, 20 * d, seed=d + 1, craters=3)
        got, scale = orc.dem_from_ldem(src, d)
        if d == 1:
            e = src.astype(np.float32) * np.float32(0.5 / 1737400.0)
        else:
            e = src.reshape(1, 12, d, 20, d).mean(4, dtype=np.float32).mean(2, dtype=np.float32).reshape(12, 20)
            e *= np.float32(0.5 / 1737400.0)
        e += np.float32(1.0)
        mx = float(e.max()); e /= np.float32(mx)
        assert np.array_equal(got.view(np.uint32), e.view(np.uint32)), d
        assert np.float32(scale) == np.float32(mx) and got.max() == 1.0


def test_multi_bounce_paths_behave_physically(oracle_lib):
    """D6 (set_uint("path_seg_range", 2, 4), moon_renderer.py:580-583): inter-reflection only ADDS light, puts some
    into directly-shadowed pixels, scales with albedo^2 (one more reflection), and is small next to the direct term."""
    dem = synth_np.corrugated_dem(720, 1440)

    def render(alb, seg):
        s = sc.named_scene("S1", 72, 72, spp_per_launch=64)
        s.const_albedo = (alb,) * 3
        s.path_seg_min, s.path_seg_max = seg
        o = orc.Oracle(s, dem)
        st = o.render(1)
        return o.linear()[..., :3].astype(np.float64), st

    d1, st1 = render(0.3, (1, 1))
    b1, stb = render(0.3, (2, 4))
    assert st1["bounce_rays"] == 0 and stb["bounce_rays"] >= stb["primary_hits"] > 0
    assert stb["bounce_rays"] <= 3 * stb["primary_hits"]              # at most 3 continuation segments per path
    ind1 = b1 - d1
    assert ind1.min() > -1e-6 and 0.0 < ind1.mean() < 0.25 * d1.mean()
    assert (ind1[..., 0] > 0).sum() > 0.3 * (d1[..., 0] > 0).sum()      # inter-reflection is widespread on steep relief
    d2, _ = render(0.6, (1, 1))
    b2, _ = render(0.6, (2, 4))
    assert abs(d2.mean() / d1.mean() - 2.0) < 1e-3                       # direct term is linear in albedo
    ratio = (b2 - d2).mean() / ind1.mean()
    assert 3.4 < ratio < 5.2, ratio                                      # ~ albedo^2 (+ higher orders)
    # (2,2): exactly one continuation ray per primary hit, no roulette
    _, st22 = render(0.3, (2, 2))
    assert st22["bounce_rays"] == st22["primary_hits"]


def test_tone_map_spec_levels_and_thresholds(oracle_lib):
    """level = #{j : x >= (float)pow((j - 0.5) / N, gamma)} -- the rounded (exposure * mean)^(1/gamma),
    decided by comparisons; a value ON a threshold gets the upper level, the float just below it the lower one."""
    for gamma in (0.5, 1.0, 2.2, 5.0):
        for n in (255, 65535):
            g32 = float(np.float32(gamma))          # tonemap_gamma is a float32 parameter (set_float); the table is built from it
            T = orc.tone_table(g32, n)
            assert T.shape == (n + 1,) and T[0] == 0.0 and np.all(np.diff(T[1:]) >= 0) and T[n] < 1.0
            assert np.array_equal(T[1:], np.power((np.arange(1, n + 1) - 0.5) / n, g32).astype(np.float32))
            js = np.unique(np.concatenate([np.arange(1, 8), np.arange(n - 6, n + 1), np.linspace(1, n, 50).astype(int)]))
            on = T[js]
            below = np.nextafter(on, np.float32(-1))
            x = np.concatenate([on, below, [0.0, -1.0, np.nan, 1.0, 7.5, np.inf]]).astype(np.float32)
            acc = np.zeros((x.size, 4), np.float32)
            acc[:, 0] = x * 4            # accumulated over 4 samples with exposure 1: x * 4 / 4 == x exactly
            if n == 255:
                out = np.empty((x.size, 4), np.uint8)
                oracle_lib.orc_resolve_rgba8(acc.ctypes.data, x.size, 4, 1.0, gamma, None, out.ctypes.data)
                lv = out[:, 0].astype(int)
                assert (out[:, 3] == 255).all() and (out[:, 1] == 0).all()
            else:
                out = np.empty((x.size, 3), np.uint16)
                oracle_lib.orc_resolve_rgb16(acc.ctypes.data, x.size, 4, 1.0, gamma, out.ctypes.data)
                lv = out[:, 0].astype(int)
            k = len(js)
            want_on = np.searchsorted(T[1:], on, side="right")          # ties between equal thresholds count all of them
            assert np.array_equal(lv[:k], want_on) and np.all(lv[:k] >= js)
            assert np.array_equal(lv[k:2 * k], np.searchsorted(T[1:], below, side="right")) and np.all(lv[k:2 * k] < js)
            assert lv[2 * k:].tolist() == [0, 0, 0, n, n, n]
    # against the textbook formula in float64: equal except within rounding of a level boundary
    rng = np.random.default_rng(9)
    x = rng.random(20000).astype(np.float32) * np.float32(1.3)
    acc = np.zeros((x.size, 4), np.float32); acc[:, 0] = x
    out = np.empty((x.size, 4), np.uint8)
    oracle_lib.orc_resolve_rgba8(acc.ctypes.data, x.size, 1, 0.9, 2.2, None, out.ctypes.data)
    ref = np.floor(np.clip((np.float32(0.9) * x).astype(np.float64) ** (1 / 2.2), 0, 1) * 255 + 0.5).astype(int)
    assert np.abs(out[:, 0].astype(int) - ref).max() <= 1 and (out[:, 0] != ref).mean() < 1e-4
