"""Pin the oracle (and the product's host-side conventions) to golden vectors captured from the
reference's own code (tests/golden/*.json, made by tests/golden/make_golden.py in the build container)."""
import json
import math
import os

import numpy as np
import pytest

from oracle import orc
from moonrtx_amd import scene as sc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def seeded_dem(h, w, seed):
    """Same generator as tests/golden/make_golden.py::seeded_dem."""
    rng = np.random.default_rng(seed)
    e = (0.99 + 0.01 * rng.random((h, w))).astype(np.float32)
    e[rng.integers(0, h), rng.integers(0, w)] = 1.0
    return e


def load(name):
    with open(os.path.join(GOLD, name)) as f:
        return json.load(f)


def test_dem_bilinear_matches_get_elevation_m(oracle_lib):
    """renderer_navigation.py:558-599: texel-centre -0.5 offset, row 0 = +90, column 0 = -180, columns wrap,
    rows clamp.  The oracle samples in float32, the reference in float64: tolerance = float32 texel
    coordinate resolution times the local slope (DEM spans 1 % of the radius)."""
    g = load("elevation_bilinear.json")
    worst = 0.0
    for case in g["cases"]:
        h, w = case["h"], case["w"]
        dem = seeded_dem(h, w, case["seed"])
        for (lat, lon), want_m in zip(case["points"], case["elevation_m"]):
            d = orc.dem_bilinear(dem, math.radians(lat), math.radians(lon))
            got_m = (d * case["radius_scale"] - 1.0) * 1737.4 * 1000.0
            # 1 % relief = 17.4 km between neighbouring random texels; float32 coords are good to ~1e-5 texel
            tol = 17400.0 * 4e-5 * max(h, w) / 16 + 0.25
            worst = max(worst, abs(got_m - want_m))
            assert abs(got_m - want_m) <= tol, (h, w, lat, lon, got_m, want_m)
    assert worst < 5.0  # metres, against a 17 km texel-to-texel swing


def test_texel_centres_return_the_texel_exactly(oracle_lib):
    dem = seeded_dem(8, 16, 1)
    for r in range(8):
        for c in range(16):
            lat = 90.0 - (r + 0.5) * 180.0 / 8
            lon = -180.0 + (c + 0.5) * 360.0 / 16
            d = orc.dem_bilinear(dem, math.radians(lat), math.radians(lon))
            assert abs(d - dem[r, c]) < 2e-7


def test_body_frame_convention():
    """renderer_navigation.py:43-57 (lat/lon -> scene) and :452-492 (scene -> lat/lon)."""
    g = load("body_frame.json")
    for case in g["cases"]:
        R = np.array(case["rotation"])
        for row in case["rows"]:
            p = R @ sc.body_point(row["lat"], row["lon"])
            assert np.allclose(p, row["scene_pos"], atol=1e-12)
            lat, lon = sc.selenographic(row["scene_pos"], R)
            assert abs(lat - row["lat_back"]) < 1e-9 and abs(lon - row["lon_back"]) < 1e-9
        assert sc.selenographic((0.0, -20.0, 0.0), R) == (None, None)
        assert case["off_moon"] == [None, None]


@pytest.mark.parametrize("which", ["lat", "lon"])
def test_oracle_sphere_mapping_agrees_with_reference_convention(oracle_lib, which):
    """Render DEMs that ENCODE latitude / longitude in the surface radius, read the oracle's hit buffer,
    and check that the (lat, lon) the reference's hit_to_selenographic convention assigns to each hit is
    the one the oracle's own DEM addressing used."""
    h, w = 180, 360
    rows = (np.arange(h) + 0.5) / h
    cols = (np.arange(w) + 0.5) / w
    if which == "lat":
        dem = np.repeat((0.9 + 0.1 * rows)[:, None], w, 1)
    else:
        dem = np.repeat((0.9 + 0.1 * cols)[None, :], h, 0)
    dem = (dem / dem.max()).astype(np.float32)
    top = 0.9 + 0.1 * (rows[-1] if which == "lat" else cols[-1])
    s = sc.named_scene("S2", 96, 96, spp_per_launch=1, libration=(21.0, -13.0))
    s.marching_step_eps = 1.0e-5
    o = orc.Oracle(s, dem)
    o.render(1)
    hits = o.hits
    checked = 0
    for y in range(8, 88, 6):
        for x in range(8, 88, 6):
            hx, hy, hz, hd = hits[y, x]
            if hd <= 0:
                continue
            lat, lon = sc.selenographic((hx, hy, hz), s.rotation)
            r = math.sqrt(hx * hx + hy * hy + hz * hz) / 10.0
            enc = (r * top - 0.9) / 0.1          # row or column fraction the surface radius encodes
            if which == "lat":
                got = 90.0 - enc * 180.0
                if abs(lat) > 85:
                    continue
                assert abs(got - lat) < 0.08, (x, y, got, lat)
            else:
                got = enc * 360.0 - 180.0
                if abs(lon) > 170 or abs(lat) > 80:
                    continue
                assert abs(got - lon) < 0.15, (x, y, got, lon)
            checked += 1
    assert checked > 40


def test_haversine_convention():
    g = load("haversine.json")
    for case in g["cases"]:
        la1, lo1, la2, lo2 = (math.radians(t) for t in case["args"])
        a = math.sin((la2 - la1) / 2) ** 2 + math.cos(la1) * math.cos(la2) * math.sin((lo2 - lo1) / 2) ** 2
        km = 2 * math.atan2(math.sqrt(a), math.sqrt(1 - a)) * sc.MOON_RADIUS_KM
        assert abs(km - case["km"]) < 1e-9
