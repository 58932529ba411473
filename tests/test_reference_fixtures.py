"""Host rows of SURVEY.md section 8 (a1, a2, a4 closed forms, a5-a9) against outputs of the REFERENCE'S OWN functions,
executed in the build container by tests/golden/make_golden_host.py and stored as data under tests/golden/host_*.

Nothing here re-types a reference formula: every expected value was computed by /root/reference/moonrtx/*.py itself."""
import json
import os
import sys
import threading

import numpy as np
import pytest

from moonrtx_amd import ephemeris as E
from moonrtx_amd import ingest
from moonrtx_amd import scene as sc
from oracle import orc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
from make_golden_host import ldem_source  # noqa: E402  (the seeded INPUT generator; imports nothing of the reference)


def _load(name):
    return json.load(open(os.path.join(GOLD, name)))


# ---------------------------------------------------------------------------------------------------- a2
def test_albedo_lut_equals_the_references_table():
    """data_loader._albedo_lut (data_loader.py:272-287) for six gammas: all 256 bytes equal."""
    g = _load("host_albedo.json")
    assert len(g["lut"]) == 6
    for gamma, want in g["lut"].items():
        got = ingest.albedo_lut(float(gamma))
        assert got.dtype == np.uint8 and got.tolist() == want, gamma


def test_moon_texture_equals_the_references_rgba():
    """data_loader._moon_texture (data_loader.py:345-368): BGR bytes -> RGBA through the LUT."""
    g = _load("host_albedo.json")
    bgr = np.array(g["bgr"], np.uint8)
    for gamma, want in g["texture"].items():
        got = ingest.moon_texture(bgr, float(gamma), order="BGR")
        assert np.array_equal(got, np.array(want, np.uint8)), gamma
        assert np.array_equal(ingest.moon_texture(bgr[..., ::-1], float(gamma), order="RGB"), got)


# ---------------------------------------------------------------------------------------------------- a1
def _elevation_cases():
    meta = _load("host_elevation.json")
    z = np.load(os.path.join(GOLD, "host_elevation.npz"))
    for c in meta["cases"]:
        src = z[f"src{c['i']}"] if c["src_stored"] else ldem_source(c["h"], c["w"], c["seed"])
        yield c, src, z[f"elev{c['i']}"]


def test_oracle_dem_ingest_equals_load_elevation_data(oracle_lib):
    """data_loader.load_elevation_data (data_loader.py:213-242) executed on seeded int16 sources, d = 1, 2, 3, 5, 8: the
    oracle's C restatement gives the same float32 bits and the same radius_scale."""
    n = 0
    for c, src, want in _elevation_cases():
        if c["src_stored"]:
            assert np.array_equal(src, ldem_source(c["h"], c["w"], c["seed"]))     # the generator is reproducible
        got, scale = orc.dem_from_ldem(src, c["downscale"])
        assert got.shape == want.shape and want.dtype == np.float32
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), c
        assert np.float32(scale) == np.float32(c["radius_scale"]) and got.max() == 1.0
        n += 1
    assert n == 6


def test_cache_sidecar_has_the_references_keys(tmp_path):
    """The .json the reference wrote beside <src>.ds<N>.npy (data_loader.py:22-95) against ours for the same file."""
    meta = _load("host_elevation.json")
    p = str(tmp_path / "ldem_1.tif")
    open(p, "wb").write(b"II*\0")
    c = [c for c in meta["cases"] if c["downscale"] == 2][0]
    assert c["cache_files"] == ["ldem_1.tif.ds2.json", "ldem_1.tif.ds2.npy"]
    fp = {**ingest.cache_fingerprint(p, downscale=2), "radius_scale": c["radius_scale"]}
    assert sorted(fp.keys()) == c["cache_json_keys"]
    assert {k: v for k, v in fp.items() if k != "source_mtime"} == c["cache_json"]
    assert meta["cases"][0]["cache_files"] == []          # downscale 1 is never cached


@pytest.mark.gpu
def test_device_dem_ingest_equals_load_elevation_data(native_lib):
    """The same fixtures through mrtx_dem_from_ldem on the GPU (ldem_block_mean_kernel + scale_by_inv_kernel)."""
    for c, src, want in _elevation_cases():
        buf, h, w, scale = ingest.elevation_to_device(src, c["downscale"])
        got = buf.download(np.float32, (h, w))
        buf.free()
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), c
        assert np.float32(scale) == np.float32(c["radius_scale"])


# ---------------------------------------------------------------------------------------------------- a4
def test_astro_closed_forms_equal_the_references():
    """astro._wrap_signed_degrees, _colongitude_from_subsolar_longitude, _parallactic_angle_deg, _latlon_from_icrf,
    _rotation_matrix, _body_altitude_at_feature (astro.py:84-139, :166-184)."""
    g = _load("host_astro.json")
    for a, want in g["wrap"]:
        assert E.wrap_signed_degrees(a) == want
    for a, want in g["colong"]:
        assert E.colongitude_from_subsolar_longitude(a) == want
    for ha, dec, lat, want in g["parallactic"]:
        assert E.parallactic_angle_deg(ha, dec, lat) == pytest.approx(want, abs=1e-12)
    for c in g["rotation"]:
        got = E.rotation_matrix(np.array(c["R_moon"]), np.array(c["R_equator"]), c["ra"], c["dec"], c["q"])
        assert np.abs(got - np.array(c["matrix"])).max() < 1e-14
    for c in g["latlon_from_icrf"]:
        la, lo = E.latlon_from_icrf(c["pos_au"], np.array(c["R"]))
        assert la == pytest.approx(c["lat"], abs=1e-12) and lo == pytest.approx(c["lon"], abs=1e-12)
    for c in g["altitude"]:
        got = E.body_altitude_at_feature(np.array(c["sub_lat"]), np.array(c["sub_lon"]), c["lat"], c["lon"])
        assert np.abs(got - np.array(c["alt"])).max() < 1e-12
    assert len(g["rotation"]) == 8 and len(g["parallactic"]) == 16


def test_view_rotation_reproduces_the_references_matrix_for_a_libration():
    """astro._rotation_matrix for a body frame whose sub-observer point is (l, b) and whose pole stands at angle P:
    ephemeris.view_rotation(l, b, P - q) -- what calculate_moon_ephemeris hands MoonRenderer -- is that matrix."""
    g = _load("host_astro.json")
    for c in g["rotation"]:
        M = np.array(c["matrix"])
        # read (l, b, P - q) off the reference's matrix: the body point facing the camera and the pole's roll
        facing = M.T @ np.array([0.0, -1.0, 0.0])
        b = np.degrees(np.arcsin(facing[2])); l = np.degrees(np.arctan2(facing[0], -facing[1]))
        pole = M @ np.array([0.0, 0.0, 1.0])
        roll = np.degrees(np.arctan2(-pole[0], pole[2]))
        assert np.abs(E.view_rotation(l, b, roll) - M).max() < 1e-12


# ---------------------------------------------------------------------------------------------------- a5-a7
def test_scene_constants_equal_the_references():
    k = _load("host_scene.json")["constants"]
    pairs = [("MOON_RADIUS", sc.MOON_RADIUS), ("MOON_RADIUS_KM", sc.MOON_RADIUS_KM), ("MOON_FILL_FRACTION", sc.MOON_FILL_FRACTION),
             ("CAMERA_DISTANCE", sc.CAMERA_DISTANCE), ("MOON_REFERENCE_DISTANCE", sc.MOON_REFERENCE_DISTANCE_KM),
             ("SUN_LIGHT_DISTANCE", sc.SUN_LIGHT_DISTANCE), ("SUN_BRIGHTNESS_SCALE", sc.SUN_BRIGHTNESS_SCALE),
             ("SCENE_EPSILON", sc.SCENE_EPSILON), ("MARCHING_STEP", sc.MARCHING_STEP), ("MARCHING_STEP_EPS", sc.MARCHING_STEP_EPS),
             ("SUN_RADIUS_KM", sc.SUN_RADIUS_KM), ("SUN_DISK_DISTANCE", sc.SUN_DISK_DISTANCE), ("SUN_DISK_COLOR", sc.SUN_DISK_COLOR),
             ("SUN_DISK_PARKED_RADIUS", sc.SUN_DISK_PARKED_RADIUS), ("ACCUMULATION_FRAMES", sc.ACCUMULATION_FRAMES)]
    for name, ours in pairs:
        assert float(k[name]) == float(ours), name


def test_light_camera_and_sun_disk_equal_the_references():
    """MoonRenderer.moon_apparent_radius / moon_camera_distance / default_camera / calculate_light_pos /
    calculate_sun_disk (moon_renderer.py:507-568, :653-778) on twelve ephemerides, incl. a parked disk, eclipse
    geometry (separation ~0) and perigee / apogee distances."""
    cases = _load("host_scene.json")["cases"]
    assert len(cases) == 12
    parked = 0
    for c in cases:
        assert sc.apparent_radius(c["distance"]) == pytest.approx(c["apparent_radius"], rel=1e-15)
        assert sc.camera_distance(c["distance"]) == pytest.approx(c["camera_distance"], rel=1e-15)
        assert np.allclose(sc.light_position(c["phase_angle"], c["bright_limb_angle"]), c["light_pos"], rtol=0, atol=1e-9)
        centre, r = sc.sun_disk(c["distance"], c["sun_distance"], c["elongation"], c["bright_limb_angle"])
        assert np.allclose(centre, c["sun_disk_pos"], rtol=0, atol=1e-9) and r == pytest.approx(c["sun_disk_radius"], rel=1e-13)
        parked += r == sc.SUN_DISK_PARKED_RADIUS
        cam = c["default_camera"]
        assert cam["eye"] == [0, -sc.camera_distance(c["distance"]), 0] or np.allclose(cam["eye"], [0, -sc.camera_distance(c["distance"]), 0], atol=1e-12)
        assert cam["target"] == [0, 0, 0] and cam["up"] == [0, 0, 1] and cam["type"] == "Pinhole"
        assert sc.default_vfov_deg() == pytest.approx(cam["fov"], rel=1e-15)
        # the headless scene builder (bench / tests) is the same composition
        s = sc.make_scene(64, 32, c["phase_angle"], c["bright_limb_angle"], distance_km=c["distance"],
                          sun_distance_km=c["sun_distance"], elongation_deg=c["elongation"])
        assert np.allclose(s.light_pos, c["light_pos"], atol=1e-9) and np.allclose(s.sun_pos, c["sun_disk_pos"], atol=1e-9)
        assert np.allclose(s.eye, cam["eye"], atol=1e-12) and s.sun_radius == pytest.approx(c["sun_disk_radius"], rel=1e-13)
    assert 2 <= parked <= 10


# ---------------------------------------------------------------------------------------------------- a8 / a9
class _Backend:
    """Records what the facade sends down to the C ABI wrapper (MoonRT's method names)."""

    def __init__(self, w, h):
        self.width, self.height, self.rank, self.world = w, h, 0, 1
        self.calls = []

    def __getattr__(self, name):
        def rec(*a, **k):
            self.calls.append((name, a, k))
            if name == "render":
                return {"kernel_ms": 0.0}
            if name == "read_rgba8":
                return np.zeros((self.height, self.width, 4), np.uint8)
        return rec

    def last(self, name):
        return [c for c in self.calls if c[0] == name][-1]


def _revive(x):
    if isinstance(x, dict) and "ndarray" in x:
        if x["data"] is not None:
            return np.array(x["data"], x["dtype"]).reshape(x["ndarray"])
        assert x["sum"] == 0.0
        return np.zeros(x["ndarray"], x["dtype"])
    if isinstance(x, dict):
        return {k: _revive(v) for k, v in x.items()}
    if isinstance(x, list):
        return [_revive(v) for v in x]
    return x


def _replay(rt, calls):
    for name, kw in calls:
        kw = _revive(dict(kw))
        args = kw.pop("args")
        getattr(rt, name)(*args, **kw)


def test_the_references_init_renderer_calls_drive_the_facade():
    """Every `self.rt.*` call MoonRenderer.init_renderer made (moon_renderer.py:570-650, recorded while the reference
    ran) is replayed on the facade: it accepts each one as sent and ends in the state those calls describe."""
    from moonrtx_amd.tkoptix import TkOptiX
    g = _load("host_scene.json")
    calls = g["init_renderer_calls"]
    assert calls[0][0] == "TkOptiX" and [c[0] for c in calls].count("set_data") == 2
    be = _Backend(calls[0][1]["width"], calls[0][1]["height"])
    rt = TkOptiX(width=be.width, height=be.height, on_launch_finished=lambda r: None, backend=be)
    _replay(rt, calls[1:])
    assert be.last("upload_dem")[1][0].shape == (4, 8) and be.last("upload_color")[1][0].shape == (4, 8, 4)
    assert be.last("upload_background")[1] == (None,)
    a = be.last("set_moon_frame")[1]
    assert np.allclose(a[0], 0) and a[1] == 10.0 and np.allclose(a[2], (0, 0, 1)) and np.allclose(a[3], (0, -1, 0))
    a = be.last("set_camera")[1]
    assert np.allclose(a[0], (0, -300, 0)) and np.allclose(a[1], 0) and np.allclose(a[2], (0, 0, 1)) and a[3] == 4.2422
    a = be.last("set_light")[1]
    assert a[1] == 100 and a[2] == pytest.approx(80 * g["constants"]["SUN_BRIGHTNESS_SCALE"])
    a = be.last("set_sun_disk")[1]
    assert np.allclose(a[0], (0, 3100, 0)) and a[1] == 0.01 and a[2] == 2.0
    rt.render_cycle()
    p = be.last("set_params")[2]
    assert (p["path_seg_min"], p["path_seg_max"]) == (2, 4) and p["spp_per_launch"] == 64
    assert p["scene_epsilon"] == 1e-4 and p["marching_step"] == 5e-3 and p["marching_step_eps"] == 3e-4
    assert p["tonemap_exposure"] == 0.9 and p["tonemap_gamma"] == 2.2
    rt.close()


def test_the_references_update_view_calls_drive_the_facade():
    """What MoonRenderer.update_view pushed through `self.rt` for twelve ephemerides (moon_renderer.py:824-871, recorded
    while the reference ran), replayed on the facade: moon axes, Sun disk, light and the camera move reach the backend
    with the reference's numbers; `scene.moon_axes / light_radius` give the same."""
    from moonrtx_amd.tkoptix import TkOptiX
    g = _load("host_scene.json")
    moved = 0
    for c in g["cases"]:
        be = _Backend(64, 32)
        rt = TkOptiX(width=64, height=32, on_launch_finished=lambda r: None, backend=be)
        _replay(rt, g["init_renderer_calls"][1:])
        names = [n for n, _ in c["update_view_calls"]]
        assert names[-4:] == ["update_data", "update_data", "update_light", "refresh_scene"]
        with rt._padlock:
            _replay(rt, c["update_view_calls"])
        R = np.array(c["rotation_matrix"])
        u, v = sc.moon_axes(R)
        a = be.last("set_moon_frame")[1]
        assert np.allclose(a[2], u, atol=0) and np.allclose(a[3], v, atol=0)
        a = be.last("set_sun_disk")[1]
        assert np.allclose(a[0], c["sun_disk_pos"], atol=0) and a[1] == c["sun_disk_radius"]
        a = be.last("set_light")[1]
        assert np.allclose(a[0], c["light_pos"], atol=0) and a[1] == pytest.approx(sc.light_radius(c["sun_distance"]), rel=1e-15)
        if "update_camera" in names:
            moved += 1
            eye = be.last("set_camera")[1][0]
            assert np.allclose(eye, [0, -c["camera_distance"], 0], atol=1e-9)      # follows the Moon's apparent size
        rt.close()
    assert moved >= 9
