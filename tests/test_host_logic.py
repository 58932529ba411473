"""Known answers the reference states in its comments for the scene maths that
sits right above the renderer boundary, checked on the headless restatement in moonrtx_amd/scene.py."""
import math

import numpy as np

from moonrtx_amd import scene as sc


def test_default_fov_and_camera_distance_band():
    # moon_renderer.py:513-519: fov = 2 atan((2*10/0.9)/(2*300)) = 4.2422 deg
    assert abs(sc.default_vfov_deg() - 4.2422) < 1e-4
    # moon_renderer.py:536-541: 27.3 radii closest, 32.2 most distant, 30 at 384400 km
    assert abs(sc.camera_distance(384_400.0) - 300.0) < 1e-9
    assert abs(sc.camera_distance(356_500.0 - 6378.0) / 10 - 27.3) < 0.06
    assert abs(sc.camera_distance(406_700.0 + 6000.0) / 10 - 32.2) < 0.06


def test_visible_cap_at_30_radii():
    # moon_renderer.py:44-48: 88.1 deg at 30 radii, 84.3 at 10 radii
    assert abs(math.degrees(math.acos(1 / 30.0)) - 88.1) < 0.05
    assert abs(math.degrees(math.acos(1 / 10.0)) - 84.3) < 0.05


def test_light_geometry():
    # moon_renderer.py:65-71: asin(100/21460) = 0.267 deg; terminator parallax asin(10/21460) = 0.027 deg
    assert abs(math.degrees(math.asin(100 / sc.SUN_LIGHT_DISTANCE)) - 0.267) < 5e-4
    assert abs(math.degrees(math.asin(10 / sc.SUN_LIGHT_DISTANCE)) - 0.027) < 5e-4
    # moon_renderer.py:683-690: 0 -> above (+Z), 90 -> left (-X), -90 -> right (+X), 180 -> below
    for beta, axis, sign in [(0, 2, 1), (90, 0, -1), (-90, 0, 1), (180, 2, -1)]:
        p = sc.light_position(90.0, beta)
        assert abs(p[axis] - sign * sc.SUN_LIGHT_DISTANCE) < 1e-6 and abs(p[1]) < 1e-6
    assert np.allclose(sc.light_position(0.0, 33.0), (0, -sc.SUN_LIGHT_DISTANCE, 0), atol=1e-9)   # full moon
    assert np.allclose(sc.light_position(180.0, 33.0), (0, sc.SUN_LIGHT_DISTANCE, 0), atol=1e-5)  # new moon
    assert abs(sc.light_radius(1.496e8) - 99.80) < 0.01


def test_brightness_calibration_constant():
    # moon_renderer.py:77-83 + SURVEY 8(a): radiance * solid angle / pi == brightness / 100
    L = sc.light_radiance(100.0)
    omega = math.pi * (100.0 / sc.SUN_LIGHT_DISTANCE) ** 2
    assert abs(L * omega / math.pi - 1.0) < 1e-3


def test_sun_disk_parks_beyond_90_degrees_and_scales_in_view():
    c, r = sc.sun_disk(384_400.0, 1.496e8, 103.0, -70.0)
    assert r == sc.SUN_DISK_PARKED_RADIUS
    c, r = sc.sun_disk(384_400.0, 1.496e8, 0.0, 0.0)       # central eclipse geometry
    assert np.allclose(c, (0.0, -300.0 + 3100.0, 0.0), atol=1e-9)
    mag = math.asin(10 / 300.0) / math.asin(1737.4 / 384_400.0)
    assert abs(r - 3100.0 * math.tan(mag * math.asin(695_700.0 / 1.496e8))) < 1e-9
    # moon_renderer.py:109-111: radiance >= 1.12 is white for every gamma in [0.5, 5]
    for g in (0.5, 2.2, 5.0):
        assert (0.9 * 1.12) ** (1 / g) >= 1.0


def test_moon_axes_and_libration():
    R = sc.libration_rotation(3.0, -5.0)
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-14) and abs(np.linalg.det(R) - 1) < 1e-14
    # the sub-observer point faces the camera (scene -Y)
    assert np.allclose(R @ sc.body_point(-5.0, 3.0, 1.0), (0, -1, 0), atol=1e-14)
    u, v = sc.moon_axes(R)
    assert np.allclose(u, R @ (0, 0, 1)) and np.allclose(v, R @ (0, -1, 0))
    u0, v0 = sc.moon_axes(np.eye(3))
    assert tuple(u0) == (0, 0, 1) and tuple(v0) == (0, -1, 0)      # moon_renderer.py:621


def test_helper_scripts_parse():
    """tools/ is not exercised by the suites: at least keep every script syntactically valid."""
    import glob, os, py_compile, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    scripts = sorted(glob.glob(os.path.join(root, "tools", "*.py"))) + [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")]
    assert len(scripts) > 10
    for path in scripts:
        py_compile.compile(path, doraise=True)
    for path in sorted(glob.glob(os.path.join(root, "tools", "*.sh"))):
        assert subprocess.run(["bash", "-n", path]).returncode == 0, path


def test_zoomed_scene_looks_at_the_terminator():
    """bench.py's close-up: the camera target lies on the terminator (normal perpendicular to the light) on the near side."""
    import numpy as np
    from moonrtx_amd import scene as sc
    for name in ("S1", "S2", "S3"):
        s = sc.zoomed_on_terminator(name, 640, 360)
        n = np.asarray(s.target) / sc.MOON_RADIUS
        light = np.asarray(s.light_pos) / np.linalg.norm(s.light_pos)
        assert abs(np.linalg.norm(n) - 1.0) < 1e-12 and abs(n @ light) < 1e-12
        assert n @ (np.asarray(s.eye) / np.linalg.norm(s.eye)) > 0.0 and s.vfov_deg == 0.7
