"""moonrtx_amd.ephemeris against the worked examples of Meeus, *Astronomical
Algorithms* 2nd ed. -- the only pins available offline (no Skyfield, no JPL kernels: parity with astro.py unpinned) --
plus eclipse geometry and an independent construction of the view rotation through the reference's own matrix
composition (astro.py:116-139, restated here)."""
import math
from datetime import datetime, timedelta, timezone

import numpy as np
import pytest

from moonrtx_amd import ephemeris as E
from moonrtx_amd.scene import selenographic, moon_axes

DEG = math.pi / 180.0
T_47A = (2448724.5 - 2451545.0) / 36525.0      # 1992 April 12, 0h TD


def test_example_47a_moon_position():
    Lp, D, M, Mp, F, Ecc = E.moon_arguments(T_47A)
    assert (Lp, D, M, Mp, F) == pytest.approx((134.290182, 113.842304, 97.643514, 5.150833, 219.889721), abs=2e-6)
    assert Ecc == pytest.approx(1.000194, abs=1e-6)
    sl, sb, sr = E.moon_sums(T_47A)
    assert (round(sl), round(sb), round(sr)) == (-1127527, -3229126, -16590875)     # the book's sums, to the unit
    lam, beta, dist = E.moon_position(T_47A)
    assert lam == pytest.approx(133.162655, abs=1e-6)
    assert beta == pytest.approx(-3.229126, abs=1e-6)
    assert dist == pytest.approx(368409.7, abs=0.05)
    dpsi, _, eps, _ = E.nutation(T_47A)
    assert dpsi == pytest.approx(0.004610, abs=5e-5)         # four-term nutation: 0.1" class
    assert eps == pytest.approx(23.440636, abs=5e-6)
    ra, dec = E.ecl_to_equ(lam + dpsi, beta, eps)
    assert ra == pytest.approx(134.688470, abs=5e-5)
    assert dec == pytest.approx(13.768368, abs=2e-5)


def test_example_25a_sun():
    T = (2448908.5 - 2451545.0) / 36525.0                    # 1992 October 13, 0h TD
    lam, _, R, true_lon = E.sun_position(T)
    assert true_lon == pytest.approx(199.90988, abs=2e-5)
    assert lam == pytest.approx(199.90895, abs=2e-5)
    assert R == pytest.approx(0.99766, abs=1e-5)
    _, _, eps, _ = E.nutation(T)
    ra, dec = E.ecl_to_equ(lam, 0.0, eps)
    assert ra == pytest.approx(198.38083, abs=2e-4)
    assert dec == pytest.approx(-7.78507, abs=1e-4)


def test_example_53a_librations_and_axis():
    dpsi, _, eps, Om = E.nutation(T_47A)
    lam, beta, _ = E.moon_position(T_47A)
    l, b, p = E.libration(lam + dpsi, beta, T_47A, dpsi, Om)
    assert p["lp"] == pytest.approx(-1.206, abs=6e-4)
    assert p["bp"] == pytest.approx(4.194, abs=6e-4)
    assert p["lpp"] == pytest.approx(-0.025, abs=6e-4)
    assert p["bpp"] == pytest.approx(0.006, abs=6e-4)
    assert l == pytest.approx(-1.23, abs=6e-3) and b == pytest.approx(4.20, abs=6e-3)
    ra, _ = E.ecl_to_equ(lam + dpsi, beta, eps)
    assert E.axis_position_angle(ra, b, T_47A, dpsi, eps, Om, p["rho"], p["sigma"]) == pytest.approx(15.08, abs=6e-3)


def test_example_48a_bright_limb():
    assert E.bright_limb_position_angle(134.6885, 13.7684, 20.6579, 8.6964) == pytest.approx(285.0, abs=0.06)


def test_julian_day_and_time_scales():
    assert E.julian_day(datetime(1957, 10, 4, 19, 26, 24)) == pytest.approx(2436116.31, abs=1e-6)     # Meeus 7.a
    assert E.julian_day(datetime(2000, 1, 1, 12, tzinfo=timezone.utc)) == 2451545.0
    assert E.tt_minus_utc(datetime(2024, 4, 8)) == pytest.approx(69.184)
    assert E.tt_minus_utc(datetime(1992, 4, 12)) == pytest.approx(58.184)


def _Rx(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[1, 0, 0], [0, c, -s], [0, s, c]])


def _Rz(a):
    c, s = math.cos(a), math.sin(a)
    return np.array([[c, -s, 0], [s, c, 0], [0, 0, 1.0]])


@pytest.mark.parametrize("jde,q", [(2448724.5, 0.0), (2460409.26, 17.0), (2455000.3, -33.0), (2462000.9, 141.0)])
def test_view_rotation_matches_reference_composition(jde, q):
    """Body frame built geometrically from Cassini's laws (equator inclined I to the ecliptic, its descending node on
    the orbit's ascending node, prime meridian at mean argument of latitude F + 180 deg), pushed through the
    reference's matrix composition, equals view_rotation(l', b', P - q) built from Meeus' closed forms."""
    T = (jde - 2451545.0) / 36525.0
    dpsi, _, eps, Om = E.nutation(T)
    lam, beta, _ = E.moon_position(T)
    lam += dpsi
    F = E.moon_arguments(T)[4]
    ra, dec = E.ecl_to_equ(lam, beta, eps)
    lp, bp, _ = E.optical_libration(lam, beta, dpsi, Om, F)
    P = E.axis_position_angle(ra, bp, T, dpsi, eps, Om, 0.0, 0.0)
    body_to_date = _Rx(eps * DEG) @ _Rz((Om + dpsi) * DEG) @ _Rx(E.MOON_INCLINATION_DEG * DEG).T @ _Rz((F + 180.0) * DEG)
    # E.rotation_matrix is the reference's composition, pinned to astro._rotation_matrix's own output by
    # tests/golden/host_astro.json (test_reference_fixtures.py); body_to_date = R_equator @ R_moon.T
    assert np.abs(E.rotation_matrix(body_to_date.T, np.eye(3), ra, dec, q) - E.view_rotation(lp, bp, P - q)).max() < 1e-12


def test_total_solar_eclipse_2024_geometry():
    """Greatest eclipse 2024-04-08 18:17:16 UTC at 25.29 N 104.14 W: the topocentric discs coincide."""
    E.init(E.Observer(25.29, -104.14, 1500))
    e = E.calculate_moon_ephemeris(datetime(2024, 4, 8, 18, 17, 16, tzinfo=timezone.utc), True)
    assert e.elongation < 0.03                       # Sun radius 0.27 deg; series accuracy ~0.01 deg
    assert e.phase_angle > 179.9 and e.phase_name == "New Moon"
    assert 29.0 < e.age_days < 29.6                  # conjunction in longitude follows at 18:21 UTC
    assert e.alt > 60.0 and 350000 < e.distance < 360000
    # geocentric observer: the Moon passes ~0.35 deg north of the Sun -- parallax is what makes the eclipse central
    g = E.calculate_moon_ephemeris(datetime(2024, 4, 8, 18, 17, 16, tzinfo=timezone.utc), True, E.Observer(90.0, 0.0, 0))
    assert g.elongation > 0.2


def test_lunar_eclipse_2025_and_phase_cycle():
    E.init(E.Observer(0.0, 0.0, 0))
    e = E.calculate_moon_ephemeris(datetime(2025, 3, 14, 6, 58, 47, tzinfo=timezone.utc), True)
    assert e.phase_angle < 1.5 and e.elongation > 178.5 and e.phase_name == "Full Moon"
    assert 13.5 < e.age_days < 15.5
    names = [E.calculate_moon_ephemeris(datetime(2025, 3, 1, tzinfo=timezone.utc) + timedelta(days=d), True).phase_name
             for d in range(0, 30, 3)]
    assert names[0] == "Waxing Crescent" and "Waxing Gibbous" in names and "Waning Crescent" in names


def test_ephemeris_fields_are_self_consistent():
    E.init(E.Observer(52.2, 21.0, 100))
    for k in range(12):
        dt = datetime(2026, 1, 3, 17, 30, tzinfo=timezone(timedelta(hours=1))) + timedelta(days=2.4 * k)
        for mode in (True, False):
            e = E.calculate_moon_ephemeris(dt, mode)
            R = np.asarray(e.rotation_matrix)
            assert abs(np.linalg.det(R) - 1.0) < 1e-12 and np.abs(R @ R.T - np.eye(3)).max() < 1e-12
            # the body point facing the camera (scene -Y) is the topocentric sub-observer point
            lat, lon = selenographic(np.array([0.0, -10.0, 0.0]), R)
            assert lat == pytest.approx(e.libr_lat_topo, abs=1e-9) and lon == pytest.approx(e.libr_long_topo, abs=1e-9)
            assert abs(e.libr_long_geo) < 8.5 and abs(e.libr_lat_geo) < 7.0
            assert abs(e.libr_long_topo - e.libr_long_geo) < 1.1 and abs(e.libr_lat_topo - e.libr_lat_geo) < 1.1
            assert e.phase_angle + e.elongation == pytest.approx(180.0, abs=0.2)      # Sun 390x farther than the Moon
            assert 350000 < e.distance < 413500 and 1.46e8 < e.sun_distance < 1.53e8
            assert e.colongitude == pytest.approx((90.0 - e.subsolar_lon) % 360.0, abs=1e-9) and abs(e.subsolar_lat) < 1.7
            # Sun over the sub-observer point <-> phase angle
            alt = E.sun_altitude_at(e.subsolar_lat, e.subsolar_lon, e.libr_lat_topo, e.libr_long_topo)
            assert 90.0 - alt == pytest.approx(e.phase_angle, abs=0.3)
            # the lit limb points at the Sun: light position of the scene (moon_renderer.py:676-727) vs subsolar point
            u, v = moon_axes(R)
            assert np.linalg.norm(u) == pytest.approx(1.0) and np.dot(u, v) == pytest.approx(0.0, abs=1e-12)
        a = E.calculate_moon_ephemeris(dt, True)
        b = E.calculate_moon_ephemeris(dt, False)
        # non-parallactic mode rolls the view by the parallactic angle and shifts the bright-limb angle by the same
        q = E.wrap_signed_degrees(a.bright_limb_angle - b.bright_limb_angle)
        pa = lambda R: math.degrees(math.atan2(-(R @ [0, 0, 1.0])[0], (R @ [0, 0, 1.0])[2]))
        assert E.wrap_signed_degrees(pa(a.rotation_matrix) - pa(b.rotation_matrix)) == pytest.approx(q, abs=1e-6)


def test_scene_light_agrees_with_subsolar_point():
    """The light position built from (phase angle, bright-limb angle) -- moon_renderer.py:676-727 -- and the subsolar
    point rotated into the scene by the ephemeris' rotation matrix are the same direction (to the series' accuracy)."""
    from moonrtx_amd.scene import body_point
    E.init(E.Observer(-33.9, 18.4, 10))
    for k in range(8):
        dt = datetime(2025, 6, 2, 3, 0, tzinfo=timezone.utc) + timedelta(days=3.7 * k)
        e = E.calculate_moon_ephemeris(dt, False)
        s = E.scene_from_ephemeris(e, 64, 64)
        light = np.asarray(s.light_pos, float)
        light /= np.linalg.norm(light)
        sub = np.asarray(e.rotation_matrix) @ body_point(e.subsolar_lat, e.subsolar_lon, 1.0)
        assert math.degrees(math.acos(float(np.clip(np.dot(light, sub), -1, 1)))) < 0.3


def test_errors():
    E._observer = None
    with pytest.raises(RuntimeError):
        E.calculate_moon_ephemeris(datetime(2025, 1, 1, tzinfo=timezone.utc), True)
    E.init(E.Observer(0, 0, 0))
    with pytest.raises(ValueError):
        E.calculate_moon_ephemeris(datetime(2025, 1, 1), True)
