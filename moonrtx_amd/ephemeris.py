"""Self-contained lunar ephemeris: the step before scene.make_scene.

Same surface as the reference's astro.py -- `init(observer)`, `calculate_moon_ephemeris(dt_local, parallactic_mode)`
-> `MoonEphemeris` with the same 19 fields (astro.py:28-46, :662-745; shared_types.py:23-42) -- but no Skyfield and
no JPL kernels (neither exists offline): the analytic series of J. Meeus, *Astronomical Algorithms* (2nd ed.):
Moon ch. 47 (ELP-2000/82 truncation, ~10" in longitude), Sun ch. 25, nutation/obliquity ch. 22, sidereal time
ch. 12, observer ch. 11, phase and bright limb ch. 48, optical + physical librations and the axis position angle
ch. 53.  Topocentric quantities come from subtracting the observer's geocentric vector (equator of date).

PARITY UNPINNED against Skyfield/DE421 (no reference fixture exists); pinned instead by
the book's worked examples 25.a, 47.a, 48.a, 53.a (tests/test_ephemeris.py) and by an independent geometric
construction of the optical libration from Cassini's laws.  Expected differences from the reference: Moon ~0.003 deg,
Sun ~0.01 deg, librations ~0.02 deg (mean-Earth axes here, principal axes there) -- a fraction of a pixel at 4K.
"""
import math
from datetime import datetime, timedelta, timezone
from typing import NamedTuple

import numpy as np

from .scene import libration_rotation

DEG = math.pi / 180.0
EARTH_RADIUS_KM = 6378.14
AU_KM = 149597870.7
MOON_INCLINATION_DEG = 1.54242          # I, Meeus ch. 53


class Observer(NamedTuple):             # shared_types.py:92-95
    lat: float
    lon: float
    elevation_m: float = 0.0


class MoonEphemeris(NamedTuple):        # shared_types.py:23-42, same order
    az: float
    alt: float
    ra: float
    dec: float
    distance: float
    sun_distance: float
    phase_angle: float
    age_days: float
    bright_limb_angle: float
    libr_long_geo: float
    libr_lat_geo: float
    libr_long_topo: float
    libr_lat_topo: float
    elongation: float
    phase_name: str
    colongitude: float
    subsolar_lat: float
    subsolar_lon: float
    rotation_matrix: np.ndarray


# ---- time scales ---------------------------------------------------------------------------------------------------
# TAI-UTC steps (IERS Bulletin C); TT = TAI + 32.184 s
_LEAP = [(1972, 1, 10), (1972, 7, 11), (1973, 1, 12), (1974, 1, 13), (1975, 1, 14), (1976, 1, 15), (1977, 1, 16),
         (1978, 1, 17), (1979, 1, 18), (1980, 1, 19), (1981, 7, 20), (1982, 7, 21), (1983, 7, 22), (1985, 7, 23),
         (1988, 1, 24), (1990, 1, 25), (1991, 1, 26), (1992, 7, 27), (1993, 7, 28), (1994, 7, 29), (1996, 1, 30),
         (1997, 7, 31), (1999, 1, 32), (2006, 1, 33), (2009, 1, 34), (2012, 7, 35), (2015, 7, 36), (2017, 1, 37)]


def tt_minus_utc(dt_utc):
    s = 10
    for y, m, v in _LEAP:
        if (dt_utc.year, dt_utc.month) >= (y, m):
            s = v
    return s + 32.184


def julian_day(dt_utc):
    """JD (UTC) of an aware or naive-UTC datetime -- Meeus ch. 7."""
    if dt_utc.tzinfo is not None:
        dt_utc = dt_utc.astimezone(timezone.utc).replace(tzinfo=None)
    y, m = dt_utc.year, dt_utc.month
    d = dt_utc.day + (dt_utc.hour + (dt_utc.minute + (dt_utc.second + dt_utc.microsecond * 1e-6) / 60.0) / 60.0) / 24.0
    if m <= 2:
        y -= 1
        m += 12
    a = y // 100
    b = 2 - a + a // 4
    return math.floor(365.25 * (y + 4716)) + math.floor(30.6001 * (m + 1)) + d + b - 1524.5


def _norm360(x):
    return x % 360.0


def wrap_signed_degrees(a):             # astro.py:84-85
    return (a + 180.0) % 360.0 - 180.0


def colongitude_from_subsolar_longitude(lon):   # astro.py:88-89
    return (90.0 - wrap_signed_degrees(lon)) % 360.0


def parallactic_angle_deg(hour_angle_deg, dec_deg, lat_deg):   # astro.py:96-103
    h, d, p = hour_angle_deg * DEG, dec_deg * DEG, lat_deg * DEG
    return math.degrees(math.atan2(math.sin(h), math.tan(p) * math.cos(d) - math.sin(d) * math.cos(h)))


def body_altitude_at_feature(sub_lat_deg, sub_lon_deg, lat_deg, lon_deg):   # astro.py:166-184
    b0, l0 = np.radians(sub_lat_deg), np.radians(sub_lon_deg)
    b, l = math.radians(lat_deg), math.radians(lon_deg)
    s = np.sin(b0) * math.sin(b) + np.cos(b0) * math.cos(b) * np.cos(l - l0)
    return np.degrees(np.arcsin(np.clip(s, -1.0, 1.0)))


def sun_altitude_at(subsolar_lat, subsolar_lon, lat_deg, lon_deg):          # astro.py:650-659
    return float(body_altitude_at_feature(np.asarray(subsolar_lat), np.asarray(subsolar_lon), lat_deg, lon_deg))


RENDERER_TO_BODY = np.array([[0.0, -1.0, 0.0], [1.0, 0.0, 0.0], [0.0, 0.0, 1.0]])   # astro.py:20-25


def latlon_from_icrf(pos, R_icrf_to_body):      # astro.py:106-113
    """ICRF position vector -> body-frame (lat_deg, lon_deg)."""
    b = np.asarray(R_icrf_to_body, float) @ np.asarray(pos, float)
    return math.degrees(math.asin(b[2] / np.linalg.norm(b))), math.degrees(math.atan2(b[1], b[0]))


def rotation_matrix(R_moon, R_equator, moon_ra_deg, moon_dec_deg, q_deg):      # astro.py:116-139
    """Renderer-body -> view rotation from the body frame (ICRF -> Moon-fixed `R_moon`), the equator-of-date frame
    `R_equator`, the Moon's apparent (RA, Dec) of date and the roll `q_deg` of 'up' from celestial north towards east:
    rows of the view basis = (right, line of sight, up).  Pinned by tests/golden/host_astro.json."""
    ra, dec, q = moon_ra_deg * DEG, moon_dec_deg * DEG, q_deg * DEG
    sight = np.array([math.cos(dec) * math.cos(ra), math.cos(dec) * math.sin(ra), math.sin(dec)])
    east = np.array([-math.sin(ra), math.cos(ra), 0.0])
    north = np.array([-math.sin(dec) * math.cos(ra), -math.sin(dec) * math.sin(ra), math.cos(dec)])
    up = math.sin(q) * east + math.cos(q) * north
    up = up / np.linalg.norm(up)
    right = np.cross(sight, up)
    right = right / np.linalg.norm(right)
    return np.vstack([right, sight, up]) @ (np.asarray(R_equator, float) @ np.asarray(R_moon, float).T) @ RENDERER_TO_BODY


# ---- Meeus ch. 22: nutation and obliquity (the four largest terms: 0.5" / 0.1") -------------------------------------
def nutation(T):
    """(dpsi_deg, deps_deg, eps_true_deg, Omega_deg)."""
    om = 125.04452 - 1934.136261 * T
    ls = 280.4665 + 36000.7698 * T
    lm = 218.3165 + 481267.8813 * T
    dpsi = (-17.20 * math.sin(om * DEG) - 1.32 * math.sin(2 * ls * DEG) - 0.23 * math.sin(2 * lm * DEG)
            + 0.21 * math.sin(2 * om * DEG)) / 3600.0
    deps = (9.20 * math.cos(om * DEG) + 0.57 * math.cos(2 * ls * DEG) + 0.10 * math.cos(2 * lm * DEG)
            - 0.09 * math.cos(2 * om * DEG)) / 3600.0
    eps0 = 23.0 + 26.0 / 60.0 + (21.448 - 46.8150 * T - 0.00059 * T * T + 0.001813 * T ** 3) / 3600.0
    return dpsi, deps, eps0 + deps, _norm360(om)


# ---- Meeus ch. 25: the Sun (0.01 deg) -------------------------------------------------------------------------------
def sun_position(T):
    """Geocentric (apparent ecliptic longitude deg, latitude 0, distance AU, true longitude deg)."""
    L0 = 280.46646 + 36000.76983 * T + 0.0003032 * T * T
    M = 357.52911 + 35999.05029 * T - 0.0001537 * T * T
    e = 0.016708634 - 0.000042037 * T - 0.0000001267 * T * T
    C = ((1.914602 - 0.004817 * T - 0.000014 * T * T) * math.sin(M * DEG)
         + (0.019993 - 0.000101 * T) * math.sin(2 * M * DEG) + 0.000289 * math.sin(3 * M * DEG))
    true_lon = L0 + C
    nu = M + C
    R = 1.000001018 * (1 - e * e) / (1 + e * math.cos(nu * DEG))
    om = 125.04 - 1934.136 * T
    lam = true_lon - 0.00569 - 0.00478 * math.sin(om * DEG)
    return _norm360(lam), 0.0, R, _norm360(true_lon)


# ---- Meeus ch. 47: the Moon -----------------------------------------------------------------------------------------
# (D, M, M', F, sum_l [1e-6 deg], sum_r [1e-3 km]) -- table 47.A
_LR = [(0, 0, 1, 0, 6288774, -20905355), (2, 0, -1, 0, 1274027, -3699111), (2, 0, 0, 0, 658314, -2955968),
       (0, 0, 2, 0, 213618, -569925), (0, 1, 0, 0, -185116, 48888), (0, 0, 0, 2, -114332, -3149),
       (2, 0, -2, 0, 58793, 246158), (2, -1, -1, 0, 57066, -152138), (2, 0, 1, 0, 53322, -170733),
       (2, -1, 0, 0, 45758, -204586), (0, 1, -1, 0, -40923, -129620), (1, 0, 0, 0, -34720, 108743),
       (0, 1, 1, 0, -30383, 104755), (2, 0, 0, -2, 15327, 10321), (0, 0, 1, 2, -12528, 0),
       (0, 0, 1, -2, 10980, 79661), (4, 0, -1, 0, 10675, -34782), (0, 0, 3, 0, 10034, -23210),
       (4, 0, -2, 0, 8548, -21636), (2, 1, -1, 0, -7888, 24208), (2, 1, 0, 0, -6766, 30824),
       (1, 0, -1, 0, -5163, -8379), (1, 1, 0, 0, 4987, -16675), (2, -1, 1, 0, 4036, -12831),
       (2, 0, 2, 0, 3994, -10445), (4, 0, 0, 0, 3861, -11650), (2, 0, -3, 0, 3665, 14403),
       (0, 1, -2, 0, -2689, -7003), (2, 0, -1, 2, -2602, 0), (2, -1, -2, 0, 2390, 10056),
       (1, 0, 1, 0, -2348, 6322), (2, -2, 0, 0, 2236, -9884), (0, 1, 2, 0, -2120, 5751),
       (0, 2, 0, 0, -2069, 0), (2, -2, -1, 0, 2048, -4950), (2, 0, 1, -2, -1773, 4130),
       (2, 0, 0, 2, -1595, 0), (4, -1, -1, 0, 1215, -3958), (0, 0, 2, 2, -1110, 0),
       (3, 0, -1, 0, -892, 3258), (2, 1, 1, 0, -810, 2616), (4, -1, -2, 0, 759, -1897),
       (0, 2, -1, 0, -713, -2117), (2, 2, -1, 0, -700, 2354), (2, 1, -2, 0, 691, 0),
       (2, -1, 0, -2, 596, 0), (4, 0, 1, 0, 549, -1423), (0, 0, 4, 0, 537, -1117),
       (4, -1, 0, 0, 520, -1571), (1, 0, -2, 0, -487, -1739), (2, 1, 0, -2, -399, 0),
       (0, 0, 2, -2, -381, -4421), (1, 1, 1, 0, 351, 0), (3, 0, -2, 0, -340, 0),
       (4, 0, -3, 0, 330, 0), (2, -1, 2, 0, 327, 0), (0, 2, 1, 0, -323, 1165),
       (1, 1, -1, 0, 299, 0), (2, 0, 3, 0, 294, 0), (2, 0, -1, -2, 0, 8752)]
# (D, M, M', F, sum_b [1e-6 deg]) -- table 47.B
_B = [(0, 0, 0, 1, 5128122), (0, 0, 1, 1, 280602), (0, 0, 1, -1, 277693), (2, 0, 0, -1, 173237),
      (2, 0, -1, 1, 55413), (2, 0, -1, -1, 46271), (2, 0, 0, 1, 32573), (0, 0, 2, 1, 17198),
      (2, 0, 1, -1, 9266), (0, 0, 2, -1, 8822), (2, -1, 0, -1, 8216), (2, 0, -2, -1, 4324),
      (2, 0, 1, 1, 4200), (2, 1, 0, -1, -3359), (2, -1, -1, 1, 2463), (2, -1, 0, 1, 2211),
      (2, -1, -1, -1, 2065), (0, 1, -1, -1, -1870), (4, 0, -1, -1, 1828), (0, 1, 0, 1, -1794),
      (0, 0, 0, 3, -1749), (0, 1, -1, 1, -1565), (1, 0, 0, 1, -1491), (0, 1, 1, 1, -1475),
      (0, 1, 1, -1, -1410), (0, 1, 0, -1, -1344), (1, 0, 0, -1, -1335), (0, 0, 3, 1, 1107),
      (4, 0, 0, -1, 1021), (4, 0, -1, 1, 833), (0, 0, 1, -3, 777), (4, 0, -2, 1, 671),
      (2, 0, 0, -3, 607), (2, 0, 2, -1, 596), (2, -1, 1, -1, 491), (2, 0, -2, 1, -451),
      (0, 0, 3, -1, 439), (2, 0, 2, 1, 422), (2, 0, -3, -1, 421), (2, 1, -1, 1, -366),
      (2, 1, 0, 1, -351), (4, 0, 0, 1, 331), (2, -1, 1, 1, 315), (2, -2, 0, -1, 302),
      (0, 0, 1, 3, -283), (2, 1, 1, -1, -229), (1, 1, 0, -1, 223), (1, 1, 0, 1, 223),
      (0, 1, -2, -1, -220), (2, 1, -1, -1, -220), (1, 0, 1, 1, -185), (2, -1, -2, -1, 181),
      (0, 1, 2, 1, -177), (4, 0, -2, -1, 176), (4, -1, -1, -1, 166), (1, 0, 1, -1, -164),
      (4, 0, 1, -1, 132), (1, 0, -1, -1, -119), (4, -1, 0, -1, 115), (2, -2, 0, 1, 107)]


def moon_arguments(T):
    """Mean elements (deg): L', D, M, M', F, and E -- Meeus 47.1-47.6."""
    Lp = 218.3164477 + 481267.88123421 * T - 0.0015786 * T * T + T ** 3 / 538841.0 - T ** 4 / 65194000.0
    D = 297.8501921 + 445267.1114034 * T - 0.0018819 * T * T + T ** 3 / 545868.0 - T ** 4 / 113065000.0
    M = 357.5291092 + 35999.0502909 * T - 0.0001536 * T * T + T ** 3 / 24490000.0
    Mp = 134.9633964 + 477198.8675055 * T + 0.0087414 * T * T + T ** 3 / 69699.0 - T ** 4 / 14712000.0
    F = 93.2720950 + 483202.0175233 * T - 0.0036539 * T * T - T ** 3 / 3526000.0 + T ** 4 / 863310000.0
    E = 1.0 - 0.002516 * T - 0.0000074 * T * T
    return _norm360(Lp), _norm360(D), _norm360(M), _norm360(Mp), _norm360(F), E


def moon_sums(T):
    """(sum_l, sum_b, sum_r) in the table units, additive terms included."""
    Lp, D, M, Mp, F, E = moon_arguments(T)
    A1, A2, A3 = 119.75 + 131.849 * T, 53.09 + 479264.290 * T, 313.45 + 481266.484 * T
    sl = sr = sb = 0.0
    for d, m, mp, f, cl, cr in _LR:
        arg = (d * D + m * M + mp * Mp + f * F) * DEG
        e = E ** abs(m)
        sl += cl * e * math.sin(arg)
        sr += cr * e * math.cos(arg)
    for d, m, mp, f, cb in _B:
        arg = (d * D + m * M + mp * Mp + f * F) * DEG
        sb += cb * E ** abs(m) * math.sin(arg)
    sl += 3958.0 * math.sin(A1 * DEG) + 1962.0 * math.sin((Lp - F) * DEG) + 318.0 * math.sin(A2 * DEG)
    sb += (-2235.0 * math.sin(Lp * DEG) + 382.0 * math.sin(A3 * DEG) + 175.0 * math.sin((A1 - F) * DEG)
           + 175.0 * math.sin((A1 + F) * DEG) + 127.0 * math.sin((Lp - Mp) * DEG) - 115.0 * math.sin((Lp + Mp) * DEG))
    return sl, sb, sr


def moon_position(T):
    """Geocentric (ecliptic longitude deg, latitude deg, distance km), mean equinox of date."""
    Lp = moon_arguments(T)[0]
    sl, sb, sr = moon_sums(T)
    return _norm360(Lp + sl * 1e-6), sb * 1e-6, 385000.56 + sr * 1e-3


# ---- frames ---------------------------------------------------------------------------------------------------------
def ecl_to_equ(lam, beta, eps):
    l, b, e = lam * DEG, beta * DEG, eps * DEG
    ra = math.atan2(math.sin(l) * math.cos(e) - math.tan(b) * math.sin(e), math.cos(l))
    dec = math.asin(math.sin(b) * math.cos(e) + math.cos(b) * math.sin(e) * math.sin(l))
    return _norm360(ra / DEG), dec / DEG


def equ_to_ecl(ra, dec, eps):
    a, d, e = ra * DEG, dec * DEG, eps * DEG
    lam = math.atan2(math.sin(a) * math.cos(e) + math.tan(d) * math.sin(e), math.cos(a))
    beta = math.asin(math.sin(d) * math.cos(e) - math.cos(d) * math.sin(e) * math.sin(a))
    return _norm360(lam / DEG), beta / DEG


def _vec(ra, dec, r):
    a, d = ra * DEG, dec * DEG
    return np.array([r * math.cos(d) * math.cos(a), r * math.cos(d) * math.sin(a), r * math.sin(d)])


def _radec(v):
    r = float(np.linalg.norm(v))
    return _norm360(math.degrees(math.atan2(v[1], v[0]))), math.degrees(math.asin(v[2] / r)), r


def sidereal_time_deg(jd_ut, T, dpsi, eps):
    """Apparent sidereal time at Greenwich -- Meeus 12.4 + nutation in right ascension."""
    th = 280.46061837 + 360.98564736629 * (jd_ut - 2451545.0) + 0.000387933 * T * T - T ** 3 / 38710000.0
    return _norm360(th + dpsi * math.cos(eps * DEG))


def observer_vector_km(obs, lst_deg):
    """Geocentric position of the observer, equator of date -- Meeus ch. 11."""
    phi = obs.lat * DEG
    u = math.atan(0.99664719 * math.tan(phi))
    h = obs.elevation_m / 6378140.0
    rs = 0.99664719 * math.sin(u) + h * math.sin(phi)
    rc = math.cos(u) + h * math.cos(phi)
    t = lst_deg * DEG
    return EARTH_RADIUS_KM * np.array([rc * math.cos(t), rc * math.sin(t), rs])


# ---- Meeus ch. 53: librations ----------------------------------------------------------------------------------------
def optical_libration(lam, beta, dpsi, Omega, F):
    """(l', b', A) for apparent longitude `lam` (nutation included), latitude `beta` -- Meeus 53.1."""
    I = MOON_INCLINATION_DEG * DEG
    W = (lam - dpsi - Omega) * DEG
    b = beta * DEG
    A = math.atan2(math.sin(W) * math.cos(b) * math.cos(I) - math.sin(b) * math.sin(I), math.cos(W) * math.cos(b))
    lp = wrap_signed_degrees(A / DEG - F)
    bp = math.degrees(math.asin(-math.sin(W) * math.cos(b) * math.sin(I) - math.sin(b) * math.cos(I)))
    return lp, bp, A


def physical_libration_terms(T, Omega):
    """(rho, sigma, tau) in degrees -- Meeus ch. 53."""
    _, D, M, Mp, F, E = moon_arguments(T)
    K1, K2 = 119.75 + 131.849 * T, 72.56 + 20.186 * T
    s = lambda x: math.sin(x * DEG)
    c = lambda x: math.cos(x * DEG)
    rho = (-0.02752 * c(Mp) - 0.02245 * s(F) + 0.00684 * c(Mp - 2 * F) - 0.00293 * c(2 * F) - 0.00085 * c(2 * F - 2 * D)
           - 0.00054 * c(Mp - 2 * D) - 0.00020 * s(Mp + F) - 0.00020 * c(Mp + 2 * F) - 0.00020 * c(Mp - F)
           + 0.00014 * c(Mp + 2 * F - 2 * D))
    sigma = (-0.02816 * s(Mp) + 0.02244 * c(F) - 0.00682 * s(Mp - 2 * F) - 0.00279 * s(2 * F) - 0.00083 * s(2 * F - 2 * D)
             + 0.00069 * s(Mp - 2 * D) + 0.00040 * c(Mp + F) - 0.00025 * s(2 * Mp) - 0.00023 * s(Mp + 2 * F)
             + 0.00020 * c(Mp - F) + 0.00019 * s(Mp - F) + 0.00013 * s(Mp + 2 * F - 2 * D) - 0.00010 * c(Mp - 3 * F))
    tau = (0.02520 * E * s(M) + 0.00473 * s(2 * Mp - 2 * F) - 0.00467 * s(Mp) + 0.00396 * s(K1) + 0.00276 * s(2 * Mp - 2 * D)
           + 0.00196 * s(Omega) - 0.00183 * c(Mp - F) + 0.00115 * s(Mp - 2 * D) - 0.00096 * s(Mp - D) + 0.00046 * s(2 * F - 2 * D)
           - 0.00039 * s(Mp - F) - 0.00032 * s(Mp - M - D) + 0.00027 * s(2 * Mp - M - 2 * D) + 0.00023 * s(K2)
           - 0.00014 * s(2 * D) + 0.00014 * c(2 * Mp - 2 * F) - 0.00012 * s(Mp - 2 * F) - 0.00012 * s(2 * Mp)
           + 0.00011 * s(2 * Mp - 2 * M - 2 * D))
    return rho, sigma, tau


def libration(lam, beta, T, dpsi, Omega):
    """Total selenographic (longitude l, latitude b) of the point the direction (lam, beta) is seen from, plus
    the pieces (l', b', l'', b'', A, rho, sigma) -- Meeus ch. 53."""
    F = moon_arguments(T)[4]
    lp, bp, A = optical_libration(lam, beta, dpsi, Omega, F)
    rho, sigma, tau = physical_libration_terms(T, Omega)
    lpp = -tau + (rho * math.cos(A) + sigma * math.sin(A)) * math.tan(bp * DEG)
    bpp = sigma * math.cos(A) - rho * math.sin(A)
    return lp + lpp, bp + bpp, dict(lp=lp, bp=bp, lpp=lpp, bpp=bpp, A=A, rho=rho, sigma=sigma)


def axis_position_angle(ra, b_total, T, dpsi, eps, Omega, rho, sigma):
    """Position angle P of the Moon's rotation axis (deg, from celestial north towards east) -- Meeus ch. 53."""
    I = MOON_INCLINATION_DEG
    V = (Omega + dpsi + sigma / math.sin(I * DEG)) * DEG
    X = math.sin((I + rho) * DEG) * math.sin(V)
    Y = math.sin((I + rho) * DEG) * math.cos(V) * math.cos(eps * DEG) - math.cos((I + rho) * DEG) * math.sin(eps * DEG)
    w = math.atan2(X, Y)
    return math.degrees(math.asin(math.sqrt(X * X + Y * Y) * math.cos(ra * DEG - w) / math.cos(b_total * DEG)))


def view_rotation(l_deg, b_deg, pole_angle_deg):
    """Renderer-body -> view rotation (x right, y along the line of sight, z up) for sub-observer point (l, b) and
    the axis at `pole_angle_deg` from 'up' towards celestial east (= to the LEFT on the sky).  Same matrix as
    astro._rotation_matrix builds from the body frame (astro.py:116-139)."""
    c, s = math.cos(-pole_angle_deg * DEG), math.sin(-pole_angle_deg * DEG)
    roll = np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]])     # about +y: the pole (0,0,1) -> (sin t, 0, cos t)
    return roll @ libration_rotation(l_deg, b_deg)


# ---- Meeus ch. 48 -----------------------------------------------------------------------------------------------------
def bright_limb_position_angle(ra, dec, ra_sun, dec_sun):
    a, d, a0, d0 = ra * DEG, dec * DEG, ra_sun * DEG, dec_sun * DEG
    return _norm360(math.degrees(math.atan2(math.cos(d0) * math.sin(a0 - a),
                                            math.sin(d0) * math.cos(d) - math.cos(d0) * math.sin(d) * math.cos(a0 - a))))


def phase_name(moon_lon, sun_lon):      # astro.py:142-163
    delta = (moon_lon - sun_lon) % 360.0
    if delta < 0.5 or delta > 359.5:
        return "New Moon"
    for lim, name in ((89.5, "Waxing Crescent"), (90.5, "First Quarter"), (179.5, "Waxing Gibbous"), (180.5, "Full Moon"),
                      (269.5, "Waning Gibbous"), (270.5, "Last Quarter")):
        if delta < lim:
            return name
    return "Waning Crescent"


def refraction_deg(alt_deg, temperature_c=10.0, pressure_mbar=1010.0):
    """Bennett's formula scaled for temperature and pressure (what Skyfield's altaz("standard") applies), iterated so
    that the argument is the apparent altitude."""
    if alt_deg < -1.0 or alt_deg > 89.9:
        return 0.0
    a = alt_deg
    for _ in range(6):
        r = 0.016667 / math.tan((a + 7.31 / (a + 4.4)) * DEG) * (0.28 * pressure_mbar / (temperature_c + 273.0))
        a = alt_deg + r
    return a - alt_deg


# ---- the astro.py surface ---------------------------------------------------------------------------------------------
_observer = None
_lunation = None


def init(observer):
    """astro.init (astro.py:28-46): remember the observing site."""
    global _observer, _lunation
    _observer = Observer(float(observer.lat), float(observer.lon), float(getattr(observer, "elevation_m", 0.0)))
    _lunation = None


def _elongation_lon(jde):
    T = (jde - 2451545.0) / 36525.0
    return wrap_signed_degrees(moon_position(T)[0] - sun_position(T)[0])


def previous_new_moon(jde):
    """JDE of the last conjunction in longitude at or before `jde` (secant iteration on the series)."""
    t = jde - (_elongation_lon(jde) % 360.0) / 12.190749
    for _ in range(8):
        f = _elongation_lon(t)
        t -= f / 12.190749
        if abs(f) < 1e-7:
            break
    if t > jde:
        t = previous_new_moon(jde - 1.0)
    return t


def moon_age_days(jde):                 # astro.py:49-69
    global _lunation
    if _lunation is None or not (_lunation[0] <= jde < _lunation[0] + 29.0):
        _lunation = (previous_new_moon(jde),)
    return jde - _lunation[0]


def calculate_moon_ephemeris(dt_local, parallactic_mode, observer=None):
    """astro.calculate_moon_ephemeris (astro.py:662-745) from analytic series.  `dt_local` must be timezone-aware."""
    obs = observer if observer is not None else _observer
    if obs is None:
        raise RuntimeError("ephemeris.init(observer) has not been called")
    if dt_local.tzinfo is None:
        raise ValueError("dt_local must be timezone-aware")
    dt_utc = dt_local.astimezone(timezone.utc)
    jd_ut = julian_day(dt_utc)
    jde = jd_ut + tt_minus_utc(dt_utc) / 86400.0
    T = (jde - 2451545.0) / 36525.0

    dpsi, _, eps, Omega = nutation(T)
    lam_m, beta_m, dist_m = moon_position(T)
    lam_m_app = lam_m + dpsi
    lam_s, _, r_s, _ = sun_position(T)
    lam_s_app = lam_s + dpsi
    ra_m, dec_m = ecl_to_equ(lam_m_app, beta_m, eps)
    ra_s, dec_s = ecl_to_equ(lam_s_app, 0.0, eps)

    # topocentric vectors, equator of date
    gst = sidereal_time_deg(jd_ut, T, dpsi, eps)
    lst = _norm360(gst + obs.lon)
    site = observer_vector_km(obs, lst)
    moon_geo, sun_geo = _vec(ra_m, dec_m, dist_m), _vec(ra_s, dec_s, r_s * AU_KM)
    moon_topo, sun_topo = moon_geo - site, sun_geo - site
    ra_t, dec_t, dist_t = _radec(moon_topo)
    ra_st, dec_st, dist_st = _radec(sun_topo)

    hour_angle = wrap_signed_degrees(lst - ra_t)
    q = 0.0 if parallactic_mode else parallactic_angle_deg(hour_angle, dec_t, obs.lat)
    phi, d, h = obs.lat * DEG, dec_t * DEG, hour_angle * DEG
    alt = math.degrees(math.asin(math.sin(phi) * math.sin(d) + math.cos(phi) * math.cos(d) * math.cos(h)))
    az = _norm360(math.degrees(math.atan2(math.sin(h), math.cos(h) * math.sin(phi) - math.tan(d) * math.cos(phi))) + 180.0)
    alt += refraction_deg(alt, 10.0, 1010.0 * math.exp(-obs.elevation_m / 9.1e3))

    um, us = moon_topo / dist_t, sun_topo / dist_st
    elongation = math.degrees(math.atan2(np.linalg.norm(np.cross(um, us)), float(np.dot(um, us))))
    bright_limb = bright_limb_position_angle(ra_t, dec_t, ra_st, dec_st) - q

    # librations: geocentric and topocentric directions through the same formulas (Meeus p. 374, the rigorous way)
    l_geo, b_geo, _ = libration(lam_m_app, beta_m, T, dpsi, Omega)
    lam_t, beta_t = equ_to_ecl(ra_t, dec_t, eps)
    l_top, b_top, parts = libration(lam_t, beta_t, T, dpsi, Omega)
    P = axis_position_angle(ra_t, b_top, T, dpsi, eps, Omega, parts["rho"], parts["sigma"])

    # selenographic position of the Sun (Meeus ch. 53): heliocentric direction of the Moon
    ratio = dist_m / (r_s * AU_KM)
    lam_h = lam_s_app + 180.0 + ratio * 57.296 * math.cos(beta_m * DEG) * math.sin((lam_s_app - lam_m_app) * DEG)
    beta_h = ratio * beta_m
    l_sun, b_sun, _ = libration(lam_h, beta_h, T, dpsi, Omega)

    sun_from_moon, obs_from_moon = sun_geo - moon_geo, site - moon_geo
    phase_angle = math.degrees(math.atan2(np.linalg.norm(np.cross(sun_from_moon, obs_from_moon)),
                                          float(np.dot(sun_from_moon, obs_from_moon))))
    return MoonEphemeris(
        az=az, alt=alt, ra=ra_t, dec=dec_t, distance=dist_t, sun_distance=dist_st, phase_angle=phase_angle,
        age_days=moon_age_days(jde), bright_limb_angle=wrap_signed_degrees(bright_limb),
        libr_long_geo=wrap_signed_degrees(l_geo), libr_lat_geo=b_geo,
        libr_long_topo=wrap_signed_degrees(l_top), libr_lat_topo=b_top, elongation=elongation,
        phase_name=phase_name(lam_m, lam_s), colongitude=colongitude_from_subsolar_longitude(l_sun),
        subsolar_lat=b_sun, subsolar_lon=wrap_signed_degrees(l_sun),
        rotation_matrix=view_rotation(l_top, b_top, P - q))


def scene_from_ephemeris(eph, width, height, **kw):
    """MoonEphemeris -> SceneDesc: what update_view pushes for one date (moon_renderer.py:824-871)."""
    from .scene import make_scene, moon_axes
    s = make_scene(width, height, eph.phase_angle, eph.bright_limb_angle, distance_km=eph.distance,
                   sun_distance_km=eph.sun_distance, elongation_deg=eph.elongation, **kw)
    s.rotation = np.asarray(eph.rotation_matrix, float)
    s.u, s.v = moon_axes(s.rotation)
    return s
