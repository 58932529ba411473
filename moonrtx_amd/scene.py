"""Scene description for the headless driver: what MoonRenderer pushes through `self.rt`.

The reference keeps this logic above the renderer boundary (moon_renderer.py:507-871) and it stays
there when MoonRTX runs on this backend.  The headless bench / tests have no MoonRenderer and no
Skyfield, so the closed-form pieces are restated here (float64, each citing its source) and the
ephemeris numbers (phase angle, bright-limb angle, distances, rotation matrix) are plain inputs.
"""
import math
from dataclasses import dataclass, field, replace
from typing import Sequence

import numpy as np

# moon_renderer.py:37-116
MOON_RADIUS = 10.0
MOON_RADIUS_KM = 1737.4
MOON_FILL_FRACTION = 0.9
CAMERA_DISTANCE = 30.0 * MOON_RADIUS
MOON_REFERENCE_DISTANCE_KM = 384_400.0
SUN_LIGHT_DISTANCE = 21460.0
SUN_RADIUS_KM = 695_700.0
SUN_BRIGHTNESS_SCALE = (2146.0 / 100.0) ** 2
SUN_DISK_DISTANCE = 3100.0
SUN_DISK_COLOR = 2.0
SUN_DISK_PARKED_RADIUS = 0.01
SCENE_EPSILON, MARCHING_STEP, MARCHING_STEP_EPS = 1.0e-4, 5.0e-3, 3.0e-4
ACCUMULATION_FRAMES = 64


def apparent_radius(distance_km):
    """Angular radius of the Moon (rad) from `distance_km` -- moon_renderer.py:522-529."""
    return math.asin(MOON_RADIUS_KM / distance_km)


def camera_distance(distance_km):
    """Scene distance that renders the Moon at its true apparent size -- moon_renderer.py:531-544."""
    return CAMERA_DISTANCE * apparent_radius(MOON_REFERENCE_DISTANCE_KM) / apparent_radius(distance_km)


def default_vfov_deg():
    """Vertical field of view of the whole-disk camera -- moon_renderer.py:513-519."""
    visible = 2.0 * MOON_RADIUS / MOON_FILL_FRACTION
    fov = math.degrees(2.0 * math.atan(visible / (2.0 * CAMERA_DISTANCE)))
    return min(90.0, max(1.0, fov))


def light_position(phase_deg, bright_limb_deg):
    """Sun light position from phase angle and bright-limb angle -- moon_renderer.py:676-727."""
    beta, phi = math.radians(bright_limb_deg), math.radians(phase_deg)
    sp = math.sin(phi) * SUN_LIGHT_DISTANCE
    return (-math.sin(beta) * sp, -math.cos(phi) * SUN_LIGHT_DISTANCE, math.cos(beta) * sp)


def light_radius(sun_distance_km):
    """Radius of the light sphere for the Sun distance of the date -- moon_renderer.py:859."""
    return SUN_LIGHT_DISTANCE * SUN_RADIUS_KM / sun_distance_km


def light_radiance(brightness):
    """Light colour (radiance) for a brightness setting -- moon_renderer.py:640, :347."""
    return brightness * SUN_BRIGHTNESS_SCALE


def sun_disk(distance_km, sun_distance_km, elongation_deg, bright_limb_deg):
    """Centre and radius of the visible Sun disk -- moon_renderer.py:749-778."""
    cam = camera_distance(distance_km)
    magnification = math.asin(MOON_RADIUS / cam) / apparent_radius(distance_km)
    ang = magnification * math.asin(SUN_RADIUS_KM / sun_distance_km)
    sep = magnification * math.radians(elongation_deg)
    visible = sep <= math.pi / 2.0
    if not visible:
        sep = math.radians(175.0)
    beta = math.radians(bright_limb_deg)
    d = (-math.sin(beta) * math.sin(sep), math.cos(sep), math.cos(beta) * math.sin(sep))
    centre = (SUN_DISK_DISTANCE * d[0], -cam + SUN_DISK_DISTANCE * d[1], SUN_DISK_DISTANCE * d[2])
    return centre, (SUN_DISK_DISTANCE * math.tan(ang) if visible else SUN_DISK_PARKED_RADIUS)


def moon_axes(rotation):
    """(u, v) handed to update_data("moon", u=, v=) -- moon_renderer.py:844-845."""
    r = np.asarray(rotation, float)
    return tuple(r[:, 2]), tuple(-r[:, 1])


def libration_rotation(l_deg, b_deg):
    """Rotation that turns the body point (lat b, lon l) toward the camera (scene -Y).

    Body frame: +Z north, -Y longitude 0, +X longitude 90 E (renderer_navigation.py:47-53)."""
    l, b = math.radians(l_deg), math.radians(b_deg)
    cz, sz = math.cos(-l), math.sin(-l)
    rz = np.array([[cz, -sz, 0.0], [sz, cz, 0.0], [0.0, 0.0, 1.0]])
    cx, sx = math.cos(b), math.sin(b)
    rx = np.array([[1.0, 0.0, 0.0], [0.0, cx, -sx], [0.0, sx, cx]])
    return rx @ rz


def body_point(lat_deg, lon_deg, radius=MOON_RADIUS):
    """Selenographic (lat, lon) -> body-frame position (renderer_navigation.py:44-53)."""
    la, lo = math.radians(lat_deg), math.radians(lon_deg)
    return np.array([radius * math.cos(la) * math.sin(lo), -radius * math.cos(la) * math.cos(lo),
                     radius * math.sin(la)])


def selenographic(hit, rotation, radius=MOON_RADIUS):
    """Scene-space hit -> (lat, lon) degrees or (None, None) -- renderer_navigation.py:471-492."""
    p = np.asarray(hit, float)
    r = float(np.linalg.norm(p))
    if r < radius * 0.9 or r > radius * 1.15:
        return None, None
    x, y, z = np.asarray(rotation, float).T @ (p / r)
    return math.degrees(math.asin(max(-1.0, min(1.0, z)))), math.degrees(math.atan2(x, -y))


@dataclass
class SceneDesc:
    width: int = 512
    height: int = 512
    # march / accumulation (moon_renderer.py:578-600)
    scene_epsilon: float = SCENE_EPSILON
    marching_step: float = MARCHING_STEP
    marching_step_eps: float = MARCHING_STEP_EPS
    exposure: float = 0.9
    gamma: float = 2.2
    spp_per_launch: int = 64
    max_spp: int = ACCUMULATION_FRAMES
    seed: int = 1
    # path length: 1 = direct light only (the headline workload); the reference sets (2, 4) -- moon_renderer.py:583
    path_seg_min: int = 1
    path_seg_max: int = 1
    const_albedo: Sequence[float] = (75.0 / 255.0,) * 3   # lut[128] at gamma 2.2 (data_loader.py:283-287)
    # camera (moon_renderer.py:513-520)
    eye: Sequence[float] = (0.0, -CAMERA_DISTANCE, 0.0)
    target: Sequence[float] = (0.0, 0.0, 0.0)
    up: Sequence[float] = (0.0, 0.0, 1.0)
    vfov_deg: float = field(default_factory=default_vfov_deg)
    # moon (moon_renderer.py:620-621)
    center: Sequence[float] = (0.0, 0.0, 0.0)
    radius: float = MOON_RADIUS
    u: Sequence[float] = (0.0, 0.0, 1.0)
    v: Sequence[float] = (0.0, -1.0, 0.0)
    rotation: np.ndarray = field(default_factory=lambda: np.eye(3))
    # light + Sun disk
    light_pos: Sequence[float] = (0.0, -SUN_LIGHT_DISTANCE, 0.0)
    light_radius: float = 100.0
    light_radiance: float = 80.0 * SUN_BRIGHTNESS_SCALE
    sun_pos: Sequence[float] = (0.0, SUN_DISK_DISTANCE, 0.0)
    sun_radius: float = SUN_DISK_PARKED_RADIUS
    sun_radiance: float = SUN_DISK_COLOR

    def with_size(self, width, height, spp_per_launch=None):
        s = replace(self, width=int(width), height=int(height))
        if spp_per_launch is not None:
            s.spp_per_launch = int(spp_per_launch)
        return s


def make_scene(width, height, phase_deg, bright_limb_deg, *, spp_per_launch=64, brightness=80.0, gamma=2.2,
               distance_km=MOON_REFERENCE_DISTANCE_KM, sun_distance_km=1.496e8, libration=(3.0, -5.0), seed=1,
               elongation_deg=None):
    """Everything init_renderer + update_view push for one date (moon_renderer.py:570-650, :824-871)."""
    rot = libration_rotation(*libration)
    u, v = moon_axes(rot)
    if elongation_deg is None:
        elongation_deg = 180.0 - phase_deg
    sun_c, sun_r = sun_disk(distance_km, sun_distance_km, elongation_deg, bright_limb_deg)
    return SceneDesc(
        width=int(width), height=int(height), spp_per_launch=int(spp_per_launch), gamma=gamma, seed=seed,
        eye=(0.0, -camera_distance(distance_km), 0.0), u=u, v=v, rotation=rot,
        light_pos=light_position(phase_deg, bright_limb_deg), light_radius=light_radius(sun_distance_km),
        light_radiance=light_radiance(brightness), sun_pos=sun_c, sun_radius=sun_r)


# the three benchmark scenes
SCENES = {
    "S1": dict(phase_deg=77.0, bright_limb_deg=-70.0),    # quarter-ish, long terminator shadows (headline)
    "S2": dict(phase_deg=5.0, bright_limb_deg=0.0),       # near-full
    "S3": dict(phase_deg=135.0, bright_limb_deg=100.0),   # crescent
}


def named_scene(name, width, height, **kw):
    return make_scene(width, height, **SCENES[name], **kw)


def zoomed_on_terminator(name, width, height, vfov_deg=0.7, **kw):
    """The interactive close-up the reference spends its time in (zoom / pan, renderer_navigation.py:226-521): scene `name`
    with the default camera turned onto the point of the terminator nearest the disc centre and the field of view
    narrowed to `vfov_deg` (the whole-disc view is 4.24 deg; at 0.7 deg a 4K pixel covers ~1.2 texels of a
    downscale-2 DEM)."""
    s = named_scene(name, width, height, **kw)
    light = np.asarray(s.light_pos, float) - np.asarray(s.center, float)
    light /= np.linalg.norm(light)
    to_eye = np.asarray(s.eye, float) - np.asarray(s.center, float)
    to_eye /= np.linalg.norm(to_eye)
    n = to_eye - (to_eye @ light) * light            # on the terminator (n . light = 0), as much toward the camera as possible
    n /= np.linalg.norm(n)
    s.target = tuple(np.asarray(s.center, float) + s.radius * n)
    s.vfov_deg = float(vfov_deg)
    return s
