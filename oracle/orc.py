"""ctypes binding of the CPU oracle (oracle/mrtx_oracle.c).

TEST INFRASTRUCTURE, NOT PRODUCT: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg import this module.  Nothing under moonrtx_amd/ does.

Parity status: UNPINNED against PlotOptiX (closed third-party renderer, absent); conventions pinned
by tests/golden/*.json captured from the reference's importable modules (see mrtx_oracle.c header).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))


class OrcScene(C.Structure):
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("scene_epsilon", C.c_float), ("marching_step", C.c_float), ("marching_step_eps", C.c_float),
        ("spp_per_block", C.c_uint32), ("seed", C.c_uint32),
        ("path_seg_min", C.c_uint32), ("path_seg_max", C.c_uint32),
        ("const_albedo", C.c_float * 3),
        ("eye", C.c_double * 3), ("target", C.c_double * 3), ("up", C.c_double * 3), ("vfov_deg", C.c_double),
        ("center", C.c_double * 3), ("radius", C.c_double), ("u", C.c_double * 3), ("v", C.c_double * 3),
        ("light_pos", C.c_double * 3), ("light_radius", C.c_double), ("light_radiance", C.c_double),
        ("sun_pos", C.c_double * 3), ("sun_radius", C.c_double), ("sun_radiance", C.c_double),
        ("dem", C.c_void_p), ("dem_h", C.c_int32), ("dem_w", C.c_int32),
        ("color", C.c_void_p), ("color_h", C.c_int32), ("color_w", C.c_int32),
        ("bg", C.c_void_p), ("bg_h", C.c_int32), ("bg_w", C.c_int32),
        ("caps", C.c_void_p), ("n_caps", C.c_int32),
    ]


STAT_NAMES = ("primary_rays", "primary_hits", "shadow_rays", "height_samples", "colour_fetches",
              "background_fetches", "bounce_rays", "bounce_sun_hits")

_lib = None


def build(force=False):
    """Compile the oracle with the recipe in oracle/Makefile (gcc only)."""
    want = [os.path.join(_HERE, n) for n in ("liborc_fma.so", "liborc_soft.so")]
    src = os.path.join(_HERE, "mrtx_oracle.c")
    if force or any((not os.path.isfile(p)) or os.path.getmtime(p) < os.path.getmtime(src) for p in want):
        subprocess.check_call(["make", "-C", _HERE, "-s", "all"])


def _has_fma():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return " fma " in (line + " ")
    except OSError:
        pass
    return False


def lib():
    global _lib
    if _lib is None:
        build()
        name = "liborc_fma.so" if _has_fma() else "liborc_soft.so"
        L = C.CDLL(os.path.join(_HERE, name))
        L.orc_latlon.restype = None
        L.orc_latlon.argtypes = [C.c_float, C.c_float, C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_float)]
        L.orc_dem_bilinear.restype = C.c_float
        L.orc_dem_bilinear.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_float]
        L.orc_render.restype = C.c_int
        L.orc_render.argtypes = [C.POINTER(OrcScene), C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32,
                                 C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_resolve_linear.restype = None
        L.orc_resolve_linear.argtypes = [C.c_void_p, C.c_int64, C.c_uint32, C.c_void_p]
        L.orc_tone_table.restype = None
        L.orc_tone_table.argtypes = [C.c_double, C.c_int32, C.c_void_p]
        L.orc_resolve_rgba8.restype = None
        L.orc_resolve_rgba8.argtypes = [C.c_void_p, C.c_int64, C.c_uint32, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
        L.orc_resolve_rgb16.restype = None
        L.orc_resolve_rgb16.argtypes = [C.c_void_p, C.c_int64, C.c_uint32, C.c_float, C.c_float, C.c_void_p]
        L.orc_frame_floats.restype = None
        L.orc_frame_floats.argtypes = [C.POINTER(OrcScene), C.c_void_p]
        L.orc_dem_from_ldem.restype = C.c_float
        L.orc_dem_from_ldem.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
        L.orc_sizeof_scene.restype = C.c_int
        L.orc_debug_counter.restype = C.c_uint64
        L.orc_set_threads.restype = C.c_int
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_set_exact.restype = C.c_int
        L.orc_set_exact.argtypes = [C.c_int]
        assert L.orc_sizeof_scene() == C.sizeof(OrcScene), "OrcScene layout mismatch"
        _lib = L
    return _lib


def quad_out_of_range():
    """Number of quadratic-segment taps that left [-1, h) x [-1, w) since load; the spec requires 0."""
    return int(lib().orc_debug_counter())


def set_exact(on):
    """Test switch: evaluate the texel coordinates of EVERY march step exactly (no per-segment quadratic).
    Returns the previous setting."""
    return bool(lib().orc_set_exact(1 if on else 0))


def set_threads(n):
    """Cap the OpenMP team of orc_render; returns the thread count in effect."""
    return int(lib().orc_set_threads(int(n)))


def latlon(a, b, c):
    """The spec's (lat, lon) of a moon-frame point, element-wise (float32 arrays in, two arrays out)."""
    L = lib()
    a, b, c = (np.asarray(v, np.float32).ravel() for v in (a, b, c))
    lat = np.empty_like(a); lon = np.empty_like(a)
    la, lo = C.c_float(), C.c_float()
    for i in range(a.size):
        L.orc_latlon(float(a[i]), float(b[i]), float(c[i]), C.byref(la), C.byref(lo))
        lat[i] = la.value; lon[i] = lo.value
    return lat, lon


def tone_table(gamma, n):
    """Thresholds T[0..n] of the spec's "Gamma" post-process (T[0] = 0 is not a threshold)."""
    T = np.empty(n + 1, np.float32)
    lib().orc_tone_table(float(gamma), int(n), T.ctypes.data)
    return T


def dem_bilinear(dem, lat_rad, lon_rad):
    dem = np.ascontiguousarray(dem, np.float32)
    return float(lib().orc_dem_bilinear(dem.ctypes.data, dem.shape[0], dem.shape[1],
                                        np.float32(lat_rad), np.float32(lon_rad)))


def dem_from_ldem(src_i16, downscale):
    src = np.ascontiguousarray(src_i16, np.int16)
    h, w = src.shape[0] // downscale, src.shape[1] // downscale
    src = np.ascontiguousarray(src[:h * downscale, :w * downscale])
    out = np.empty((h, w), np.float32)
    mx = lib().orc_dem_from_ldem(src.ctypes.data, h, w, downscale, out.ctypes.data)
    return out, float(mx)


class Oracle:
    """Stateful wrapper that mirrors the product's create/upload/set/render/read sequence."""

    def __init__(self, scene, dem, color=None, bg=None, capsules=None):
        """`scene` is any object with the attribute names of moonrtx_amd.scene.SceneDesc."""
        self.L = lib()
        self.dem = np.ascontiguousarray(dem, np.float32)
        self.color = None if color is None else np.ascontiguousarray(color, np.uint8)
        self.bg = None if bg is None else np.ascontiguousarray(bg, np.uint8)
        s = OrcScene()
        s.width, s.height = int(scene.width), int(scene.height)
        s.scene_epsilon = scene.scene_epsilon
        s.marching_step = scene.marching_step
        s.marching_step_eps = scene.marching_step_eps
        s.spp_per_block = int(scene.spp_per_launch)
        s.seed = int(scene.seed)
        s.path_seg_min = int(getattr(scene, "path_seg_min", 1))
        s.path_seg_max = int(getattr(scene, "path_seg_max", 1))
        s.const_albedo = (C.c_float * 3)(*scene.const_albedo)
        for name in ("eye", "target", "up", "center", "u", "v", "light_pos", "sun_pos"):
            setattr(s, name, (C.c_double * 3)(*[float(t) for t in getattr(scene, name)]))
        s.vfov_deg = float(scene.vfov_deg)
        s.radius = float(scene.radius)
        s.light_radius = float(scene.light_radius)
        s.light_radiance = float(scene.light_radiance)
        s.sun_radius = float(scene.sun_radius)
        s.sun_radiance = float(scene.sun_radiance)
        s.dem = self.dem.ctypes.data
        s.dem_h, s.dem_w = self.dem.shape
        if self.color is not None:
            s.color = self.color.ctypes.data
            s.color_h, s.color_w = self.color.shape[:2]
        if self.bg is not None:
            s.bg = self.bg.ctypes.data
            s.bg_h, s.bg_w = self.bg.shape[:2]
        self.caps = None
        if capsules is not None and len(capsules):
            self.caps = np.ascontiguousarray(capsules, np.float32).reshape(-1, 12)
            s.caps = self.caps.ctypes.data
            s.n_caps = self.caps.shape[0]
        self.s = s
        self.accum = np.zeros((s.height, s.width, 4), np.float32)
        self.hits = np.zeros((s.height, s.width, 4), np.float32)
        self.stats = np.zeros(len(STAT_NAMES), np.uint64)
        self.blocks_done = 0

    def reset(self):
        self.accum[:] = 0
        self.hits[:] = 0
        self.stats[:] = 0
        self.blocks_done = 0

    def render(self, n_blocks=1, region=None):
        x0, y0, x1, y1 = region if region else (0, 0, self.s.width, self.s.height)
        rc = self.L.orc_render(C.byref(self.s), x0, y0, x1, y1, self.blocks_done, n_blocks,
                               self.accum.ctypes.data, self.hits.ctypes.data, self.stats.ctypes.data)
        if rc != 0:
            raise ValueError("orc_render rejected the scene")
        self.blocks_done += n_blocks
        return dict(zip(STAT_NAMES, (int(v) for v in self.stats)))

    def linear(self):
        out = np.empty_like(self.accum)
        self.L.orc_resolve_linear(self.accum.ctypes.data, self.s.width * self.s.height,
                                  self.blocks_done * self.s.spp_per_block, out.ctypes.data)
        return out

    def rgba8(self, exposure=0.9, gamma=2.2, overlay=None):
        """The tone-mapped 8-bit frame (exposure + "Gamma" post-process, optional RGBA8 overlay composited on top)."""
        out = np.empty((self.s.height, self.s.width, 4), np.uint8)
        ov = None if overlay is None else np.ascontiguousarray(overlay, np.uint8)
        self.L.orc_resolve_rgba8(self.accum.ctypes.data, self.s.width * self.s.height, self.blocks_done * self.s.spp_per_block,
                                 exposure, gamma, None if ov is None else ov.ctypes.data, out.ctypes.data)
        return out

    def rgb16(self, exposure=0.9, gamma=2.2):
        out = np.empty((self.s.height, self.s.width, 3), np.uint16)
        self.L.orc_resolve_rgb16(self.accum.ctypes.data, self.s.width * self.s.height, self.blocks_done * self.s.spp_per_block,
                                 exposure, gamma, out.ctypes.data)
        return out

    def frame_floats(self):
        out = np.zeros(64, np.float32)
        self.L.orc_frame_floats(C.byref(self.s), out.ctypes.data)
        return out[:46]
