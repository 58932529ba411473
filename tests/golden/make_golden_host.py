"""Generate tests/golden/host_*.json|npz by EXECUTING the reference's own host functions (SURVEY.md section 8 rows
a1, a2, a4 closed forms, a5-a9) on seeded inputs and storing inputs + outputs.

Run in the build container only (needs /root/reference):   python tests/golden/make_golden_host.py

The functions executed are pure numpy / math.  Their MODULES also import packages this image does not have (cv2,
plotoptix, skyfield, tkinter, tzlocal: the renderer, the ephemeris library, the GUI).  None of those is on the path of
the functions called here, so for the import lines only an inert placeholder module stands in `sys.modules`; a
placeholder attribute that is ever CALLED raises, which proves that no value in a fixture came out of a placeholder.
The two places where the reference hands data through the missing packages are the inputs of the fixtures:
`plotoptix.utils.read_image` (returns the seeded int16 LDEM source) and `cv2.imread` (returns the seeded BGR colour map).
The renderer object `self.rt` is a recorder: what MoonRenderer sends through the boundary (names and arguments of the
`self.rt.*` calls of `init_renderer` / `update_view`) is the fixture the facade tests replay.

The outputs are data (inputs + expected outputs); no reference source text is stored.
"""
import importlib.abc
import importlib.machinery
import json
import os
import sys
import tempfile
import types
from datetime import datetime, timezone

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ABSENT = ("cv2", "plotoptix", "skyfield", "tkinter", "tzlocal")


class PlaceholderUsed(RuntimeError):
    pass


class _InertMeta(type):
    def __getattr__(cls, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _inert(f"{cls.__name__}.{name}")

    def __call__(cls, *a, **k):
        raise PlaceholderUsed(f"placeholder {cls.__name__} was called: a fixture value would depend on a missing package")


def _inert(name):
    return _InertMeta(name, (), {})


class _InertModule(types.ModuleType):
    __path__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        v = _inert(f"{self.__name__}.{name}")
        setattr(self, name, v)
        return v


class _Finder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in ABSENT:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        return _InertModule(spec.name)

    def exec_module(self, module):
        pass


def import_reference():
    for name in ABSENT:
        try:
            __import__(name)
            raise SystemExit(f"{name} is importable here: use the real package, not a placeholder")
        except ImportError:
            pass
    sys.meta_path.append(_Finder())
    sys.path.insert(0, "/root/reference")
    from moonrtx import data_loader, astro, moon_renderer, shared_types
    return data_loader, astro, moon_renderer, shared_types


class RecordingRt:
    """Stands where `self.rt` (plotoptix.TkOptiX) stands: records every call MoonRenderer makes through the boundary."""

    def __init__(self, **kw):
        import threading
        self.calls = [("TkOptiX", {k: v for k, v in kw.items() if k in ("width", "height")})]
        self._padlock = threading.RLock()
        self._is_started = True
        self._cam = None

    def get_camera(self, name):
        return self._cam

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)

        def rec(*a, **k):
            self.calls.append((name, _plain({"args": list(a), **k})))
            if name == "setup_camera":
                self._cam = {"Eye": list(k["eye"]), "Target": list(k["target"]), "Up": list(k["up"])}
            if name == "update_camera" and "eye" in k:
                self._cam["Eye"] = list(k["eye"])
        return rec


def _plain(x):
    if isinstance(x, dict):
        return {k: _plain(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_plain(v) for v in x]
    if isinstance(x, np.ndarray):
        return {"ndarray": list(x.shape), "dtype": str(x.dtype),
                "data": x.tolist() if x.size <= 64 else None, "sum": float(np.asarray(x, np.float64).sum())}
    if isinstance(x, (np.floating, np.integer)):
        return x.item()
    if isinstance(x, (int, float, str, bool)) or x is None:
        return x
    return repr(type(x).__name__)


def rot(ax, deg):
    a = np.radians(deg); c, s = np.cos(a), np.sin(a)
    return {"x": np.array([[1, 0, 0], [0, c, -s], [0, s, c]]),
            "y": np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]),
            "z": np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])}[ax]


def ldem_source(h, w, seed):
    """Seeded LOLA-like int16 source (0.5 m units), with values beyond +-2^15/2 so that the sign of the reinterpretation
    matters, and one sharp peak."""
    rng = np.random.default_rng(seed)
    a = rng.integers(-18200, 21600, size=(h, w), dtype=np.int64)
    a[rng.integers(0, h), rng.integers(0, w)] = 21599
    return a.astype(np.int16)


def main():
    dl, astro, mr, st = import_reference()
    rng = np.random.default_rng(20261004)

    # ---- a2: _albedo_lut / _moon_texture (data_loader.py:272-287, :345-368) ----
    gammas = [0.5, 1.0, 1.8, 2.2, 3.3, 5.0]
    luts = {repr(g): dl._albedo_lut(g).tolist() for g in gammas}
    bgr = rng.integers(0, 256, size=(7, 11, 3), dtype=np.uint8)
    tex = {repr(g): dl._moon_texture(bgr, g).tolist() for g in (0.5, 2.2, 5.0)}
    json.dump({"source": "moonrtx.data_loader._albedo_lut / _moon_texture (data_loader.py:272-287, :345-368), executed",
               "lut": luts, "bgr": bgr.tolist(), "texture": tex},
              open(os.path.join(HERE, "host_albedo.json"), "w"))

    # ---- a1: load_elevation_data (data_loader.py:166-247) through an injected read_image ----
    out = {}
    meta = {"source": "moonrtx.data_loader.load_elevation_data (data_loader.py:166-247), executed; "
                      "plotoptix.utils.read_image replaced by the seeded source", "cases": []}
    with tempfile.TemporaryDirectory() as tmp:
        for i, (h, w, d, seed) in enumerate([(48, 96, 1, 11), (48, 96, 2, 12), (48, 96, 3, 13), (64, 128, 8, 14),
                                             (30, 60, 5, 15), (1200, 2400, 8, 16)]):
            src = ldem_source(h, w, seed)
            path = os.path.join(tmp, f"ldem_{i}.tif")
            open(path, "wb").write(b"II*\0")
            dl.read_image = lambda p, _s=src: _s.view(np.uint16).copy()     # the reference reinterprets as int16 itself
            elev, radius_scale = dl.load_elevation_data(path, d)
            assert elev.dtype == np.float32 and elev.shape == (h // d, w // d)
            if h <= 64:
                out[f"src{i}"] = src
            out[f"elev{i}"] = elev
            meta["cases"].append({"i": i, "h": h, "w": w, "downscale": d, "seed": seed, "radius_scale": radius_scale,
                                  "src_stored": h <= 64,
                                  "cache_files": sorted(f for f in os.listdir(tmp) if f.startswith(f"ldem_{i}.tif."))})
            if d > 1:   # the cache the reference wrote: its JSON side-car is part of the format (data_loader.py:19-95)
                js = json.load(open(f"{path}.ds{d}.json"))
                meta["cases"][-1]["cache_json_keys"] = sorted(js.keys())
                meta["cases"][-1]["cache_json"] = {k: v for k, v in js.items() if k not in ("source_mtime_ns", "source_mtime")}
                again, rs2 = dl.load_elevation_data(path, d)      # second call comes from the cache
                assert np.array_equal(again, elev) and rs2 == radius_scale
    np.savez_compressed(os.path.join(HERE, "host_elevation.npz"), **out)
    json.dump(meta, open(os.path.join(HERE, "host_elevation.json"), "w"), indent=1)

    # ---- a4 closed forms (astro.py:84-139, :167-184) ----
    cf = {"source": "moonrtx.astro closed forms (astro.py:84-139, :167-184), executed", "wrap": [], "colong": [],
          "parallactic": [], "rotation": [], "altitude": [], "latlon_from_icrf": []}
    for a in [-540.0, -180.0, -179.9, 0.0, 179.9, 180.0, 359.0, 725.5] + rng.uniform(-720, 720, 6).tolist():
        cf["wrap"].append([a, astro._wrap_signed_degrees(a)])
        cf["colong"].append([a, astro._colongitude_from_subsolar_longitude(a)])
    for _ in range(16):
        ha, dec, lat = float(rng.uniform(-180, 180)), float(rng.uniform(-28, 28)), float(rng.uniform(-89, 89))
        cf["parallactic"].append([ha, dec, lat, astro._parallactic_angle_deg(ha, dec, lat)])
    for k in range(8):
        Rm = rot("z", rng.uniform(0, 360)) @ rot("x", rng.uniform(-30, 30)) @ rot("z", rng.uniform(0, 360))
        Re = rot("x", rng.uniform(-1, 1)) @ rot("z", rng.uniform(-1, 1))
        ra, dec, q = float(rng.uniform(0, 360)), float(rng.uniform(-28, 28)), float(rng.uniform(-180, 180))
        if k == 0:
            Rm, Re, q = np.eye(3), np.eye(3), 0.0
        M = astro._rotation_matrix(Rm, Re, ra, dec, q)
        cf["rotation"].append({"R_moon": Rm.tolist(), "R_equator": Re.tolist(), "ra": ra, "dec": dec, "q": q,
                               "matrix": np.asarray(M).tolist()})
        p = rng.normal(size=3) * 0.0026
        la, lo = astro._latlon_from_icrf(p, Rm)
        cf["latlon_from_icrf"].append({"pos_au": p.tolist(), "R": Rm.tolist(), "lat": la, "lon": lo})
    for _ in range(6):
        sl, so = rng.uniform(-1.6, 1.6, 3), rng.uniform(-180, 180, 3)
        la, lo = float(rng.uniform(-90, 90)), float(rng.uniform(-180, 180))
        cf["altitude"].append({"sub_lat": sl.tolist(), "sub_lon": so.tolist(), "lat": la, "lon": lo,
                               "alt": astro._body_altitude_at_feature(sl, so, la, lo).tolist()})
    json.dump(cf, open(os.path.join(HERE, "host_astro.json"), "w"), indent=1)

    # ---- a5-a9: MoonRenderer's scene maths and what it sends through self.rt ----
    MR = mr.MoonRenderer
    consts = {k: getattr(MR, k) for k in
              ("MOON_RADIUS", "MOON_RADIUS_KM", "MOON_FILL_FRACTION", "CAMERA_DISTANCE", "MOON_REFERENCE_DISTANCE",
               "SUN_LIGHT_DISTANCE", "SUN_RADIUS", "SUN_BRIGHTNESS_SCALE", "SCENE_EPSILON", "MARCHING_STEP",
               "MARCHING_STEP_EPS", "SUN_RADIUS_KM", "SUN_DISK_DISTANCE", "SUN_DISK_COLOR", "SUN_DISK_PARKED_RADIUS",
               "ACCUMULATION_FRAMES", "PREVIEW_ACCUMULATION_FRAMES", "CAMERA_NAME", "LIGHT_NAME", "MOON_OBJECT_NAME",
               "SUN_DISK_NAME")}

    def ephem(distance, sun_distance, phase, limb, elong, R):
        return st.MoonEphemeris(az=120.0, alt=45.0, ra=10.0, dec=5.0, distance=distance, sun_distance=sun_distance,
                                phase_angle=phase, age_days=7.0, bright_limb_angle=limb, libr_long_geo=0.0,
                                libr_lat_geo=0.0, libr_long_topo=3.0, libr_lat_topo=-5.0, elongation=elong,
                                phase_name="x", colongitude=0.0, subsolar_lat=0.0, subsolar_lon=0.0, rotation_matrix=R)

    cases = []
    grid = [(384400.0, 1.496e8, 77.0, -70.0, 103.0), (384400.0, 1.496e8, 5.0, 0.0, 175.0),
            (384400.0, 1.496e8, 135.0, 100.0, 45.0), (356500.0 - 6378.0, 1.471e8, 179.6, 33.0, 0.4),
            (406700.0 + 6000.0, 1.521e8, 0.0, 180.0, 180.0), (370000.0, 1.5e8, 178.9, -12.0, 1.1)]
    for _ in range(6):
        ph = float(rng.uniform(0, 180))
        grid.append((float(rng.uniform(350000, 412000)), float(rng.uniform(1.47e8, 1.53e8)), ph,
                     float(rng.uniform(-180, 180)), float(np.clip(180.0 - ph + rng.uniform(-0.3, 0.3), 0, 180))))
    for (dist, sdist, ph, limb, elong) in grid:
        R = rot("x", float(rng.uniform(-8, 8))) @ rot("z", float(rng.uniform(-8, 8))) @ rot("y", float(rng.uniform(-30, 30)))
        obj = object.__new__(MR)
        obj.moon_ephem = ephem(dist, sdist, ph, limb, elong, R)
        light = [float(t) for t in obj.calculate_light_pos()]
        sun_c, sun_r = obj.calculate_sun_disk()
        cam = obj.default_camera
        row = {"distance": dist, "sun_distance": sdist, "phase_angle": ph, "bright_limb_angle": limb, "elongation": elong,
               "rotation_matrix": R.tolist(), "apparent_radius": obj.moon_apparent_radius(),
               "camera_distance": float(obj.moon_camera_distance()), "light_pos": light,
               "sun_disk_pos": [float(t) for t in sun_c], "sun_disk_radius": sun_r,
               "default_camera": {"eye": [float(t) for t in cam.eye], "target": list(cam.target), "up": list(cam.up),
                                  "fov": float(cam.fov), "type": cam.type}}
        # update_view with this ephemeris: everything it pushes through self.rt (moon_renderer.py:824-871)
        obj.rt = RecordingRt(width=64, height=32)
        obj.rt.setup_camera("cam1", eye=[0.0, -300.0, 0.0], target=[0.0, 0.0, 0.0], up=[0, 0, 1])
        obj.rt.calls.clear()
        obj.dt_local = datetime(2026, 10, 4, 20, 0, tzinfo=timezone.utc)
        obj.parallactic_mode = False
        obj._apparent_radius = obj.moon_apparent_radius(384400.0)
        obj.moon_grid_visible = obj.standard_labels_visible = obj.spot_labels_visible = obj.pins_visible = False
        obj.sync_datetime_dialog = lambda: None
        saved = astro.calculate_moon_ephemeris
        astro.calculate_moon_ephemeris = lambda dt, mode, _e=obj.moon_ephem: _e
        try:
            obj.update_view()
        finally:
            astro.calculate_moon_ephemeris = saved
        row["update_view_calls"] = obj.rt.calls
        cases.append(row)

    # init_renderer: the whole call sequence (moon_renderer.py:570-650)
    obj = object.__new__(MR)
    obj.width, obj.height, obj.gamma, obj.brightness = 64, 32, 2.2, 80
    obj.starmap_file, obj.color_file, obj.color_downscale = None, "/nonexistent/color.tif", 1
    obj.elevation = np.ones((4, 8), np.float32)
    obj.initial_camera = st.Camera(eye=[0, -300.0, 0], target=[0, 0, 0], up=[0, 0, 1], fov=4.2422)
    obj._on_launch_finished = lambda rt: None
    mr.TkOptiX = RecordingRt
    mr.load_color_data = lambda f, g, d: np.zeros((4, 8, 4), np.uint8)
    mr.m_diffuse = {"ClosestHitPrograms": ["x"], "ColorTextures": []}
    obj._no_shadow_flat_material = lambda: {"flat": True}
    obj.init_renderer()
    init_calls = obj.rt.calls

    json.dump({"source": "moonrtx.moon_renderer.MoonRenderer (moon_renderer.py:37-136 constants, :507-568 camera, "
                         ":570-650 init_renderer, :653-778 light / Sun disk, :824-871 update_view), executed with a "
                         "recording self.rt", "constants": consts, "cases": cases, "init_renderer_calls": init_calls},
              open(os.path.join(HERE, "host_scene.json"), "w"), indent=1)
    print("host golden vectors written to", HERE)


if __name__ == "__main__":
    main()
