"""TkOptiX-compatible facade over MoonRT: the object MoonRTX keeps in `self.rt`.

Every member the reference touches on its renderer object (call sites cited per
method) is provided with the same name, argument meaning and threading contract:

  * scene edits may come from any thread; multi-call edits are grouped under `rt._padlock`, a re-entrant
    lock (moon_renderer.py:849-852);
  * a render thread runs accumulation cycles: after every launch it calls `on_launch_finished(rt)`
    (moon_renderer.py:574, renderer_status.py:239), after the last launch of a cycle it calls the
    `set_accum_done_cb` callback WITH the padlock held (renderer_video.py:260, :276-281), then idles
    until `refresh_scene()` (moon_renderer.py:867-871);
  * `_get_hit_at(x, y)` fetches ONE texel of the hit buffer from the device (16 bytes, under the padlock): the buffer
    itself stays in HBM -- reading all of it back after every launch costs 10.9 ms at 4K against a 0.9 ms preview launch
    (moon_renderer.py:1138 calls it once per mouse event).

What differs, by design: there is no Tk window here (no Tk on a headless GPU node) -- `_root` / `_canvas` are the
display-less stand-ins of moonrtx_amd/headless_ui.py (Tk's timer queue `after / after_idle / after_cancel` served by one thread,
inert window and canvas calls: what moon_renderer.py:398-404, :467-480 and renderer_video.py:213, :361-363 need to run
unchanged; `headless_ui=False` leaves them None, which the reference reads as "no GUI") and the `_gui_*` handler slots are plain
attributes a viewer may call.  One launch adds
`spp_per_launch` samples per pixel (a whole 64-lane wavefront per pixel) instead of one:
max_accumulation_frames=64 is ONE launch, max_accumulation_frames=1 (the interactive preview,
moon_renderer.py:457-488) is a 1-spp launch.  The encoder (NVENC H.264 in the reference) writes a Motion-JPEG AVI instead
(moonrtx_amd/video.py): same encoder_* calls, one frame per finished cycle.
"""
import os
import threading
import warnings

import numpy as np

__version__ = "0.19.2"   # the PlotOptiX API level this facade mirrors (main.py:185-201)


def write_tiff16(path, rgb16):
    """Uncompressed little-endian baseline TIFF, 3 x 16 bits per pixel, one strip."""
    import struct
    a = np.ascontiguousarray(rgb16, "<u2")
    h, w, ch = a.shape
    if ch != 3:
        raise ValueError("write_tiff16 takes an (h, w, 3) uint16 array")
    data = a.tobytes()
    n_tags = 10
    ifd_off = 8
    bps_off = ifd_off + 2 + n_tags * 12 + 4
    data_off = bps_off + 6
    tags = [(256, 4, 1, w), (257, 4, 1, h), (258, 3, 3, bps_off), (259, 3, 1, 1), (262, 3, 1, 2), (273, 4, 1, data_off),
            (277, 3, 1, 3), (278, 4, 1, h), (279, 4, 1, len(data)), (284, 3, 1, 1)]
    with open(path, "wb") as f:
        f.write(struct.pack("<2sHI", b"II", 42, ifd_off))
        f.write(struct.pack("<H", n_tags))
        for tag, typ, cnt, val in tags:
            f.write(struct.pack("<HHI", tag, typ, cnt) + (struct.pack("<HH", val, 0) if typ == 3 and cnt == 1 else struct.pack("<I", val)))
        f.write(struct.pack("<I", 0))
        f.write(struct.pack("<HHH", 16, 16, 16))
        f.write(data)


def write_png16(path, rgb16):
    """PNG, colour type 2 (RGB), bit depth 16."""
    import struct, zlib
    a = np.ascontiguousarray(rgb16, ">u2")
    h, w, ch = a.shape
    if ch != 3:
        raise ValueError("write_png16 takes an (h, w, 3) uint16 array")
    raw = np.zeros((h, 1 + w * 6), np.uint8)
    raw[:, 1:] = a.view(np.uint8).reshape(h, w * 6)          # filter type 0 on every scanline

    def chunk(kind, payload):
        return struct.pack(">I", len(payload)) + kind + payload + struct.pack(">I", zlib.crc32(kind + payload) & 0xFFFFFFFF)

    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n")
        f.write(chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 16, 2, 0, 0, 0)))
        f.write(chunk(b"IDAT", zlib.compress(raw.tobytes(), 6)))
        f.write(chunk(b"IEND", b""))


def _as3(v):
    a = np.asarray(v, float).reshape(-1)
    if a.size == 1:
        return np.array([a[0]] * 3)
    return a[:3].copy()


class _OptixShim:
    """`rt._optix.get_camera_fov(0)` / `.set_camera_fov(f)` (renderer_navigation.py:508, :521, renderer_fov.py:61)."""

    def __init__(self, owner):
        self._o = owner

    def get_camera_fov(self, handle=0):
        return float(self._o._cam["fov"])

    def set_camera_fov(self, fov):
        self._o.update_camera(fov=float(fov))


class TkOptiX:
    def __init__(self, width=-1, height=-1, on_launch_finished=None, on_rt_completed=None, start_now=False,
                 device=0, backend=None, progressive=False, headless_ui=True, **_ignored):
        """`progressive` (this backend's own option, default off): honour `min_accumulation_step` launch by launch as PlotOptiX
        does -- see set_param()."""
        if width <= 0 or height <= 0:
            raise ValueError("headless backend needs explicit width and height")
        self._width, self._height = int(width), int(height)
        self._progressive = bool(progressive) or os.environ.get("MOONRT_PROGRESSIVE") == "1"
        if backend is None:
            from .renderer import MoonRT
            backend = MoonRT(self._width, self._height, device=device)
        self._rt = backend
        self._padlock = threading.RLock()
        self._on_launch_finished = on_launch_finished
        self._on_rt_completed = on_rt_completed
        self._accum_done_cb = None
        self._optix = _OptixShim(self)
        # viewer plumbing the reference pokes at (moon_renderer.py:998-1201); no Tk here
        self._root = self._canvas = None
        if headless_ui:
            from .headless_ui import HeadlessRoot, HeadlessCanvas
            self._root, self._canvas = HeadlessRoot(self._width, self._height), HeadlessCanvas(self._width, self._height)
        self._status_action = self._status_action_text = self._status_fps = None
        self._view_orientation = None
        self._any_mouse = self._any_key = self._right_mouse = False
        self._selection_handle = None
        self._mouse_from_x = self._mouse_from_y = self._mouse_to_x = self._mouse_to_y = 0
        for slot in ("_gui_key_pressed", "_gui_motion", "_gui_pressed_left", "_gui_released_left",
                     "_gui_motion_pressed", "_gui_apply_scene_edits"):
            setattr(self, slot, lambda *a, **k: None)
        # state
        self._is_started = False
        self._is_closed = False
        self._thread = None
        self._wake = threading.Condition(self._padlock)
        self._dirty = True
        self._params = {"min_accumulation_step": 1, "max_accumulation_frames": 64}
        self._floats = {}
        self._uints = {}
        self._postproc = []
        self._textures = {}
        self._materials = {}
        self._geoms = {}
        self._graphs = {}
        self._cam_name = None
        self._cam = {"eye": np.array([0.0, -300.0, 0.0]), "target": np.zeros(3), "up": np.array([0.0, 0.0, 1.0]),
                     "fov": 4.2421875, "type": "Pinhole"}
        self._lights = {}
        self._moon_name = None
        self._sun_name = None
        self._frames_done = 0
        self._image = np.zeros((self._height, self._width, 4), np.uint8)   # reused for every read-back
        self._warned = set()
        self._encoder_cfg = None
        self._encoder = None
        self._encode_ring = []            # spare read-back buffers while an encoder holds earlier frames
        self._image_lent = False
        self.encoder_file = None
        if start_now:
            self.start()

    # ------------------------------------------------------------------ parameters
    def set_param(self, **kwargs):
        """set_param(min_accumulation_step=, max_accumulation_frames=) -- moon_renderer.py:578, :475, :487.

        DEVIATION FROM PLOTOPTIX, on purpose: the reference asks for `min_accumulation_step=1, max_accumulation_frames=64`, i.e.
        64 launches of one frame each and one `on_launch_finished` per launch (its status line counts them,
        renderer_status.py:239).  This backend renders a cycle of n frames as ONE launch of n samples per pixel (n <= 64; beyond
        that, n / 64 launches of 64) and calls `on_launch_finished` once per launch -- once per 64-frame cycle: a converged 4K
        image takes 22 ms that way, so there is nothing for a progress display to show, and 64 separate launches + image
        read-backs would cost ~5x as much (bench.py's `facade` entry times both).  `min_accumulation_step` is stored and
        otherwise ignored.  Opt in to PlotOptiX's launch-by-launch behaviour with TkOptiX(..., progressive=True) or
        MOONRT_PROGRESSIVE=1: every launch then adds `min_accumulation_step` frames (rounded down to a power of two <= 64) and
        fires its own callback; the converged image is the same bits either way (the samples are keyed by their index)."""
        with self._padlock:
            for k, v in kwargs.items():
                if k not in ("min_accumulation_step", "max_accumulation_frames", "light_shading", "compute_timeout",
                             "rt_timeout", "save_albedo", "save_normals"):
                    raise ValueError(f"unknown parameter {k}")
                self._params[k] = int(v)
            self._dirty = True
            self._wake.notify_all()

    def get_param(self, name):
        return self._params.get(name)

    def set_float(self, name, x, y=None, z=None, refresh=False):
        """set_float("scene_epsilon"|"marching_step"|"marching_step_eps"|"tonemap_exposure"|"tonemap_gamma", x)
        -- moon_renderer.py:586-599, :367."""
        with self._padlock:
            self._floats[name] = float(x)
            if name not in ("tonemap_exposure", "tonemap_gamma"):
                self._dirty = True                      # geometry-affecting: restart the cycle
            self._wake.notify_all()

    def set_uint(self, name, x, y=None, refresh=False):
        """set_uint("path_seg_range", 2, 4) -- moon_renderer.py:583."""
        with self._padlock:
            self._uints[name] = (int(x),) if y is None else (int(x), int(y))
            self._dirty = True

    def set_ambient(self, color, refresh=False):
        """set_ambient(0) -- moon_renderer.py:595.  Only zero ambient exists in this scene."""
        if float(np.max(np.asarray(color, float))) != 0.0:
            self._warn_once("ambient", "non-zero ambient light is not supported by this backend (MoonRTX uses 0)")

    def add_postproc(self, stage, refresh=False):
        """add_postproc("Gamma") -- moon_renderer.py:600; "Overlay" -- renderer_video.py:143."""
        self._postproc.append(stage)
        if stage == "Overlay":
            with self._padlock:
                self._bind_overlay()
        if stage not in ("Gamma", "Overlay"):
            self._warn_once("pp" + stage, f"post-processing stage {stage!r} is not implemented")

    # ------------------------------------------------------------------ resources
    def set_background_mode(self, mode, refresh=False):
        self._bg_mode = mode

    def set_background(self, bg, gamma=1.0, rt_format="Float4", refresh=False, **_k):
        """set_background(star_map, gamma=, rt_format="UByte4") / set_background(0) -- moon_renderer.py:606-609.

        A float (h, w, 3) map in display space is brought to the renderer's linear space with ^gamma (so the
        Gamma stage shows it unchanged) and stored as RGBA8."""
        with self._padlock:
            if np.isscalar(bg) or np.asarray(bg).ndim < 2:
                self._rt.upload_background(None)
            else:
                a = np.asarray(bg)
                if a.dtype != np.uint8:
                    lin = np.power(np.clip(a[..., :3].astype(np.float32), 0.0, 1.0), np.float32(gamma))
                    rgb = np.floor(lin * 255.0 + 0.5).astype(np.uint8)
                else:
                    rgb = a[..., :3]
                rgba = np.empty(rgb.shape[:2] + (4,), np.uint8)
                rgba[..., :3] = rgb
                rgba[..., 3] = 255
                self._rt.upload_background(rgba)
            self._dirty = True

    def set_texture_2d(self, name, data, addr_mode=None, filter_mode=None, keep_on_host=False, refresh=False, **_k):
        """set_texture_2d("moon_color", rgba_u8) -- moon_renderer.py:614; overlay textures renderer_video.py:137."""
        a = np.asarray(data)
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 4:
            raise ValueError("textures are (h, w, 4) uint8 arrays (data_loader.py:345-368)")
        with self._padlock:
            self._textures[name] = a
            self._bind_moon_material()
            self._bind_overlay()

    def setup_material(self, name, data):
        """setup_material("flat", {...}) -- moon_renderer.py:647, renderer_labels.py:288."""
        with self._padlock:
            self._materials[name] = dict(data)
            self._bind_moon_material()

    def update_material(self, name, data, refresh=False):
        """update_material("diffuse", {"ColorTextures": ["moon_color"], ...}) -- moon_renderer.py:617."""
        self.setup_material(name, data)

    OVERLAY_TEXTURE = "frame_overlay"   # renderer_video.py:27

    def _bind_overlay(self):
        """The "Overlay" stage composites the frame-sized texture named frame_overlay (renderer_video.py:137-144)."""
        tex = self._textures.get(self.OVERLAY_TEXTURE)
        if "Overlay" in self._postproc and tex is not None and tex.shape[:2] == (self._height, self._width):
            self._rt.upload_overlay(tex)
            if self._frames_done:
                self._read_image()

    def _bind_moon_material(self):
        moon = self._geoms.get(self._moon_name) if self._moon_name else None
        mat = self._materials.get(moon["mat"] if moon else "diffuse", {})
        tex = [t for t in mat.get("ColorTextures", []) if t in self._textures]
        if tex:
            self._rt.upload_color(self._textures[tex[0]])
            self._dirty = True

    def set_displacement(self, name, data, addr_mode=None, keep_on_host=False, refresh=False, **_k):
        """set_displacement("moon", elevation_f32, refresh=False) -- moon_renderer.py:624."""
        a = np.asarray(data)
        if a.ndim != 2:
            raise ValueError("displacement map must be a 2-D array")
        with self._padlock:
            self._rt.upload_dem(a.astype(np.float32, copy=False))
            self._dirty = True

    # ------------------------------------------------------------------ geometry
    def set_data(self, name, pos=None, c=None, r=None, u=None, v=None, w=None, geom="ParticleSet", geom_attr=None,
                 mat="diffuse", rnd=True, **_k):
        """set_data("moon", geom="ParticleSetTextured", geom_attr="DisplacedSurface", pos, u, v, r) and
        set_data("sun_disk", geom="ParticleSet", mat="flat", pos, r, c) -- moon_renderer.py:620-621, :648-650."""
        with self._padlock:
            g = {"geom": geom, "attr": geom_attr, "mat": mat, "pos": _as3(np.asarray(pos, float).reshape(-1)[:3]),
                 "r": float(np.asarray(r, float).reshape(-1)[0]) if r is not None else 1.0,
                 "u": _as3(u) if u is not None else np.array([0.0, 0.0, 1.0]),
                 "v": _as3(v) if v is not None else np.array([0.0, -1.0, 0.0]),
                 "c": float(np.mean(np.asarray(c, float))) if c is not None else 0.94}
            self._geoms[name] = g
            if geom_attr == "DisplacedSurface" or geom == "ParticleSetTextured":
                self._moon_name = name
                self._bind_moon_material()
            elif self._sun_name is None or name == self._sun_name:
                self._sun_name = name
            self._push_geometry()

    def update_data(self, name, pos=None, c=None, r=None, u=None, v=None, w=None, **_k):
        """update_data("moon", u=, v=) / update_data("sun_disk", pos=, r=) -- moon_renderer.py:854-855."""
        with self._padlock:
            g = self._geoms[name]
            if pos is not None:
                g["pos"] = _as3(np.asarray(pos, float).reshape(-1)[:3])
            if r is not None:
                g["r"] = float(np.asarray(r, float).reshape(-1)[0])
            if u is not None:
                g["u"] = _as3(u)
            if v is not None:
                g["v"] = _as3(v)
            if c is not None:
                g["c"] = float(np.mean(np.asarray(c, float)))
            self._push_geometry()

    def _push_geometry(self):
        if self._moon_name:
            m = self._geoms[self._moon_name]
            self._rt.set_moon_frame(m["pos"], m["r"], m["u"], m["v"])
        if self._sun_name and self._sun_name in self._geoms:
            s = self._geoms[self._sun_name]
            self._rt.set_sun_disk(s["pos"], s["r"], s["c"])
        self._dirty = True
        self._wake.notify_all()

    def set_graph(self, name, pos=None, edges=None, r=None, c=None, mat=None, **_k):
        """set_graph(name, pos=, edges=, r=, c=, mat=) -- overlay tubes (renderer_labels.py:295-300, :367-373,
        renderer_pins.py:54): flat colour, never shadowing (renderer_labels.py:132-139)."""
        with self._padlock:
            self._graphs[name] = {"pos": np.asarray(pos, float), "edges": np.asarray(edges), "r": r, "c": c, "mat": mat}
            self._push_graphs()

    def update_graph(self, name, pos=None, r=None, c=None, **_k):
        """update_graph(name, pos=, r=) -- renderer_labels.py:209, :231-234, :324-325 (r=0 hides)."""
        with self._padlock:
            g = self._graphs.get(name)
            if g is None:
                return
            if pos is not None:
                g["pos"] = np.asarray(pos, float)
            if r is not None:
                g["r"] = r
            if c is not None:
                g["c"] = c
            self._push_graphs()

    def _push_graphs(self):
        from .overlays import graph_to_capsules
        parts = [graph_to_capsules(g["pos"], g["edges"], g["r"] if g["r"] is not None else 0.01,
                                   g["c"] if g["c"] is not None else 0.8) for g in self._graphs.values()]
        caps = np.concatenate(parts) if parts else np.zeros((0, 12), np.float32)
        self._rt.set_capsules(caps)
        self._dirty = True
        self._wake.notify_all()

    def delete_geometry(self, name):
        with self._padlock:
            if self._graphs.pop(name, None) is not None:
                self._push_graphs()
            if self._geoms.pop(name, None) is not None and name == self._sun_name:
                self._rt.set_sun_disk((0, 0, 0), 0.0, 0.0)
                self._sun_name = None
                self._dirty = True

    # ------------------------------------------------------------------ camera / light
    def setup_camera(self, name, eye=None, target=None, up=None, cam_type="Pinhole", fov=None, make_current=True, **_k):
        """setup_camera(name, cam_type=, eye=, target=, up=, fov=, aperture_*=, focal_scale=) -- moon_renderer.py:627-635."""
        if cam_type != "Pinhole":
            self._warn_once("cam", f"camera type {cam_type!r} renders as Pinhole (shared_types.py:56-69)")
        self._cam_name = name
        self.update_camera(name, eye=eye, target=target, up=up, fov=fov)

    def update_camera(self, name=None, eye=None, target=None, up=None, fov=None, **_k):
        """update_camera(name, eye=, target=, up=, fov=), any subset -- renderer_navigation.py:73, :150, :224, :450."""
        with self._padlock:
            if eye is not None:
                self._cam["eye"] = _as3(eye)
            if target is not None:
                self._cam["target"] = _as3(target)
            if up is not None:
                self._cam["up"] = _as3(up)
            if fov is not None:
                self._cam["fov"] = float(fov)
            self._rt.set_camera(self._cam["eye"], self._cam["target"], self._cam["up"], self._cam["fov"])
            self._dirty = True
            self._wake.notify_all()

    def get_camera(self, name=None):
        """-> dict with "Eye", "Target", "Up" (moon_renderer.py:565-567, renderer_navigation.py:60-62)."""
        with self._padlock:
            return {"Eye": self._cam["eye"].tolist(), "Target": self._cam["target"].tolist(),
                    "Up": self._cam["up"].tolist(), "Fov": self._cam["fov"], "Type": self._cam["type"]}

    def setup_light(self, name, light_type=None, pos=None, color=None, radius=None, in_geometry=True, **_k):
        """setup_light("sun", color=brightness*460.53, radius=100, in_geometry=False) -- moon_renderer.py:640-641."""
        with self._padlock:
            self._lights[name] = {"pos": _as3(pos) if pos is not None else np.array([0.0, -21460.0, 0.0]),
                                  "radiance": float(np.mean(np.asarray(color, float))) if color is not None else 10.0,
                                  "radius": float(radius) if radius is not None else 1.0}
            self._push_light(name)

    def update_light(self, name, pos=None, color=None, radius=None, **_k):
        """update_light("sun", pos=, color=, radius=) -- moon_renderer.py:347, :859-860."""
        with self._padlock:
            lt = self._lights[name]
            if pos is not None:
                lt["pos"] = _as3(pos)
            if color is not None:
                lt["radiance"] = float(np.mean(np.asarray(color, float)))
            if radius is not None:
                lt["radius"] = float(radius)
            self._push_light(name)

    def _push_light(self, name):
        lt = self._lights[name]
        self._rt.set_light(lt["pos"], lt["radius"], lt["radiance"])
        self._dirty = True
        self._wake.notify_all()

    # ------------------------------------------------------------------ render loop
    def _cycle_plan(self):
        n = max(1, int(self._params["max_accumulation_frames"]))
        cap = min(n, 64)
        if self._progressive:               # PlotOptiX's way: min_accumulation_step frames per launch (see set_param)
            cap = min(cap, max(1, int(self._params["min_accumulation_step"])))
        s = 1
        while s * 2 <= cap:
            s *= 2
        return s, max(1, n // s)            # samples per launch, launches per cycle

    def _push_params(self, spp):
        f = self._floats
        kw = dict(spp_per_launch=spp, max_spp=int(self._params["max_accumulation_frames"]))
        for src, dst in (("scene_epsilon", "scene_epsilon"), ("marching_step", "marching_step"),
                         ("marching_step_eps", "marching_step_eps"), ("tonemap_exposure", "tonemap_exposure"),
                         ("tonemap_gamma", "tonemap_gamma")):
            if src in f:
                kw[dst] = f[src]
        if "path_seg_range" in self._uints and len(self._uints["path_seg_range"]) == 2:
            kw["path_seg_min"], kw["path_seg_max"] = self._uints["path_seg_range"]
        self._rt.set_params(**kw)

    def _launch_once(self):
        """One launch under the padlock: (re)start the cycle if the scene changed, add one block, read back."""
        spp, launches = self._cycle_plan()
        if self._dirty:
            self._rt.reset()
            self._push_params(spp)
            self._frames_done = 0
            self._dirty = False
        else:
            self._push_params(spp)
        self._rt.render(1)
        self._frames_done += 1
        self._read_image()              # 33 MB at 4K, 0.7 ms; the hit buffer stays on the device (see _get_hit_at)
        return self._frames_done >= launches

    def _read_image(self):
        if self._image_lent:             # an encoder thread is still reading that buffer: this read-back goes to the next one of the ring
            self._encode_ring.append(self._image)
            self._image = self._encode_ring.pop(0)
            self._image_lent = False
        try:
            got = self._rt.read_rgba8(out=self._image)
        except TypeError:                # a backend without the `out` parameter
            got = self._rt.read_rgba8()
        if got is not None:
            self._image = got

    def _render_loop(self):
        while True:
            with self._padlock:
                while not self._is_closed and not self._dirty and self._frames_done >= self._cycle_plan()[1]:
                    self._wake.wait(0.25)
                if self._is_closed:
                    return
                done = self._launch_once()
            if self._on_launch_finished is not None:
                self._on_launch_finished(self)                   # render thread, lock released
            if done:
                with self._padlock:
                    if self._on_rt_completed is not None:
                        self._on_rt_completed(self)
                    self._encode_cycle()
                    cb = self._accum_done_cb
                    if cb is not None and not self._dirty:
                        cb(self)                                 # padlock held (renderer_video.py:276-281)

    def start(self):
        """rt.start() -- moon_renderer.py:875-878: spawn the render thread (no Tk mainloop here)."""
        if self._is_started:
            return
        self._is_started = True
        self._thread = threading.Thread(target=self._render_loop, name="moonrt-render", daemon=True)
        self._thread.start()

    def refresh_scene(self):
        """rt.refresh_scene() -- moon_renderer.py:488, :871: force a new accumulation cycle."""
        with self._padlock:
            self._dirty = True
            self._wake.notify_all()

    def render_cycle(self):
        """Headless helper: run one full accumulation cycle synchronously on the calling thread; `on_launch_finished` fires after
        every launch, as on the render thread."""
        with self._padlock:
            self._dirty = True
            while True:
                done = self._launch_once()
                if self._on_launch_finished is not None:
                    self._on_launch_finished(self)
                if done:
                    self._encode_cycle()
                    return self._image

    def set_accum_done_cb(self, cb):
        """set_accum_done_cb(cb_or_None) -- renderer_video.py:260, :322."""
        with self._padlock:
            self._accum_done_cb = cb

    def set_launch_finished_cb(self, cb):
        self._on_launch_finished = cb

    def close(self):
        """rt.close() -- moon_renderer.py:880-884."""
        with self._padlock:
            self._is_closed = True
            self._wake.notify_all()
        if self._thread is not None and self._thread is not threading.current_thread():
            self._thread.join(timeout=10.0)
        if self._root is not None:
            self._root.destroy()                 # no callback runs after close()
        with self._padlock:
            if self._encoder is not None:
                self._encoder.close()
            if self._rt is not None:
                self._rt.close()
                self._rt = None
        self._is_started = False

    # ------------------------------------------------------------------ read-back
    def _get_image_xy(self, wx, wy):
        """Window -> image pixel (moon_renderer.py:1137); identity without a scaled Tk canvas."""
        return int(wx), int(wy)

    def _get_hit_at(self, x, y):
        """-> (hx, hy, hz, hd), hd <= 0 == miss (moon_renderer.py:1138-1142, renderer_navigation.py:195-203)."""
        if 0 <= x < self._width and 0 <= y < self._height:
            with self._padlock:
                if self._rt is None or not self._frames_done:
                    return 0.0, 0.0, 0.0, -1.0
                if hasattr(self._rt, "read_hit"):
                    h = self._rt.read_hit(int(x), int(y))
                else:
                    h = self._rt.read_hits()[int(y), int(x)]
            return float(h[0]), float(h[1]), float(h[2]), float(h[3])
        return 0.0, 0.0, 0.0, -1.0

    def get_image(self):
        return self._image

    def save_image(self, file_name, bps="Bps8"):
        """save_image(path, bps="Bps8"|"Bps16") -- renderer_dialogs.py:1222-1224 (".tiff" is saved with 16 bits per sample).

        Bps16 writes 16 bits or raises -- it never degrades to 8 bits silently: uncompressed baseline TIFF (.tif/.tiff) or
        PNG (.png), both written here (Pillow cannot write 16-bit RGB)."""
        from PIL import Image
        bps = getattr(bps, "name", bps)
        with self._padlock:
            if str(bps) == "Bps16":
                # exposure / gamma as they stand now -- and ONLY those: pushing the cycle plan here as well would try to change
                # spp_per_launch inside an accumulation cycle after set_param(max_accumulation_frames=...) (MRTX_E_STATE)
                tone = {k: self._floats[k] for k in ("tonemap_exposure", "tonemap_gamma") if k in self._floats}
                if tone:
                    self._rt.set_params(**tone)
                arr = self._rt.read_rgb16()                   # exact 16-bit "Gamma" post-process on the device
                ext = str(file_name).lower().rsplit(".", 1)[-1]
                if ext in ("tif", "tiff"):
                    write_tiff16(file_name, arr)
                elif ext == "png":
                    write_png16(file_name, arr)
                else:
                    raise ValueError(f"16 bits per sample need a .tiff/.tif or .png file name, not {file_name!r}")
                return
            Image.fromarray(self._image[..., :3]).save(file_name)

    # ------------------------------------------------------------------ headless conveniences (bench.py, tools/)
    def bind_device_inputs(self, dem_buf, dem_h, dem_w, col_buf=None, col_shape=None):
        """Device-resident DEM / colour map (moonrtx_amd.renderer.DeviceBuffer) instead of set_displacement / set_texture_2d."""
        with self._padlock:
            self._rt.bind_dem(dem_buf, dem_h, dem_w)
            if col_buf is not None:
                self._rt.bind_color(col_buf, col_shape[0], col_shape[1])
            self._dirty = True

    def apply_scene_desc(self, s):
        """Everything init_renderer + update_view push for a moonrtx_amd.scene.SceneDesc, through the PlotOptiX-named calls."""
        self.set_float("scene_epsilon", s.scene_epsilon); self.set_float("marching_step", s.marching_step)
        self.set_float("marching_step_eps", s.marching_step_eps)
        self.set_float("tonemap_exposure", s.exposure); self.set_float("tonemap_gamma", s.gamma)
        self.set_uint("path_seg_range", s.path_seg_min, s.path_seg_max)
        self.set_data("moon", geom="ParticleSetTextured", geom_attr="DisplacedSurface", pos=list(s.center), u=list(s.u),
                      v=list(s.v), r=s.radius)
        self.setup_camera("cam1", cam_type="Pinhole", eye=list(s.eye), target=list(s.target), up=list(s.up), fov=s.vfov_deg)
        self.setup_light("sun", pos=list(s.light_pos), color=s.light_radiance, radius=s.light_radius, in_geometry=False)
        self.set_data("sun_disk", geom="ParticleSet", mat="flat", pos=[list(s.sun_pos)], r=s.sun_radius, c=s.sun_radiance)

    # ------------------------------------------------------------------ encoder (NVENC H.264 in the reference)
    # PlotOptiX captures one video frame per finished accumulation cycle between encoder_start() and the frame limit /
    # encoder_stop() (renderer_video.py:219-260, :276-340).  No NVENC and no FFmpeg here: the frames go into a Motion-JPEG AVI
    # (moonrtx_amd/video.py).  A requested name that does not end in .avi gets that suffix appended (`encoder_file` says where
    # the frames went): an .mp4 name on a RIFF file would mislead every player.
    def encoder_create(self, fps, bitrate=2, idrrate=None, profile=None, preset=None):
        """encoder_create(fps=, bitrate=[Mbit/s]) -- renderer_video.py:222.  idrrate / profile / preset are H.264 notions; accepted, unused."""
        if not fps or fps <= 0 or bitrate <= 0:
            raise ValueError("fps and bitrate must be positive")
        with self._padlock:
            self._encoder_cfg = (fps, float(bitrate))

    def encoder_start(self, out_name, n_frames=0):
        """encoder_start(filename, n_frames) -- renderer_video.py:241: capture starts with the NEXT finished cycle; after n_frames
        frames (0 = until encoder_stop) the file closes by itself."""
        from .video import MjpegAviWriter
        with self._padlock:
            if self._encoder_cfg is None:
                raise RuntimeError("encoder_create() first")
            if self._encoder is not None and self._encoder.open:
                raise RuntimeError("encoder is already running")
            name = out_name if str(out_name).lower().endswith(".avi") else str(out_name) + ".avi"
            fps, mbps = self._encoder_cfg
            self._encoder = MjpegAviWriter(name, self._width, self._height, fps, mbps, n_frames=int(n_frames))
            self.encoder_file = name

    def encoder_stop(self):
        """encoder_stop() -- renderer_video.py:340."""
        with self._padlock:
            if self._encoder is not None:
                self._encoder.close()

    def encoder_is_open(self):
        """renderer_video.py:252, :290: True from encoder_start until the frame limit or encoder_stop."""
        enc = self._encoder
        return enc is not None and enc.open

    def encoded_frames(self):
        enc = self._encoder
        return enc.frames if enc is not None else 0

    def encoding_frames(self):
        enc = self._encoder
        return enc.limit if enc is not None else 0

    def _encode_cycle(self):
        """Called with the padlock held when a cycle has converged, before the accum-done callback moves the scene on."""
        enc = self._encoder
        if enc is not None and enc.open and not self._dirty:
            try:
                # the JPEG coding runs on the writer's own threads, on THIS array: the next read-backs go to other buffers of a
                # small ring (a frame is handed over, not copied)
                ring = self._encode_ring
                while len(ring) < enc.max_pending + 3:
                    ring.append(np.zeros_like(self._image))
                enc.add_frame(self._image, copy=False)
                self._image_lent = True            # still the frame get_image() shows; replaced at the next read-back
            except Exception as e:                 # PlotOptiX logs encoder failures and closes (renderer_video.py:243-251, :288-292)
                self._warn_once("encoder", f"video encoder stopped after {enc.frames} frames: {e}")
                try:
                    enc.close()
                except Exception:
                    pass

    def _warn_once(self, key, msg):
        if key not in self._warned:
            self._warned.add(key)
            warnings.warn(msg, stacklevel=3)
