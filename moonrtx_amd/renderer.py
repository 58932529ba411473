"""MoonRT: object wrapper over the C ABI of libmoonrt.so (include/moonrt.h).

One instance == one `mrtx_ctx` == what the reference holds in `self.rt` (moon_renderer.py:571-575),
minus the Tk window.  The PlotOptiX-named surface (set_data / set_displacement / setup_camera / ...)
lives in moonrtx_amd/tkoptix.py and is a thin adapter over this class.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import MrtxConfig, MrtxParams, MrtxStats, vec3


class MoonRTError(RuntimeError):
    pass


class DeviceBuffer:
    """A raw device allocation owned by the host side (inputs built on the GPU: synthetic DEMs...)."""

    def __init__(self, nbytes, device=0):
        self._lib = _lib.load()
        self.device = int(device)
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        rc = self._lib.mrtx_dev_alloc(self.device, self.nbytes, C.byref(p))
        if rc != 0 or not p.value:
            raise MoonRTError(f"device allocation of {self.nbytes} bytes failed (code {rc})")
        self.ptr = p.value

    def download(self, dtype, shape):
        out = np.empty(shape, dtype)
        if out.nbytes > self.nbytes:
            raise ValueError("download larger than the buffer")
        rc = self._lib.mrtx_dev_download(self.device, out.ctypes.data, self.ptr, out.nbytes)
        if rc != 0:
            raise MoonRTError(f"device download failed (code {rc})")
        return out

    def upload(self, array, offset=0):
        a = np.ascontiguousarray(array)
        if offset < 0 or offset + a.nbytes > self.nbytes:
            raise ValueError("upload outside the buffer")
        rc = self._lib.mrtx_dev_upload(self.device, self.ptr + int(offset), a.ctypes.data, a.nbytes)
        if rc != 0:
            raise MoonRTError(f"device upload failed (code {rc})")

    def free(self):
        if self.ptr:
            self._lib.mrtx_dev_free(self.device, self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class MoonRT:
    def __init__(self, width, height, device=0, rank=0, world=1, tile=(0, 0)):
        """`tile` = (0, 0): the library's default sharding / culling tile (16 x 16 on one or two GPUs, 32 x 32 from four ranks up)."""
        self._lib = _lib.load()
        self.width, self.height = int(width), int(height)
        self.rank, self.world = int(rank), int(world)
        cfg = MrtxConfig(int(device), self.width, self.height, self.rank, self.world, int(tile[0]), int(tile[1]))
        ctx = C.c_void_p()
        rc = self._lib.mrtx_create(C.byref(cfg), C.byref(ctx))
        self._ctx = ctx
        if rc != 0:
            msg = self._lib.mrtx_last_error(ctx).decode() if ctx.value else "invalid configuration"
            if ctx.value:
                self._lib.mrtx_destroy(ctx)
            self._ctx = None
            raise MoonRTError(f"mrtx_create failed ({rc}): {msg}")
        self.params = MrtxParams()
        self._lib.mrtx_default_params(C.byref(self.params))
        self._keepalive = {}
        import os
        if os.environ.get("MOONRT_DEFAULT_FLAGS"):      # test / debugging aid: e.g. 1 = maintain the sample counters
            self.set_params(flags=int(os.environ["MOONRT_DEFAULT_FLAGS"]))

    # ---- plumbing
    def _check(self, rc, what):
        if rc != 0:
            raise MoonRTError(f"{what} failed ({rc}): {self._lib.mrtx_last_error(self._ctx).decode()}")

    def close(self):
        if self._ctx is not None:
            self._lib.mrtx_destroy(self._ctx)
            self._ctx = None
            self._keepalive.clear()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- resources
    def upload_dem(self, elevation):
        a = np.ascontiguousarray(elevation, np.float32)
        if a.ndim != 2:
            raise ValueError("elevation must be a 2-D float32 array")
        self._check(self._lib.mrtx_upload_dem(self._ctx, a.ctypes.data, a.shape[0], a.shape[1]), "mrtx_upload_dem")
        self._keepalive.pop("dem", None)

    def bind_dem(self, buf, h, w):
        """Ingest a device-resident float32 (h, w) DEM; the context makes its own row-pair copy (8 B per texel)."""
        self._check(self._lib.mrtx_bind_dem_device(self._ctx, buf.ptr, h, w), "mrtx_bind_dem_device")

    def upload_color(self, rgba):
        if rgba is None:
            self._check(self._lib.mrtx_upload_color(self._ctx, None, 0, 0), "mrtx_upload_color")
            return
        a = np.ascontiguousarray(rgba, np.uint8)
        if a.ndim != 3 or a.shape[2] != 4:
            raise ValueError("colour texture must be (h, w, 4) uint8")
        self._check(self._lib.mrtx_upload_color(self._ctx, a.ctypes.data, a.shape[0], a.shape[1]), "mrtx_upload_color")
        self._keepalive.pop("color", None)

    def bind_color(self, buf, h, w):
        self._check(self._lib.mrtx_bind_color_device(self._ctx, buf.ptr if buf else None, h, w), "mrtx_bind_color_device")

    def upload_background(self, rgba):
        if rgba is None:
            self._check(self._lib.mrtx_upload_background(self._ctx, None, 0, 0), "mrtx_upload_background")
            return
        a = np.ascontiguousarray(rgba, np.uint8)
        if a.ndim != 3 or a.shape[2] != 4:
            raise ValueError("background must be (h, w, 4) uint8")
        self._check(self._lib.mrtx_upload_background(self._ctx, a.ctypes.data, a.shape[0], a.shape[1]),
                    "mrtx_upload_background")

    def upload_overlay(self, rgba):
        """Frame-sized RGBA8 texture composited over the tone-mapped image (the "Overlay" post-process); None removes it."""
        if rgba is None:
            self._check(self._lib.mrtx_upload_overlay(self._ctx, None, 0, 0), "mrtx_upload_overlay")
            return
        a = np.ascontiguousarray(rgba, np.uint8)
        if a.shape != (self.height, self.width, 4):
            raise ValueError("the overlay must be (height, width, 4) uint8")
        self._check(self._lib.mrtx_upload_overlay(self._ctx, a.ctypes.data, a.shape[0], a.shape[1]), "mrtx_upload_overlay")

    # ---- scene state
    def set_params(self, **kw):
        for k, v in kw.items():
            if k == "const_albedo":
                self.params.const_albedo = (C.c_float * 3)(*[float(t) for t in v])
            elif hasattr(self.params, k):
                setattr(self.params, k, v)
            else:
                raise KeyError(f"unknown renderer parameter {k!r}")
        self._check(self._lib.mrtx_set_params(self._ctx, C.byref(self.params)), "mrtx_set_params")

    def set_camera(self, eye, target, up, vfov_deg):
        self._check(self._lib.mrtx_set_camera(self._ctx, vec3(eye), vec3(target), vec3(up), float(vfov_deg)),
                    "mrtx_set_camera")

    def set_moon_frame(self, center, radius, u, v):
        self._check(self._lib.mrtx_set_moon_frame(self._ctx, vec3(center), float(radius), vec3(u), vec3(v)),
                    "mrtx_set_moon_frame")

    def set_light(self, pos, radius, radiance):
        self._check(self._lib.mrtx_set_light(self._ctx, vec3(pos), float(radius), float(radiance)), "mrtx_set_light")

    def set_sun_disk(self, pos, radius, radiance):
        self._check(self._lib.mrtx_set_sun_disk(self._ctx, vec3(pos), float(radius), float(radiance)),
                    "mrtx_set_sun_disk")

    def set_capsules(self, capsules):
        """Overlay tubes: (n, 12) float32 (see moonrtx_amd.overlays.graph_to_capsules); None / empty removes them."""
        if capsules is None or len(capsules) == 0:
            self._check(self._lib.mrtx_set_capsules(self._ctx, None, 0), "mrtx_set_capsules")
            return
        a = np.ascontiguousarray(capsules, np.float32).reshape(-1, 12)
        self._check(self._lib.mrtx_set_capsules(self._ctx, a.ctypes.data, a.shape[0]), "mrtx_set_capsules")

    def apply_scene(self, s):
        """Push a moonrtx_amd.scene.SceneDesc (everything except textures)."""
        self.set_params(scene_epsilon=s.scene_epsilon, marching_step=s.marching_step,
                        marching_step_eps=s.marching_step_eps, tonemap_exposure=s.exposure,
                        tonemap_gamma=s.gamma, spp_per_launch=s.spp_per_launch, max_spp=s.max_spp,
                        seed=s.seed, const_albedo=s.const_albedo, path_seg_min=s.path_seg_min,
                        path_seg_max=s.path_seg_max)
        self.set_camera(s.eye, s.target, s.up, s.vfov_deg)
        self.set_moon_frame(s.center, s.radius, s.u, s.v)
        self.set_light(s.light_pos, s.light_radius, s.light_radiance)
        self.set_sun_disk(s.sun_pos, s.sun_radius, s.sun_radiance)

    # ---- rendering
    def reset(self):
        self._check(self._lib.mrtx_reset_accum(self._ctx), "mrtx_reset_accum")

    def render(self, n_blocks=1):
        st = MrtxStats()
        self._check(self._lib.mrtx_render(self._ctx, int(n_blocks), C.byref(st)), "mrtx_render")
        return {name: getattr(st, name) for name, _ in MrtxStats._fields_ if name != "reserved"}

    def render_part(self, n_blocks, part, n_parts):
        """One part of the tile list (mrtx_render_part): the exchange moves part k while part k+1 renders."""
        st = MrtxStats()
        self._check(self._lib.mrtx_render_part(self._ctx, int(n_blocks), int(part), int(n_parts), C.byref(st)), "mrtx_render_part")
        return {name: getattr(st, name) for name, _ in MrtxStats._fields_ if name != "reserved"}

    def samples_done(self):
        n = C.c_uint32()
        self._check(self._lib.mrtx_samples_done(self._ctx, C.byref(n)), "mrtx_samples_done")
        return n.value

    def read_linear(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(self._lib.mrtx_read_linear(self._ctx, out.ctypes.data), "mrtx_read_linear")
        return out

    def read_rgba8(self, out=None):
        """The tone-mapped frame; `out` (a C-contiguous (H, W, 4) uint8 array) is reused when given."""
        if out is None:
            out = np.empty((self.height, self.width, 4), np.uint8)
        elif out.shape != (self.height, self.width, 4) or out.dtype != np.uint8 or not out.flags.c_contiguous:
            raise ValueError("out must be a C-contiguous (height, width, 4) uint8 array")
        self._check(self._lib.mrtx_read_rgba8(self._ctx, out.ctypes.data), "mrtx_read_rgba8")
        return out

    def read_rgb16(self):
        """The tone-mapped frame at 16 bits per sample, (H, W, 3) uint16 -- save_image(bps="Bps16")."""
        out = np.empty((self.height, self.width, 3), np.uint16)
        self._check(self._lib.mrtx_read_rgb16(self._ctx, out.ctypes.data), "mrtx_read_rgb16")
        return out

    def read_hit(self, x, y):
        """One texel of the hit buffer: (hx, hy, hz, hd), hd <= 0 == miss."""
        out = (C.c_float * 4)()
        self._check(self._lib.mrtx_read_hit(self._ctx, int(x), int(y), out), "mrtx_read_hit")
        return float(out[0]), float(out[1]), float(out[2]), float(out[3])

    def read_hits(self):
        out = np.empty((self.height, self.width, 4), np.float32)
        self._check(self._lib.mrtx_read_hits(self._ctx, out.ctypes.data), "mrtx_read_hits")
        return out

    def config(self):
        """The configuration the context runs with, defaults filled in (mrtx_get_config): device, width, height, rank, world,
        tile_w, tile_h."""
        cfg = MrtxConfig()
        self._check(self._lib.mrtx_get_config(self._ctx, C.byref(cfg)), "mrtx_get_config")
        return {name: int(getattr(cfg, name)) for name, _ in MrtxConfig._fields_}

    # ---- multi-GPU exchange
    def set_gather_hits(self, on):
        """Whether the hit buffer travels with the radiance in pack / unpack (mrtx_set_gather_hits; alike on every rank)."""
        self._check(self._lib.mrtx_set_gather_hits(self._ctx, 1 if on else 0), "mrtx_set_gather_hits")

    def shard_bytes(self, rank=None):
        n = C.c_uint64()
        self._check(self._lib.mrtx_shard_bytes(self._ctx, self.rank if rank is None else rank, C.byref(n)),
                    "mrtx_shard_bytes")
        return n.value

    def shard_bytes_active(self):
        """Bytes one rank's shard occupies for the scene as it stands (only tiles the sky cull keeps)."""
        n = C.c_uint64()
        self._check(self._lib.mrtx_shard_bytes_active(self._ctx, C.byref(n)), "mrtx_shard_bytes_active")
        return n.value

    def shard_parts(self, wanted):
        n = C.c_int32()
        self._check(self._lib.mrtx_shard_parts(self._ctx, int(wanted), C.byref(n)), "mrtx_shard_parts")
        return n.value

    def pack_part(self, dev_ptr, part, n_parts, stream=None):
        """Pack part `part` of the shard; returns (byte offset, byte length) of the piece inside the shard buffer."""
        off, ln = C.c_uint64(), C.c_uint64()
        self._check(self._lib.mrtx_pack_part(self._ctx, dev_ptr, int(part), int(n_parts), C.byref(off), C.byref(ln), stream),
                    "mrtx_pack_part")
        return off.value, ln.value

    def pack_shard(self, dev_ptr, stream=None):
        self._check(self._lib.mrtx_pack_shard(self._ctx, dev_ptr, stream), "mrtx_pack_shard")

    def unpack_shard(self, src_rank, dev_ptr, stream=None):
        self._check(self._lib.mrtx_unpack_shard(self._ctx, int(src_rank), dev_ptr, stream), "mrtx_unpack_shard")

    def unpack_all(self, dev_ptrs):
        """dev_ptrs[r] = device address of rank r's packed shard (entry 0 ignored); one sync for all peers."""
        arr = (C.c_void_p * len(dev_ptrs))(*[C.c_void_p(p) for p in dev_ptrs])
        self._check(self._lib.mrtx_unpack_all(self._ctx, arr, len(dev_ptrs)), "mrtx_unpack_all")

    def device_ptr(self, which):
        p, n = C.c_void_p(), C.c_uint64()
        self._check(self._lib.mrtx_device_ptr(self._ctx, which, C.byref(p), C.byref(n)), "mrtx_device_ptr")
        return p.value, n.value


# ---- context-free device helpers ------------------------------------------------------------------
def _err_call(fn, *args):
    lib = _lib.load()
    buf = C.create_string_buffer(256)
    rc = getattr(lib, fn)(*args, buf, 256)
    if rc != 0:
        raise MoonRTError(f"{fn} failed ({rc}): {buf.value.decode()}")


def synth_ldem(h, w, seed=0x4D525458, device=0):
    """Seeded synthetic int16 LDEM-like source, generated on the device."""
    buf = DeviceBuffer(h * w * 2, device)
    _err_call("mrtx_synth_ldem", device, buf.ptr, h, w, seed & 0xFFFFFFFF)
    return buf


def synth_color(h, w, seed=0x4D525458, device=0):
    buf = DeviceBuffer(h * w * 4, device)
    _err_call("mrtx_synth_color", device, buf.ptr, h, w, seed & 0xFFFFFFFF)
    return buf


def dem_from_ldem(src_buf, h, w, downscale=1, device=0):
    """Device restatement of load_elevation_data (data_loader.py:166-247).

    `src_buf` holds the int16 (h*downscale, w*downscale) source; returns (float32 DeviceBuffer (h, w),
    radius_scale)."""
    lib = _lib.load()
    dst = DeviceBuffer(h * w * 4, device)
    scale = C.c_float()
    buf = C.create_string_buffer(256)
    rc = lib.mrtx_dem_from_ldem(device, src_buf.ptr, h, w, downscale, dst.ptr, C.byref(scale), buf, 256)
    if rc != 0:
        raise MoonRTError(f"mrtx_dem_from_ldem failed ({rc}): {buf.value.decode()}")
    return dst, float(scale.value)


def probe_latlon(a, b, c, device=0):
    """Device evaluation of the renderer's (lat, lon) primitive for moon-frame points (a, b, c)."""
    lib = _lib.load()
    a, b, c = (np.ascontiguousarray(v, np.float32).ravel() for v in (a, b, c))
    lat = np.empty_like(a); lon = np.empty_like(a)
    rc = lib.mrtx_probe_latlon(device, a.ctypes.data, b.ctypes.data, c.ctypes.data, lat.ctypes.data,
                               lon.ctypes.data, a.size)
    if rc != 0:
        raise MoonRTError(f"mrtx_probe_latlon failed ({rc})")
    return lat, lon
