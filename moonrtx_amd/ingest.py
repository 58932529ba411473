"""Data ingest for the MI355X backend: the step right before `set_displacement` / `set_texture_2d`.

Mirrors the interface of the reference's `moonrtx/data_loader.py` (same function names, arguments, return
values and on-disk cache format) so MoonRenderer can call it unchanged, but does the heavy part on the GPU:

  * `read_image`        -- replaces `plotoptix.utils.read_image` (data_loader.py:10, :206): a TIFF / BigTIFF reader
                           for uncompressed strip or tile images (the 7.9 GB LDEM is 92160 x 46080 x 16 bit, beyond
                           classic TIFF's 4 GB), memory-mapped so nothing is copied until it is uploaded;
  * `load_elevation_data` -- data_loader.py:166-247: int16 LDEM -> block mean -> *0.5/1737400 -> +1 -> /max, with the
                           reduction run by `mrtx_dem_from_ldem` on the device (HBM-bound, milliseconds instead of the
                           ~1 min the reference quotes, data_loader.py:12-14) and the reference's `<src>.ds<N>.npy` +
                           `.json` sidecar cache read and written in its own format (data_loader.py:19-95), so caches
                           made by either side are valid for the other;
  * `load_color_data`   -- data_loader.py:290-368: BGR/RGB bytes -> RGBA bytes through the 256-entry albedo LUT;
  * `load_starmap`      -- data_loader.py:371-425.
"""
import json
import mmap
import os
import struct

import numpy as np

CACHE_VERSION = 1                      # data_loader.py:19
LDEM_METERS_PER_UNIT = 0.5             # data_loader.py:162
MOON_REFERENCE_RADIUS_M = 1_737_400.0  # data_loader.py:163
ALBEDO_MIN, ALBEDO_RANGE = 0.2, 0.75   # data_loader.py:263-264

_TIFF_TYPES = {1: "B", 2: "c", 3: "H", 4: "I", 5: "II", 6: "b", 8: "h", 9: "i", 11: "f", 12: "d", 16: "Q", 17: "q", 18: "Q"}


class TiffError(ValueError):
    pass


def _read_ifd(buf, big, bo):
    """First IFD of a (Big)TIFF held in `buf` -> {tag: tuple of values}."""
    if big:
        (off,) = struct.unpack_from(bo + "Q", buf, 8)
        (n,) = struct.unpack_from(bo + "Q", buf, off)
        pos, esz, cnt_fmt, inline = off + 8, 20, "Q", 8
    else:
        (off,) = struct.unpack_from(bo + "I", buf, 4)
        (n,) = struct.unpack_from(bo + "H", buf, off)
        pos, esz, cnt_fmt, inline = off + 2, 12, "I", 4
    tags = {}
    for i in range(n):
        e = pos + i * esz
        tag, typ = struct.unpack_from(bo + "HH", buf, e)
        (count,) = struct.unpack_from(bo + cnt_fmt, buf, e + 4)
        fmt = _TIFF_TYPES.get(typ)
        if fmt is None:
            continue
        unit = struct.calcsize(bo + fmt)
        vpos = e + 4 + struct.calcsize(cnt_fmt)
        if unit * count > inline:
            (vpos,) = struct.unpack_from(bo + cnt_fmt, buf, vpos)
        if typ == 2:
            tags[tag] = bytes(buf[vpos:vpos + count])
        elif typ == 5:
            vals = struct.unpack_from(bo + "II" * count, buf, vpos)
            tags[tag] = tuple(vals[2 * k] / max(1, vals[2 * k + 1]) for k in range(count))
        else:
            tags[tag] = struct.unpack_from(bo + fmt * count, buf, vpos)
    return tags


def read_image(path, normalized=False):
    """Read an uncompressed TIFF or BigTIFF into a numpy array (2-D, or 3-D for multi-sample pixels).

    The result is a read-only view of a memory map whenever the pixel data is one contiguous run (the LDEM
    products are), so an 8 GB file costs no host memory until it is uploaded.  Like `plotoptix.utils.read_image`
    the dtype is the file's unsigned type; MoonRTX reinterprets the LDEM as int16 itself (data_loader.py:215)."""
    f = open(path, "rb")
    try:
        mm = mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ)
    finally:
        f.close()
    head = mm[:4]
    if head[:2] == b"II":
        bo = "<"
    elif head[:2] == b"MM":
        bo = ">"
    else:
        raise TiffError(f"{path}: not a TIFF file")
    (magic,) = struct.unpack_from(bo + "H", mm, 2)
    if magic not in (42, 43):
        raise TiffError(f"{path}: bad TIFF magic {magic}")
    t = _read_ifd(mm, magic == 43, bo)
    w, h = int(t[256][0]), int(t[257][0])
    bits = t.get(258, (1,))
    spp = int(t.get(277, (1,))[0])
    comp = int(t.get(259, (1,))[0])
    fmt = int(t.get(339, (1,))[0])
    planar = int(t.get(284, (1,))[0])
    if comp != 1:
        raise TiffError(f"{path}: compressed TIFF (compression={comp}) is not supported by this reader; "
                        "use Pillow (moonrtx_amd.ingest.load_color_data does) or convert to uncompressed")
    if planar != 1 and spp > 1:
        raise TiffError(f"{path}: planar sample layout is not supported")
    if len(set(bits)) != 1 or bits[0] not in (8, 16, 32):
        raise TiffError(f"{path}: unsupported BitsPerSample {bits}")
    base = {(8, 1): "u1", (16, 1): "u2", (32, 1): "u4", (8, 2): "i1", (16, 2): "i2", (32, 2): "i4", (32, 3): "f4"}.get((bits[0], fmt))
    if base is None:
        raise TiffError(f"{path}: unsupported sample format {fmt} at {bits[0]} bits")
    dt = np.dtype(bo + base)
    row_bytes = w * spp * dt.itemsize
    shape = (h, w) if spp == 1 else (h, w, spp)
    if 322 in t:   # tiled
        tw, tl = int(t[322][0]), int(t[323][0])
        offs, cnts = t[324], t[325]
        out = np.empty(shape, dt.newbyteorder("="))
        tiles_x = (w + tw - 1) // tw
        for k, (o, c) in enumerate(zip(offs, cnts)):
            ty, tx = divmod(k, tiles_x)
            tile = np.frombuffer(mm, dt, count=tw * tl * spp, offset=o).reshape((tl, tw) if spp == 1 else (tl, tw, spp))
            y0, x0 = ty * tl, tx * tw
            out[y0:y0 + tl, x0:x0 + tw] = tile[:min(tl, h - y0), :min(tw, w - x0)]
        return out
    offs, cnts = t[273], t.get(279)
    rps = int(t.get(278, (h,))[0])
    contiguous = all(offs[i + 1] == offs[i] + min(rps, h - i * rps) * row_bytes for i in range(len(offs) - 1))
    if contiguous:
        arr = np.frombuffer(mm, dt, count=h * w * spp, offset=offs[0]).reshape(shape)
        return arr if bo == "<" else arr.astype(dt.newbyteorder("="))
    out = np.empty(shape, dt.newbyteorder("="))
    for i, o in enumerate(offs):
        r0 = i * rps
        n = min(rps, h - r0)
        out[r0:r0 + n] = np.frombuffer(mm, dt, count=n * w * spp, offset=o).reshape((n,) + shape[1:])
    return out


# ---------------------------------------------------------------------------------- cache (data_loader.py:19-95)
def cache_fingerprint(filepath, **params):
    fp = {"version": CACHE_VERSION, **params}
    if os.path.isfile(filepath):
        fp["source_size"] = os.path.getsize(filepath)
        fp["source_mtime"] = int(os.path.getmtime(filepath))
    return fp


def cache_meta(cache_base, fingerprint):
    try:
        with open(cache_base + ".json", "r", encoding="utf-8") as f:
            meta = json.load(f)
    except Exception:
        return None
    if any(meta.get(k) != v for k, v in fingerprint.items()):
        return None
    return meta if os.path.isfile(cache_base + ".npy") else None


def downscale_cache_available(filepath, downscale):
    """data_loader.py:63-85."""
    if downscale <= 1:
        return False
    return cache_meta(f"{filepath}.ds{downscale}", cache_fingerprint(filepath, downscale=downscale)) is not None


def _save_cache(cache_base, array, meta):
    try:
        np.save(cache_base + ".npy", array)
        with open(cache_base + ".json", "w", encoding="utf-8") as f:
            json.dump(meta, f)
    except Exception as e:   # a cache can only cost time, never correctness (data_loader.py:16-18)
        print(f"Warning: could not write cache {cache_base}.npy: {e}")


# ---------------------------------------------------------------------------------- elevation
def elevation_to_device(elev_src, downscale, device=0, chunk_rows=2048):
    """int16 (H, W) host array (possibly a memory map) -> (float32 DeviceBuffer (h, w), h, w, radius_scale).

    The source is streamed to the GPU in row bands and reduced there by mrtx_dem_from_ldem."""
    from .renderer import DeviceBuffer, dem_from_ldem
    src = elev_src.view(np.int16) if elev_src.dtype != np.int16 else elev_src
    h, w = src.shape[0] // downscale, src.shape[1] // downscale
    H, W = h * downscale, w * downscale
    dev = DeviceBuffer(H * W * 2, device)
    for r0 in range(0, H, chunk_rows):
        band = np.ascontiguousarray(src[r0:min(H, r0 + chunk_rows), :W])
        dev.upload(band, offset=r0 * W * 2)
    out, scale = dem_from_ldem(dev, h, w, downscale, device)
    dev.free()
    return out, h, w, scale


def load_elevation_data(filepath, downscale, device=0):
    """data_loader.py:166-247 -> (float32 (h, w) displacement factors with peak exactly 1.0, radius_scale)."""
    cache_base = f"{filepath}.ds{downscale}"
    fingerprint = None
    if downscale > 1:
        fingerprint = cache_fingerprint(filepath, downscale=downscale)
        meta = cache_meta(cache_base, fingerprint)
        if meta is not None:
            try:
                return np.load(cache_base + ".npy"), float(meta["radius_scale"])
            except Exception:
                pass
    if not os.path.isfile(filepath):
        raise FileNotFoundError(f"Elevation file not found: {filepath}, and no cache of it downscaled by {downscale} beside it.")
    src = read_image(filepath)
    if src.ndim != 2 or src.dtype.itemsize != 2:
        raise ValueError(f"{filepath}: expected a single-channel 16-bit LDEM image, got {src.shape} {src.dtype}")
    buf, h, w, scale = elevation_to_device(src, downscale, device)
    elevation = buf.download(np.float32, (h, w))
    buf.free()
    if fingerprint is not None:
        _save_cache(cache_base, elevation, {**fingerprint, "radius_scale": scale})
    return elevation, scale


# ---------------------------------------------------------------------------------- colour
def albedo_lut(gamma):
    """The colour pipeline as a 256-entry byte table (data_loader.py:272-287): source byte v -> albedo
    0.2 + 0.75 v/255 -> ^gamma (the inverse of the Gamma post-process) -> byte, all in float32, truncating."""
    v = np.arange(256, dtype=np.float32)
    a = np.float32(ALBEDO_MIN) + np.float32(ALBEDO_RANGE / 255) * v
    return (np.power(a, np.float32(gamma), dtype=np.float32) * np.float32(255)).astype(np.uint8)


def moon_texture(rgb, gamma, order="RGB"):
    """(H, W, 3) bytes -> (H, W, 4) RGBA texture through the LUT, alpha 255 (data_loader.py:345-368)."""
    lut = albedo_lut(gamma)
    src = np.asarray(rgb)
    if src.ndim == 2:
        src = np.repeat(src[..., None], 3, axis=2)
    idx = (0, 1, 2) if order == "RGB" else (2, 1, 0)
    out = np.empty(src.shape[:2] + (4,), np.uint8)
    for y in range(0, src.shape[0], 1024):
        band = src[y:y + 1024]
        for k, ch in enumerate(idx):
            out[y:y + 1024, :, k] = lut[band[..., ch]]
    out[..., 3] = 255
    return out


def _open_rgb(filepath, downscale=1):
    try:
        a = read_image(filepath)                      # uncompressed TIFF: zero-copy
        if a.dtype != np.uint8:
            a = (a >> (8 * (a.dtype.itemsize - 1))).astype(np.uint8)
        if a.ndim == 2:                                # greyscale map: three equal channels, like cv2.imread gives the reference
            a = np.repeat(a[..., None], 3, axis=2)
        if downscale > 1:
            h, w = a.shape[0] // downscale, a.shape[1] // downscale
            a = a[:h * downscale, :w * downscale].reshape(h, downscale, w, downscale, -1).mean((1, 3)).astype(np.uint8)
        return a
    except (TiffError, KeyError, struct.error):
        from PIL import Image
        Image.MAX_IMAGE_PIXELS = None
        im = Image.open(filepath)
        if downscale > 1:
            im = im.reduce(downscale)
        return np.asarray(im.convert("RGB"))


def load_color_data(filepath, gamma=2.2, downscale=1):
    """data_loader.py:290-342 -> RGBA bytes ready for set_texture_2d.  The `.ds<N>.npy` cache of the reference holds
    BGR bytes (cv2 order); it is read as such, and written in the same order."""
    cache_base = f"{filepath}.ds{downscale}"
    if downscale > 1:
        fingerprint = cache_fingerprint(filepath, downscale=downscale)
        if cache_meta(cache_base, fingerprint) is not None:
            try:
                return moon_texture(np.load(cache_base + ".npy"), gamma, order="BGR")
            except Exception:
                pass
    if not os.path.isfile(filepath):
        raise FileNotFoundError(f"Color file not found: {filepath}, and no cache of it downscaled by {downscale} beside it.")
    rgb = _open_rgb(filepath, downscale)
    if downscale > 1:
        _save_cache(cache_base, np.ascontiguousarray(rgb[..., ::-1]), cache_fingerprint(filepath, downscale=downscale))
    return moon_texture(rgb, gamma, order="RGB")


def load_starmap(filepath, target_width):
    """data_loader.py:371-425 -> float32 (h, w, 3) in [0, 1], or None."""
    if not os.path.isfile(filepath):
        return None
    cache_base = f"{filepath}.w{target_width}"
    fingerprint = cache_fingerprint(filepath, target_width=target_width)
    if cache_meta(cache_base, fingerprint) is not None:
        try:
            return np.load(cache_base + ".npy")
        except Exception:
            pass
    from PIL import Image
    Image.MAX_IMAGE_PIXELS = None
    star = np.asarray(Image.open(filepath).convert("RGB"), np.float32)
    star *= np.float32(1 / 255)
    if target_width < star.shape[1]:
        star = resize_cubic(star, int(star.shape[0] * target_width / star.shape[1]), target_width)
        np.clip(star, 0, 1, out=star)
    _save_cache(cache_base, star, fingerprint)
    return star


def _cubic_taps(n_dst, n_src):
    """Source indices (n_dst, 4) and weights (n_dst, 4) of cv2.resize(..., interpolation=cv2.INTER_CUBIC): pixel-centre
    alignment x_src = (x_dst + 0.5) * n_src / n_dst - 0.5, Keys kernel with a = -0.75, replicated border, no pre-filter."""
    a = np.float32(-0.75)
    x = (np.arange(n_dst, dtype=np.float64) + 0.5) * (n_src / n_dst) - 0.5
    x0 = np.floor(x)
    t = (x - x0).astype(np.float32)
    w = np.empty((n_dst, 4), np.float32)
    w[:, 0] = ((a * (t + 1) - 5 * a) * (t + 1) + 8 * a) * (t + 1) - 4 * a
    w[:, 1] = ((a + 2) * t - (a + 3)) * t * t + 1
    w[:, 2] = ((a + 2) * (1 - t) - (a + 3)) * (1 - t) * (1 - t) + 1
    w[:, 3] = 1 - w[:, 0] - w[:, 1] - w[:, 2]
    idx = np.clip(x0.astype(np.int64)[:, None] + np.arange(-1, 3)[None, :], 0, n_src - 1)
    return idx, w


def resize_cubic(img, out_h, out_w):
    """float32 (h, w, c) -> (out_h, out_w, c), the rule data_loader.py:415-416 applies through cv2.resize
    (INTER_CUBIC on float32 data): separable, columns first, in bands to bound the temporaries."""
    src = np.asarray(img, np.float32)
    ci, cw = _cubic_taps(out_w, src.shape[1])
    ri, rw = _cubic_taps(out_h, src.shape[0])
    out = np.empty((out_h, out_w) + src.shape[2:], np.float32)
    for y0 in range(0, out_h, 256):
        y1 = min(out_h, y0 + 256)
        rows = np.unique(ri[y0:y1])
        band = src[rows]                                             # the source rows this band needs
        hb = sum(band[:, ci[:, k]] * cw[:, k].reshape((1, -1) + (1,) * (src.ndim - 2)) for k in range(4))
        pos = np.searchsorted(rows, ri[y0:y1])
        out[y0:y1] = sum(hb[pos[:, k]] * rw[y0:y1, k].reshape((-1, 1) + (1,) * (src.ndim - 2)) for k in range(4))
    return out
