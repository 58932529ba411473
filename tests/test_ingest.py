"""Ingest row: TIFF/BigTIFF reader, the reference's cache format, the albedo LUT,
and -- on the GPU -- load_elevation_data against the oracle's restatement of data_loader.py:166-247."""
import json
import os
import struct

import numpy as np
import pytest

import synth_np
from moonrtx_amd import ingest


def write_tiff(path, arr, big=False, rows_per_strip=None, gap=0):
    """Minimal uncompressed little-endian strip writer (classic or BigTIFF) for tests."""
    arr = np.ascontiguousarray(arr)
    h, w = arr.shape[:2]
    spp = 1 if arr.ndim == 2 else arr.shape[2]
    bits = arr.dtype.itemsize * 8
    fmt = 2 if arr.dtype.kind == "i" else 1
    rps = rows_per_strip or h
    strips = [arr[r:r + rps].tobytes() for r in range(0, h, rps)]
    hdr = 16 if big else 8
    offs, pos, body = [], hdr, b""
    for s in strips:
        offs.append(pos + len(body)); body += s + b"\0" * gap
    ifd_off = hdr + len(body)
    otype, osz = (16, 8) if big else (4, 4)
    entries = [(256, 4, [w]), (257, 4, [h]), (258, 3, [bits] * spp), (259, 3, [1]), (262, 3, [1 if spp == 1 else 2]),
               (273, otype, offs), (277, 3, [spp]), (278, 4, [rps]), (279, otype, [len(s) for s in strips]), (284, 3, [1]),
               (339, 3, [fmt] * spp)]
    tfmt = {3: "H", 4: "I", 16: "Q"}
    esz, cfmt, inline = (20, "Q", 8) if big else (12, "I", 4)
    n = len(entries)
    extra_off = ifd_off + (8 if big else 2) + n * esz + (8 if big else 4)
    ifd, extra = b"", b""
    for tag, typ, vals in entries:
        data = struct.pack("<" + tfmt[typ] * len(vals), *vals)
        if len(data) <= inline:
            field = data.ljust(inline, b"\0")
        else:
            field = struct.pack("<" + cfmt, extra_off + len(extra)); extra += data
        ifd += struct.pack("<HH" + cfmt, tag, typ, len(vals)) + field
    with open(path, "wb") as f:
        if big:
            f.write(b"II" + struct.pack("<HHHQ", 43, 8, 0, ifd_off))
        else:
            f.write(b"II" + struct.pack("<HI", 42, ifd_off))
        f.write(body)
        f.write(struct.pack("<" + ("Q" if big else "H"), n) + ifd + struct.pack("<" + ("Q" if big else "I"), 0) + extra)


@pytest.mark.parametrize("big", [False, True])
@pytest.mark.parametrize("rps,gap", [(None, 0), (7, 0), (5, 3)])
def test_read_image_strips(tmp_path, big, rps, gap):
    src = synth_np.ldem_source(48, 96, seed=3, craters=4)
    p = str(tmp_path / "ldem.tif")
    write_tiff(p, src.view(np.uint16), big=big, rows_per_strip=rps, gap=gap)
    got = ingest.read_image(p)
    assert got.dtype == np.uint16 and got.shape == (48, 96)
    got = got.copy(); got.dtype = np.int16                    # what data_loader.py:215 does
    assert np.array_equal(got, src)


def test_read_image_matches_pillow_and_rejects_compressed(tmp_path):
    from PIL import Image
    rgb = synth_np.colour_map(40, 80)[..., :3]
    p = str(tmp_path / "c.tif")
    Image.fromarray(rgb).save(p)                                # Pillow: uncompressed strips by default
    assert np.array_equal(ingest.read_image(p), rgb)
    q = str(tmp_path / "lzw.tif")
    Image.fromarray(rgb).save(q, compression="tiff_lzw")
    with pytest.raises(ingest.TiffError, match="compress"):
        ingest.read_image(q)
    tex = ingest.load_color_data(q, gamma=2.2)                  # falls back to Pillow for compressed maps
    assert tex.shape == (40, 80, 4) and (tex[..., 3] == 255).all()
    assert np.array_equal(tex, ingest.moon_texture(rgb, 2.2))
    with pytest.raises(ingest.TiffError):
        open(str(tmp_path / "x.tif"), "wb").write(b"not a tiff at all")
        ingest.read_image(str(tmp_path / "x.tif"))


def test_albedo_lut_known_answers():
    """data_loader.py:261-287: dark maria 0.2, brightest highlands 0.95; lut[v] = uint8(255 (0.2+0.75 v/255)^gamma)."""
    for g in (0.5, 2.2, 5.0):
        lut = ingest.albedo_lut(g)
        assert lut.dtype == np.uint8 and lut.shape == (256,) and np.all(np.diff(lut.astype(int)) >= 0)
        assert lut[0] == int(np.float32(255) * np.power(np.float32(0.2), np.float32(g), dtype=np.float32))
        assert lut[255] == int(np.float32(255) * np.power(np.float32(0.2) + np.float32(0.75 / 255) * np.float32(255), np.float32(g)))
        full = (255 * (0.2 + 0.75 / 255 * np.arange(256)) ** g)
        assert np.abs(lut.astype(float) - np.floor(full)).max() <= 1      # truncating cast, float32 vs float64
    assert ingest.albedo_lut(2.2)[128] == 75                                # the bench's grey (SURVEY 8(d) cfg 1)
    rgb = np.stack([np.arange(256, dtype=np.uint8)] * 3, -1)[None]
    tex = ingest.moon_texture(rgb[..., ::-1], 2.2, order="BGR")
    assert np.array_equal(tex[0, :, 0], ingest.albedo_lut(2.2)) and (tex[..., 3] == 255).all()


def test_cache_format_is_the_references(tmp_path):
    """data_loader.py:22-95: <src>.ds<N>.npy + .json {version, downscale, source_size, source_mtime, radius_scale}."""
    src_path = str(tmp_path / "ldem.tif")
    write_tiff(src_path, np.zeros((8, 16), np.uint16))
    assert not ingest.downscale_cache_available(src_path, 2)
    fp = ingest.cache_fingerprint(src_path, downscale=2)
    assert fp == {"version": 1, "downscale": 2, "source_size": os.path.getsize(src_path),
                  "source_mtime": int(os.path.getmtime(src_path))}
    elev = np.linspace(0.99, 1.0, 32, dtype=np.float32).reshape(4, 8)
    np.save(src_path + ".ds2.npy", elev)
    json.dump({**fp, "radius_scale": 1.0062}, open(src_path + ".ds2.json", "w"))
    assert ingest.downscale_cache_available(src_path, 2) and not ingest.downscale_cache_available(src_path, 1)
    got, scale = ingest.load_elevation_data(src_path, 2)          # served from the cache: no GPU involved
    assert np.array_equal(got, elev) and scale == 1.0062
    os.remove(src_path)                                            # source deleted: cache taken on trust
    assert ingest.downscale_cache_available(src_path, 2)
    json.dump({**fp, "version": 0, "radius_scale": 1.0}, open(src_path + ".ds2.json", "w"))
    assert not ingest.downscale_cache_available(src_path, 2)
    with pytest.raises(FileNotFoundError):
        ingest.load_elevation_data(src_path, 2)


@pytest.mark.gpu
@pytest.mark.parametrize("d", [1, 2, 3, 8])
def test_load_elevation_data_on_device_matches_oracle(native_lib, tmp_path, d):
    from oracle import orc
    src = synth_np.ldem_source(24 * d + (d > 1), 40 * d + (d > 1), seed=d, craters=3)   # ragged: trailing row/col dropped
    p = str(tmp_path / "ldem_big.tif")
    write_tiff(p, src.view(np.uint16), big=True, rows_per_strip=5)
    got, scale = ingest.load_elevation_data(p, d)
    want, wscale = orc.dem_from_ldem(src, d)
    assert got.shape == want.shape == (src.shape[0] // d, src.shape[1] // d)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and np.float32(scale) == np.float32(wscale)
    assert got.max() == 1.0
    if d > 1:                                                      # second call is served by the cache it wrote
        meta = json.load(open(p + f".ds{d}.json"))
        assert meta["downscale"] == d and meta["version"] == 1 and abs(meta["radius_scale"] - scale) < 1e-9
        again, scale2 = ingest.load_elevation_data(p, d)
        assert np.array_equal(again, got) and scale2 == scale


def test_load_starmap_resize_rule_clip_and_cache(tmp_path):
    """data_loader.py:371-425: BGR->RGB, /255, cubic resize (cv2.INTER_CUBIC: pixel-centre alignment, Keys kernel a = -0.75,
    no pre-filter) only when the source is wider than the target, clip to [0, 1], `.w<N>.npy` + `.json` cache."""
    from PIL import Image
    from moonrtx_amd import ingest
    h, w = 24, 48
    ramp = np.tile(np.linspace(20, 220, w).round().astype(np.uint8)[None, :, None], (h, 1, 3))
    ramp[:, :, 1] = 128                                       # a constant channel
    ramp[:, 30:, 2] = 255; ramp[:, :30, 2] = 0               # a step: cubic overshoot must be clipped
    p = tmp_path / "stars.tif"
    Image.fromarray(ramp).save(p)
    same = ingest.load_starmap(str(p), 96)                    # target wider than the source: no resize
    assert same.shape == (h, w, 3) and same.dtype == np.float32
    assert np.array_equal(same, ramp.astype(np.float32) * np.float32(1 / 255))
    half = ingest.load_starmap(str(p), 24)
    assert half.shape == (int(h * 24 / w), 24, 3) and half.dtype == np.float32
    assert np.allclose(half[..., 1], 128 / 255, atol=1e-6)                            # constants survive
    want = (np.arange(24) + 0.5) * 2 - 0.5                                            # source x of every output column
    lin = (20 + want * (200 / (w - 1))) / 255
    assert np.abs(half[5, 2:-2, 0] - lin[2:-2]).max() < 2.5e-3                        # cubic convolution reproduces a ramp
    assert half.min() >= 0.0 and half.max() <= 1.0 and half[5, 14, 2] < half[5, 15, 2]  # clipped, step kept
    # the rule itself, against the 4-tap formula evaluated by hand at one output pixel (row 3, column 10, channel 0)
    src = ramp.astype(np.float32) * np.float32(1 / 255)
    def taps(xd, n_dst, n_src):
        x = (xd + 0.5) * n_src / n_dst - 0.5; x0 = int(np.floor(x)); t = x - x0; a = -0.75
        wts = [((a * (t + 1) - 5 * a) * (t + 1) + 8 * a) * (t + 1) - 4 * a, ((a + 2) * t - (a + 3)) * t * t + 1,
               ((a + 2) * (1 - t) - (a + 3)) * (1 - t) * (1 - t) + 1]
        wts.append(1 - sum(wts))
        return [min(max(x0 - 1 + k, 0), n_src - 1) for k in range(4)], wts
    ci, cw = taps(10, 24, w); ri, rw = taps(3, 12, h)
    by_hand = sum(rw[a] * sum(cw[b] * src[ri[a], ci[b], 0] for b in range(4)) for a in range(4))
    assert abs(by_hand - half[3, 10, 0]) < 1e-6
    # cache round trip: the second call must not touch the source
    assert (tmp_path / "stars.tif.w24.npy").is_file() and (tmp_path / "stars.tif.w24.json").is_file()
    p.write_bytes(p.read_bytes())                             # same bytes, new mtime: fingerprint changes -> recomputed
    again = ingest.load_starmap(str(p), 24)
    assert np.array_equal(again, half)
    assert ingest.load_starmap(str(tmp_path / "missing.tif"), 24) is None


def test_greyscale_colour_map_with_downscale_and_cache(tmp_path):
    """A 2-D (greyscale) colour map: block-mean downscale applies, the cache holds three equal channels (no east-west
    mirroring), and the cached load equals the fresh one."""
    from PIL import Image
    from moonrtx_amd import ingest
    g = (np.arange(16 * 32).reshape(16, 32) % 251).astype(np.uint8)
    p = tmp_path / "grey.tif"
    Image.fromarray(g).save(p)
    a = ingest.load_color_data(str(p), gamma=2.2, downscale=2)
    assert a.shape == (8, 16, 4) and (a[..., 3] == 255).all()
    assert np.array_equal(a[..., 0], a[..., 1]) and np.array_equal(a[..., 1], a[..., 2])
    mean = g.reshape(8, 2, 16, 2).mean((1, 3)).astype(np.uint8)
    assert np.array_equal(a[..., 0], ingest.albedo_lut(2.2)[mean])          # block mean, then the LUT; not mirrored
    cached = np.load(str(p) + ".ds2.npy")
    assert cached.shape == (8, 16, 3)
    b = ingest.load_color_data(str(p), gamma=2.2, downscale=2)
    assert np.array_equal(a, b)
