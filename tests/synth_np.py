"""Small numpy synthetic inputs for CPU-side tests (no GPU): LDEM-like int16 sources, DEMs, colour maps."""
import numpy as np


def _smooth_noise(rng, h, w, cells):
    """Bilinear upsample of a (cells, 2*cells) random grid, periodic in longitude."""
    g = rng.standard_normal((cells + 1, 2 * cells)).astype(np.float64)
    g = np.concatenate([g, g[:, :1]], axis=1)
    r = np.linspace(0, cells, h, endpoint=False)
    c = np.linspace(0, 2 * cells, w, endpoint=False)
    r0 = np.floor(r).astype(int); c0 = np.floor(c).astype(int)
    fr = (r - r0)[:, None]; fc = (c - c0)[None, :]
    fr = fr * fr * (3 - 2 * fr); fc = fc * fc * (3 - 2 * fc)
    a = g[r0][:, c0]; b = g[r0][:, c0 + 1]; cc = g[r0 + 1][:, c0]; d = g[r0 + 1][:, c0 + 1]
    return (a * (1 - fc) + b * fc) * (1 - fr) + (cc * (1 - fc) + d * fc) * fr


def ldem_source(h, w, seed=7, craters=40):
    """int16 (h, w) LDEM-like heights (0.5 m units) in the LOLA-like range."""
    rng = np.random.default_rng(seed)
    km = np.zeros((h, w))
    amp, cells = 2.5, 2
    while cells * 2 <= max(4, h // 2):
        km += amp * _smooth_noise(rng, h, w, cells)
        amp *= 0.55; cells *= 2
    lat = (0.5 - (np.arange(h) + 0.5) / h) * np.pi
    lon = ((np.arange(w) + 0.5) / w - 0.5) * 2 * np.pi
    cl = np.cos(lat)[:, None]
    p = np.stack([cl * np.sin(lon)[None, :], -cl * np.cos(lon)[None, :], np.sin(lat)[:, None] * np.ones((1, w))], -1)
    for _ in range(craters):
        q = rng.standard_normal(3); q /= np.linalg.norm(q)
        rad = rng.uniform(0.02, 0.15)
        t = np.linalg.norm(p - q, axis=-1) / rad
        depth = min(0.4 * rad * 1737.4, 2.5 + 0.006 * rad * 1737.4)
        km += np.where(t < 1, -depth * (1 - t * t) + 0.22 * depth * t * t,
                       np.where(t < 1.4, 0.22 * depth * ((1.4 - t) * 2.5) ** 2, 0.0))
    return np.clip(np.rint(km * 2000.0), -18200, 21600).astype(np.int16)


def dem_from_source(src):
    """float32 displacement factors from an int16 source at downscale 1 (data_loader.py:216-242 arithmetic)."""
    e = src.astype(np.float32) * np.float32(0.5 / 1737400.0)
    e += np.float32(1.0)
    scale = float(e.max())
    e /= np.float32(scale)
    return e, scale


def dem(h, w, seed=7, craters=40):
    return dem_from_source(ldem_source(h, w, seed, craters))[0]


def colour_map(h, w, seed=11):
    rng = np.random.default_rng(seed)
    v = 0.5 + 0.25 * _smooth_noise(rng, h, w, 4) + 0.12 * _smooth_noise(rng, h, w, 16)
    b = np.clip(v * 255, 0, 255).astype(np.uint8)
    out = np.empty((h, w, 4), np.uint8)
    out[..., 0] = b; out[..., 1] = (b * 0.97).astype(np.uint8); out[..., 2] = (b * 0.9).astype(np.uint8); out[..., 3] = 255
    return out


def cone_dem(h, w, lat_deg, lon_deg, height_km, base_km):
    """Flat sphere with one cone peak: analytic known-answer scene (shadow length = h / tan(alt))."""
    lat = (0.5 - (np.arange(h) + 0.5) / h) * np.pi
    lon = ((np.arange(w) + 0.5) / w - 0.5) * 2 * np.pi
    la0, lo0 = np.radians(lat_deg), np.radians(lon_deg)
    cosd = (np.sin(lat)[:, None] * np.sin(la0) + np.cos(lat)[:, None] * np.cos(la0) * np.cos(lon[None, :] - lo0))
    dist_km = np.arccos(np.clip(cosd, -1, 1)) * 1737.4
    hk = np.clip(1.0 - dist_km / base_km, 0, None) * height_km
    e = (1.0 + hk / 1737.4).astype(np.float32)
    return e / e.max()


def corrugated_dem(h, w, amplitude_km=3.0, wavelength_km=40.0):
    """Egg-crate relief with steep (tens of degrees) slopes: surfaces see each other, so inter-reflection exists."""
    lat = (0.5 - (np.arange(h) + 0.5) / h) * np.pi
    lon = ((np.arange(w) + 0.5) / w - 0.5) * 2 * np.pi
    k = 2 * np.pi * 1737.4 / wavelength_km
    hk = amplitude_km * np.sin(k * lat)[:, None] * np.sin(k * lon)[None, :]
    e = (1.0 + hk / 1737.4).astype(np.float32)
    return e / e.max()
