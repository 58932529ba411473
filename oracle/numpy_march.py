"""numpy ray-march of the same scene: the INDEPENDENT statistical check of the C oracle
(TEST INFRASTRUCTURE, NOT PRODUCT -- see oracle/mrtx_oracle.c for the parity status).

Deliberately written with library math (np.arctan2, np.sqrt, float64 where convenient) and no attempt at
bit-compatibility: it restates the *model* (SURVEY.md section 2.1 D1-D5, D9), not the arithmetic spec, so
an error shared by the C oracle and the HIP kernels (which follow one spec) would show up against it.
Vectorised over all samples of a tile with a masked `while` over march steps, as BASELINE.md section 3 plans.
"""
import numpy as np


def _dem_bilinear(dem, lat, lon):
    """renderer_navigation.py:575-593 convention, vectorised, float64."""
    h, w = dem.shape
    row = (np.pi / 2 - lat) / np.pi * h - 0.5
    col = ((lon + np.pi) / (2 * np.pi) * w - 0.5) % w
    r0 = np.clip(np.floor(row), 0, h - 2).astype(np.int64)
    fr = np.clip(row - r0, 0.0, 1.0)
    c0 = np.floor(col).astype(np.int64) % w
    c1 = (c0 + 1) % w
    fc = col - np.floor(col)
    f64 = np.float64
    return (dem[r0, c0].astype(f64) * (1 - fr) * (1 - fc) + dem[r0 + 1, c0].astype(f64) * fr * (1 - fc)
            + dem[r0, c1].astype(f64) * (1 - fr) * fc + dem[r0 + 1, c1].astype(f64) * fr * fc)


def _height(dem, p):
    rho = np.hypot(p[..., 0], p[..., 1])
    return _dem_bilinear(dem, np.arctan2(p[..., 2], rho), np.arctan2(p[..., 0], p[..., 1]))


def _below(dem, R, p):
    return np.sqrt((p * p).sum(-1)) <= R * _height(dem, p)


def _mix32(x):
    """lowbias32 on uint32 arrays (the spec's integer hash)."""
    x = x.astype(np.uint32)
    x ^= x >> np.uint32(16); x *= np.uint32(0x7feb352d)
    x ^= x >> np.uint32(15); x *= np.uint32(0x846ca68b)
    x ^= x >> np.uint32(16)
    return x


def spec_uniforms(scene, gs, dims=4):
    """The spec's counter-based uniforms u_d(pixel, sample gs), d < dims, as (dims, H, W) float64 arrays
    (u_d = (mix32(ks + (d+1)*0x9E3779B9) >> 8) * 2^-24)."""
    W, H = scene.width, scene.height
    with np.errstate(over="ignore"):
        pix = (np.arange(H, dtype=np.uint32)[:, None] * np.uint32(W) + np.arange(W, dtype=np.uint32)[None, :])
        key0 = _mix32(np.array([np.uint32(scene.seed) ^ np.uint32(0x9E3779B9)], np.uint32))[0]
        kp = _mix32(pix + key0)
        ks = _mix32(kp ^ np.uint32((gs * 0x85EBCA6B + 1) & 0xFFFFFFFF))
        return [(_mix32(ks + np.uint32(((d + 1) * 0x9E3779B9) & 0xFFFFFFFF)) >> np.uint32(8)).astype(np.float64) * 2.0 ** -24
                for d in range(dims)]


def _duff_basis(n):
    """Duff et al., "Building an Orthonormal Basis, Revisited" (the spec's basis about a unit vector)."""
    sg = np.where(n[:, 2] >= 0, 1.0, -1.0)
    a = -1.0 / (sg + n[:, 2])
    b = n[:, 0] * n[:, 1] * a
    b1 = np.stack([1.0 + sg * n[:, 0] * n[:, 0] * a, sg * b, -sg * n[:, 0]], -1)
    b2 = np.stack([b, sg + n[:, 1] * n[:, 1] * a, -n[:, 1]], -1)
    return b1, b2


def render(scene, dem, spp, seed=1234, albedo=None, spec_rng=False, region=None, counts=None):
    """Mean linear radiance (H, W, 3) with `spp` jittered samples per pixel.

    `region` = (x0, y0, x1, y1): only these pixels of the frame (the result has their shape) -- bench.py's numpy CPU baseline
    times row bands of a crop in parallel processes.  `counts`: a dict that receives "rays" and "dem_samples" (bilinear DEM
    evaluations performed).  The DEM is used as it comes (float32 stays float32 in memory: a cfg3 DEM is 4.25 GB; every fetched
    value is widened to float64 for the arithmetic, which is what converting the whole array first did).

    spec_rng=False: own RNG (statistical check).  spec_rng=True: the spec's uniforms for (pixel, sample) and its light-cone
    parameterisation (Duff basis), still in float64 with library trig and exact evaluation at every step -- the SAME rays up
    to rounding, so the comparison with the oracle becomes per pixel: only samples that sit on a hit / shadow decision
    boundary may differ."""
    dem = np.asarray(dem)
    n_dem = [0]
    W, H, R = scene.width, scene.height, float(scene.radius)
    x0, y0, x1, y1 = region if region is not None else (0, 0, W, H)
    rng = np.random.default_rng(seed)
    eye, tgt, up = (np.asarray(v, float) for v in (scene.eye, scene.target, scene.up))
    wv = tgt - eye; wv /= np.linalg.norm(wv)
    uv = np.cross(wv, up); uv /= np.linalg.norm(uv)
    vv = np.cross(uv, wv)
    th = np.tan(np.radians(scene.vfov_deg) / 2)
    ez = np.asarray(scene.u, float); ez /= np.linalg.norm(ez)
    v0 = np.asarray(scene.v, float); v0 = v0 - (v0 @ ez) * ez; v0 /= np.linalg.norm(v0)
    M = np.stack([np.cross(ez, v0), v0, ez])                     # scene -> (east90, lon0, north)
    centre = np.asarray(scene.center, float)
    Lb = M @ (np.asarray(scene.light_pos, float) - centre)
    alb = np.asarray(scene.const_albedo if albedo is None else albedo, float)
    step, eps, seps = scene.marching_step, scene.marching_step_eps, scene.scene_epsilon
    out = np.zeros((y1 - y0, x1 - x0, 3))
    ys, xs = np.mgrid[y0:y1, x0:x1]
    for gs in range(spp):
        if spec_rng:
            u0, u1, u2, u3 = (u[y0:y1, x0:x1] for u in spec_uniforms(scene, gs))
        else:
            u0, u1, u2, u3 = (rng.random((y1 - y0, x1 - x0)) for _ in range(4))
        fx = xs + u0; fy = ys + u1
        sx = (fx / W * 2 - 1) * th * W / H; sy = (1 - fy / H * 2) * th
        d = wv + sx[..., None] * uv + sy[..., None] * vv
        d /= np.linalg.norm(d, axis=-1, keepdims=True)
        oc = eye - centre
        b = d @ oc; c = oc @ oc - R * R
        disc = b * b - c
        ok = disc > 0
        sq = np.sqrt(np.where(ok, disc, 0))
        t0 = np.maximum(-b - sq, 0); t1 = -b + sq
        ok &= t1 > 0
        idx = np.argwhere(ok)
        if idx.size == 0:
            continue
        dd = d[ok] @ M.T
        pe = (oc + t0[ok][:, None] * d[ok]) @ M.T
        smax = (t1 - t0)[ok]
        n = len(pe)
        active = np.ones(n, bool); hit = np.zeros(n, bool)
        lo = np.zeros(n); hi = np.zeros(n)
        k = 1
        while active.any():
            s = k * step
            active &= s <= smax
            a = np.flatnonzero(active)
            if a.size == 0:
                break
            bel = _below(dem, R, pe[a] + s * dd[a]); n_dem[0] += a.size
            h_ = a[bel]
            hit[h_] = True; hi[h_] = s; lo[h_] = (k - 1) * step
            active[h_] = False
            k += 1
        hh = np.flatnonzero(hit)
        if hh.size == 0:
            continue
        width = step
        while width > eps:
            mid = 0.5 * (lo[hh] + hi[hh])
            bel = _below(dem, R, pe[hh] + mid[:, None] * dd[hh]); n_dem[0] += hh.size
            hi[hh] = np.where(bel, mid, hi[hh]); lo[hh] = np.where(bel, lo[hh], mid)
            width *= 0.5
        p = pe[hh] + lo[hh][:, None] * dd[hh]
        # normal from central differences of D in texel units
        hgt, wid = dem.shape
        r = np.linalg.norm(p, axis=-1); rho = np.maximum(np.hypot(p[:, 0], p[:, 1]), 1e-6)
        lat = np.arctan2(p[:, 2], rho); lon = np.arctan2(p[:, 0], p[:, 1])
        dla, dlo = np.pi / hgt, 2 * np.pi / wid
        dlat = (_dem_bilinear(dem, np.clip(lat + dla, -np.pi / 2, np.pi / 2), lon)
                - _dem_bilinear(dem, np.clip(lat - dla, -np.pi / 2, np.pi / 2), lon)) / (2 * dla)
        dlon = (_dem_bilinear(dem, lat, lon + dlo) - _dem_bilinear(dem, lat, lon - dlo)) / (2 * dlo)
        n_dem[0] += 4 * hh.size
        rhat = p / r[:, None]
        sphi, cphi, slam, clam = p[:, 2] / r, rho / r, p[:, 0] / rho, p[:, 1] / rho
        north = np.stack([-sphi * slam, -sphi * clam, cphi], -1)
        east = np.stack([clam, -slam, np.zeros_like(clam)], -1)
        nrm = rhat - (R / r * dlat)[:, None] * north - (R / rho * dlon)[:, None] * east
        nrm /= np.linalg.norm(nrm, axis=-1, keepdims=True)
        # one uniform sample of the cone the light sphere subtends
        o = p + seps * nrm
        tl = Lb - o; dist = np.linalg.norm(tl, axis=-1); ld = tl / dist[:, None]
        sin2 = np.minimum((scene.light_radius / dist) ** 2, 1.0)
        omc = sin2 / (1 + np.sqrt(1 - sin2))
        pix = idx[hh]
        ct = 1 - u2[pix[:, 0], pix[:, 1]] * omc; st = np.sqrt(np.maximum(0, 1 - ct * ct))
        ph = 2 * np.pi * u3[pix[:, 0], pix[:, 1]]
        if spec_rng:
            b1, b2 = _duff_basis(ld)
        else:
            hlp = np.where(np.abs(ld[:, [2]]) < 0.9, [[0, 0, 1.0]], [[1.0, 0, 0]])
            b1 = np.cross(hlp, ld); b1 /= np.linalg.norm(b1, axis=-1, keepdims=True)
            b2 = np.cross(ld, b1)
        wi = st[:, None] * (np.cos(ph)[:, None] * b1 + np.sin(ph)[:, None] * b2) + ct[:, None] * ld
        cosi = (nrm * wi).sum(-1)
        lit = cosi > 0
        act = lit.copy()
        k = 1
        while act.any():
            a = np.flatnonzero(act)
            q = o[a] + (k * step) * wi[a]
            outside = (q * q).sum(-1) > R * R
            act[a[outside]] = False
            a = a[~outside]
            if a.size:
                bel = _below(dem, R, o[a] + (k * step) * wi[a]); n_dem[0] += a.size
                lit[a[bel]] = False; act[a[bel]] = False
            k += 1
        wgt = np.where(lit, 2 * scene.light_radiance * omc * cosi, 0.0)
        np.add.at(out, (pix[:, 0], pix[:, 1]), wgt[:, None] * alb[None, :])
    if counts is not None:
        counts["rays"] = counts.get("rays", 0) + (y1 - y0) * (x1 - x0) * spp
        counts["dem_samples"] = counts.get("dem_samples", 0) + n_dem[0]
    return out / spp
