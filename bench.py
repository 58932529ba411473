#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MoonRTX hot path on MI355X.

Metric (BASELINE.json): Mrays/s + ms/frame at 3840x2160, 64 spp, --downscale-2-sized DEM; 1 ray = 1 primary
camera sample.  A "step" is one full frame of the workload the reference itself runs --
`path_seg_range (2, 4)` (moon_renderer.py:583): restart the accumulation cycle (`refresh_scene`), render all samples of
every pixel (camera ray, first vertex + its direct light in render_kernel, everything after the first vertex in
path_kernel, per-pixel sums in resolve_paths_kernel) and -- on N > 1 GPUs -- gather the tiles to rank 0.
Inputs (synthetic LOLA-like DEM + colour map) are generated on the device and are
resident in HBM before the timed region.  The direct-light-only frame (path_seg_range (1, 1), round 1's headline) is
measured beside it and reported under "also".

  python bench.py --gpus 1 --steps 5 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.
"""
import argparse
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)

WORKLOADS = {
    # name: (W, H, spp, dem_h, dem_w, colour (h, w) or None)
    "cfg1": (512, 512, 1, 5760, 11520, None),
    "cfg2": (1920, 1080, 16, 11520, 23040, None),
    "cfg3": (3840, 2160, 64, 23040, 46080, (13680, 27360)),
    "cfg4": (7680, 4320, 256, 46080, 92160, (13680, 27360)),
}
COUNT_KEYS = ("primary_rays", "primary_hits", "shadow_rays", "bounce_rays", "height_samples", "colour_fetches",
              "background_fetches", "dem_fetches", "mip_fetches", "camera_height_samples", "camera_dem_fetches",
              "camera_mip_fetches", "camera_colour_fetches", "camera_background_fetches")


def synth_starmap(h, w, seed=0x53544152, n_stars=400000):
    """RGBA8 stand-in for the reference's starmap_16k.tif (main.py:36, bound by moon_renderer.py:604-607): black sky, a dim
    milky band, stars of random colour and brightness."""
    import numpy as np
    rng = np.random.default_rng(seed)
    img = np.zeros((h, w, 4), np.uint8)
    band = (6.0 * np.exp(-((np.arange(h) - h / 2) / (h / 14)) ** 2)).astype(np.uint8)
    img[:, :, :3] = band[:, None, None]
    r, c = rng.integers(0, h, n_stars), rng.integers(0, w, n_stars)
    img[r, c, :3] = np.minimum(255, rng.gamma(0.6, 40.0, (n_stars, 1)) + rng.integers(0, 30, (n_stars, 3))).astype(np.uint8)
    img[:, :, 3] = 255
    return img


def source_hash():
    """Identity of the kernel sources a profile was taken with (the GPU box has no .git)."""
    h = hashlib.sha256()
    for rel in ("moonrtx_amd/csrc/mrtx_kernels.hip", "moonrtx_amd/csrc/mrtx_api.hip", "moonrtx_amd/csrc/mrtx_device.h",
                "moonrtx_amd/csrc/Makefile", "include/moonrt.h"):
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def algorithmic_bytes(st, width, height, nominal=False, pixels=True):
    """16 B per DEM bilinear evaluation, 16 B per colour fetch, 4 B per background texel,
    32 B per pixel (one float4 radiance + one float4 hit write).

    Strict (default): the DEM evaluations the kernels actually PERFORM (`dem_fetches`) plus the max-mip and medium-mip
    texels they read to prove the others unnecessary (4 B each).  nominal=True: the evaluations the march DEFINES
    (`height_samples`, the oracle's count) -- what a kernel without the result-preserving skip would read."""
    dem = st["height_samples"] if nominal else st["dem_fetches"]
    mip = 0 if nominal else st["mip_fetches"]
    return (16 * dem + 4 * mip + 16 * st["colour_fetches"] + 4 * st["background_fetches"]
            + (32 * width * height if pixels else 0))


def cpu_baseline(scene, dem_buf, dem_shape, col_buf, col_shape, frame_stats, budget_s=20.0):
    """Time the CPU oracle (kind "port") on a bounded sample of the same workload, on this host's cores.

    A short probe crop gives the host's DEM-samples/s; the timed sample is then the largest centred crop of
    the SAME frame (same scene, DEM, spp, path length) predicted to fit `budget_s` -- the whole frame when it fits.  The
    rate is taken in DEM samples per second and converted to whole-frame Mrays/s with the frame's own
    deterministic sample counts, so cheap sky pixels are not mis-priced."""
    import numpy as np
    from oracle import orc
    dem = dem_buf.download(np.float32, dem_shape)
    col = col_buf.download(np.uint8, col_shape + (4,)) if col_buf is not None else None
    host_cores = os.cpu_count() or 1
    threads = orc.set_threads(min(host_cores, 16))   # the box's CPU share for one GPU
    W, H = scene.width, scene.height

    def crop(frac):
        w, h = max(16, int(W * frac)), max(16, int(H * frac))
        x0, y0 = (W - w) // 2, (H - h) // 2
        return (x0, y0, x0 + w, y0 + h)

    def run(region):
        o = orc.Oracle(scene, dem, col)
        t = time.perf_counter()
        st = o.render(1, region)
        return st, time.perf_counter() - t

    st, dt = run(crop(0.06))                                   # probe
    rate = st["height_samples"] / max(dt, 1e-6)
    frac = 1.0
    if frame_stats["height_samples"] / rate > budget_s:        # centre crops are disk-heavy: scale by area
        frac = max(0.06, min(1.0, (budget_s * rate / frame_stats["height_samples"]) ** 0.5 * 0.75))
    region = crop(frac)
    st, dt = run(region)
    hs_per_s = st["height_samples"] / dt
    frame_seconds_on_cpu = frame_stats["height_samples"] / hs_per_s
    whole = frac >= 1.0
    return {
        "value": round(frame_stats["primary_rays"] / frame_seconds_on_cpu / 1e6, 4), "unit": "Mrays/s",
        "cores": threads, "host_cores": host_cores, "kind": "port",
        "sample": (f"oracle/mrtx_oracle.c, OpenMP x{threads} of {host_cores} host cores: " + ("the WHOLE frame" if whole else
                   f"centred {region[2]-region[0]}x{region[3]-region[1]} crop of the frame")
                   + f" at {scene.spp_per_launch} spp, path_seg_range ({scene.path_seg_min}, {scene.path_seg_max}), same scene/DEM/colour map: "
                   f"{st['primary_rays']} rays, {st['height_samples']} DEM samples in {dt:.2f} s = {hs_per_s/1e6:.1f} M DEM samples/s"
                   + ("" if whole else f"; scaled to the frame's {frame_stats['height_samples']} DEM samples")),
        "seconds": round(dt, 2),
    }


def _numpy_band(job):
    """Worker of cpu_baseline_numpy (a spawned process: numpy only, never the GPU): one row band of a crop."""
    dem_path, scene_vars, spp, region = job
    import types
    import numpy as np
    from oracle import numpy_march
    dem = np.load(dem_path, mmap_mode="r")
    c = {}
    t = time.perf_counter()
    numpy_march.render(types.SimpleNamespace(**scene_vars), dem, spp, spec_rng=True, region=region, counts=c)
    return c["rays"], c["dem_samples"], time.perf_counter() - t


def host_workers():
    """Worker processes for the numpy baseline: the cores this process may use (affinity mask, cgroup quota), capped by
    MOONRT_CPU_WORKERS (default 16 -- a one-GPU box of the pool shares its 256-core host and asks for pools of that size)."""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except (AttributeError, OSError):
        pass
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, int(os.environ.get("MOONRT_CPU_WORKERS", "16"))))


def cpu_baseline_numpy(scene, dem, workers, spp, region, tag):
    """The baseline north_star names: a NUMPY ray-march of the same scene (oracle/numpy_march.py: library trig, float64, every
    step evaluated exactly -- camera ray, bisection, normal, one light sample + marched shadow ray; it has no path continuation,
    so it is timed on the direct-light model), `multiprocessing` over row bands on this host's cores."""
    import multiprocessing as mp
    import tempfile
    import numpy as np
    x0, y0, x1, y1 = region
    bands = min(workers * 2, y1 - y0)
    edges = [y0 + (y1 - y0) * i // bands for i in range(bands + 1)]
    fd, path = tempfile.mkstemp(suffix=".npy", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    os.close(fd)
    try:
        np.save(path, dem)
        sv = {k: v for k, v in vars(scene).items()}
        jobs = [(path, sv, spp, (x0, edges[i], x1, edges[i + 1])) for i in range(bands) if edges[i + 1] > edges[i]]
        with mp.get_context("spawn").Pool(workers) as pool:
            pool.map(_numpy_band, [(path, sv, 1, (x0, y0, min(x1, x0 + 8), min(y1, y0 + 8)))] * workers)   # imports, page-in: untimed
            t = time.perf_counter()
            res = pool.map(_numpy_band, jobs)
            dt = time.perf_counter() - t
    finally:
        os.unlink(path)
    rays = sum(r[0] for r in res)
    dem_samples = sum(r[1] for r in res)
    return {"workload": tag, "rays": int(rays), "dem_samples": int(dem_samples), "seconds": round(dt, 3),
            "mrays_per_s": round(rays / dt / 1e6, 4), "dem_samples_per_s": round(dem_samples / dt, 1), "workers": workers}


def facade_rates(dem_buf, dem_h, dem_w, col_buf, col_shape, W, H, cycles=8):
    """Frames per second THROUGH the PlotOptiX-named surface (moonrtx_amd.tkoptix.TkOptiX.render_cycle: launch(es),
    on_launch_finished callbacks, the tone-mapped RGBA8 image in host memory) for the two cycles the reference alternates
    between (moon_renderer.py:121-129): the 1-frame interactive preview and the 64-frame converged image."""
    from moonrtx_amd.tkoptix import TkOptiX
    from moonrtx_amd.scene import named_scene
    out = {}
    for label, frames, progressive in (("preview_1spp", 1, False), ("converged_64spp", 64, False), ("converged_64spp_progressive", 64, True)):
        if progressive:
            cycles = max(2, cycles // 4)
        rt = TkOptiX(width=W, height=H, start_now=False, progressive=progressive)
        rt.bind_device_inputs(dem_buf, dem_h, dem_w, col_buf, col_shape)
        rt.apply_scene_desc(named_scene("S1", W, H, spp_per_launch=min(frames, 64)))
        rt.set_uint("path_seg_range", 2, 4)
        rt.set_param(min_accumulation_step=1, max_accumulation_frames=frames)
        rt.render_cycle()                                   # warm-up (allocations, mip)
        t = time.perf_counter()
        for _ in range(cycles):
            rt.refresh_scene()
            rt.render_cycle()
        dt = (time.perf_counter() - t) / cycles
        out[label] = {"ms_per_cycle": round(dt * 1e3, 3), "cycles_per_s": round(1.0 / dt, 2)}
        rt.close()
    out["note"] = ("TkOptiX.render_cycle at %dx%d: launches + callbacks + RGBA8 image read back to host memory; the hit buffer "
                   "stays on the device until _get_hit_at asks for it. converged_64spp = this backend's default, ONE launch of 64 "
                   "samples and one on_launch_finished per cycle; converged_64spp_progressive = TkOptiX(progressive=True), "
                   "PlotOptiX's min_accumulation_step=1: 64 launches of one frame, 64 image read-backs, 64 callbacks "
                   "(moon_renderer.py:578, renderer_status.py:239), the same final image. Reference remarks (unnamed RTX GPU, unnamed "
                   "resolution): ~20 preview steps/s, ~1.5 s per converged image (moon_renderer.py:121-129)" % (W, H))
    return out


def numpy_baselines(args, scene, dem_buf, dem_h, dem_w, W, H, dev, direct_counts, frame):
    """BASELINE.md section 3 / SURVEY.md 8(d): the numpy march on all the cores this box grants, cfg1 in full and a 256 x 256 crop
    through the terminator of the timed workload at 4 spp."""
    import copy
    import numpy as np
    from moonrtx_amd.renderer import synth_ldem, dem_from_ldem
    from moonrtx_amd.scene import named_scene
    workers = host_workers()
    out = {"unit": "Mrays/s", "cores": workers, "host_cores": os.cpu_count(), "kind": "port",
           "what": "oracle/numpy_march.py (numpy, float64, library trig, every march step evaluated exactly; camera ray + bisection + "
                   "normal + one light sample with its marched shadow ray: the direct-light model, it has no path continuation), "
                   f"multiprocessing (spawn) over row bands, {workers} worker processes"}
    # (a) cfg1 in full: its own 5760 x 11520 DEM, 512 x 512 x 1 spp
    w1, h1, spp1, dh1, dw1, _ = WORKLOADS["cfg1"]
    src = synth_ldem(dh1, dw1, device=dev)
    d1, _ = dem_from_ldem(src, dh1, dw1, 1, device=dev)
    src.free()
    dem1 = d1.download(np.float32, (dh1, dw1))
    d1.free()
    s1 = named_scene(args.scene, w1, h1, spp_per_launch=1)
    a = cpu_baseline_numpy(s1, dem1, workers, spp1, (0, 0, w1, h1), "cfg1 in full: 512x512, 1 spp, DEM 5760x11520, direct light")
    del dem1
    out["cfg1"] = a
    # (b) the timed workload's own frame: a 256 x 256 crop centred on the terminator, 4 spp
    dem = dem_buf.download(np.float32, (dem_h, dem_w))
    s2 = copy.copy(scene)
    s2.path_seg_min = s2.path_seg_max = 1
    wv = np.asarray(s2.target, float) - np.asarray(s2.eye, float); dist = float(np.linalg.norm(wv)); wv /= dist
    uv = np.cross(wv, np.asarray(s2.up, float)); uv /= np.linalg.norm(uv)
    vv = np.cross(uv, wv)
    ld = np.asarray(s2.light_pos, float) - np.asarray(s2.center, float); ld /= np.linalg.norm(ld)
    r_px = float(s2.radius) / dist / np.tan(np.radians(s2.vfov_deg) / 2) * (H / 2)      # disc radius in pixels
    sd = np.array([ld @ uv, ld @ vv]); sd /= max(1e-9, np.linalg.norm(sd))
    off = -r_px * float(ld @ (-wv))                       # the terminator's centre: cos(phase) disc radii towards the night side
    ctr_dir = (np.asarray(s2.center, float) - np.asarray(s2.eye, float)) / dist
    cx = W / 2 + (ctr_dir @ uv) / np.tan(np.radians(s2.vfov_deg) / 2) * (H / 2) + sd[0] * off
    cy = H / 2 - (ctr_dir @ vv) / np.tan(np.radians(s2.vfov_deg) / 2) * (H / 2) - sd[1] * off
    x0 = int(min(max(0, cx - 128), W - 256)); y0 = int(min(max(0, cy - 128), H - 256))
    b = cpu_baseline_numpy(s2, dem, workers, 4, (x0, y0, x0 + 256, y0 + 256),
                           f"{args.workload}: 256x256 crop at ({x0}, {y0}) through the terminator, 4 spp, DEM {dem_h}x{dem_w}, direct light")
    out["crop"] = b
    # scaled to the whole frame by DEM samples (the crop is all Moon, 64 % of the frame's rays are sky): the frame's direct-light
    # evaluations as the spec defines them / the numpy march's evaluations per second
    ref = direct_counts if direct_counts is not None else frame
    out["value"] = round(frame["primary_rays"] / (ref["height_samples"] / b["dem_samples_per_s"]) / 1e6, 4)
    out["sample"] = (f"whole-frame equivalent of the crop's rate: {ref['height_samples']} DEM evaluations of the "
                     + ("direct-light " if direct_counts is not None else "") + f"frame / {b['dem_samples_per_s'] / 1e6:.2f} M numpy DEM evaluations/s "
                     f"({b['rays']} rays, {b['dem_samples']} evaluations in {b['seconds']} s on {workers} processes); cfg1 in full: "
                     f"{a['mrays_per_s']} Mrays/s ({a['seconds']} s)")
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--scene", default="S1", choices=["S1", "S2", "S3"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the side measurements (direct-only frame, facade rates)")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    ap.add_argument("--dem-scale", type=int, default=1, help="debug: shrink the DEM by this factor")
    ap.add_argument("--path-seg", type=int, nargs=2, default=(2, 4), metavar=("MIN", "MAX"),
                    help="path length in segments; (2,4) is what the reference sets (headline), (1,1) = direct light only")
    ap.add_argument("--zoom", type=float, default=0.0, metavar="VFOV_DEG",
                    help="profiling aid: make the zoomed terminator view (this vertical field of view) the timed scene")
    ap.add_argument("--inwave-paths", action="store_true", help="A/B: keep D6 paths inside the render wave (MRTX_F_INWAVE_PATHS)")
    args = ap.parse_args()

    from moonrtx_amd import build, _lib
    build.build_native()
    _lib.load()
    from moonrtx_amd import dist as mdist
    from moonrtx_amd.renderer import MoonRT, synth_ldem, synth_color, dem_from_ldem
    from moonrtx_amd.scene import named_scene
    import torch

    rank, world, local = mdist.init_process_group()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    dev = local
    torch.cuda.set_device(dev)
    backend = torch.distributed.get_backend() if world > 1 else None
    dist_world = torch.distributed.get_world_size() if world > 1 else 1

    W, H, spp, dem_h, dem_w, col_shape = WORKLOADS[args.workload]
    dem_h //= args.dem_scale
    dem_w //= args.dem_scale
    S = min(spp, 64)
    n_blocks = spp // S
    seg = tuple(args.path_seg)

    # ---- inputs, resident in HBM before anything is timed
    t0 = time.perf_counter()
    src = synth_ldem(dem_h, dem_w, device=dev)
    dem_buf, radius_scale = dem_from_ldem(src, dem_h, dem_w, 1, device=dev)
    src.free()
    col_buf = synth_color(col_shape[0], col_shape[1], device=dev) if col_shape else None
    t_inputs = time.perf_counter() - t0

    scene = named_scene(args.scene, W, H, spp_per_launch=S)
    if args.zoom > 0.0:
        from moonrtx_amd.scene import zoomed_on_terminator
        scene = zoomed_on_terminator(args.scene, W, H, vfov_deg=args.zoom, spp_per_launch=S)
    scene.max_spp = spp
    scene.path_seg_min, scene.path_seg_max = seg
    rt = MoonRT(W, H, device=dev, rank=rank, world=world)
    rt.bind_dem(dem_buf, dem_h, dem_w)
    if col_buf is not None:
        rt.bind_color(col_buf, col_shape[0], col_shape[1])
    rt.apply_scene(scene)
    gather = mdist.FrameGather(rt, torch.device("cuda", dev))
    base_flags = _lib.F_INWAVE_PATHS if args.inwave_paths else 0

    def step():
        rt.reset()
        return gather.render_and_gather(n_blocks)     # world 1: plain render; world > 1: render in parts, gather overlapped

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    def timed(n):
        acc = {"kernel_ms": 0.0, "primary_ms": 0.0, "paths_ms": 0.0, "pack_ms": 0.0, "gather_wait_ms": 0.0, "unpack_ms": 0.0}
        barrier()
        t = time.perf_counter()
        for _ in range(n):
            st_ = step()
            for k in ("kernel_ms", "primary_ms", "paths_ms"):
                acc[k] += st_[k]
            for k, v in (getattr(gather, "last_timing", None) or {}).items():
                if k in acc:
                    acc[k] += v
        barrier()
        el = time.perf_counter() - t
        return el, {k: v / max(1, n) for k, v in acc.items()}

    # one counted frame (deterministic sample counts for the roofline), untimed
    rt.set_params(flags=_lib.F_COUNT_STATS | base_flags)
    counted = step()
    rt.set_params(flags=base_flags)
    for _ in range(args.warmup):
        step()
    elapsed, km = timed(args.steps)

    # whole-job numbers: max time over ranks, counts summed over ranks
    red_dev = "cpu" if backend == "gloo" else "cuda"
    mine = torch.tensor([elapsed, km["kernel_ms"], km["primary_ms"], km["paths_ms"], km["pack_ms"], km["gather_wait_ms"], km["unpack_ms"]],
                        dtype=torch.float64, device=red_dev)
    tt = mine.clone()
    cnt = torch.tensor([counted[k] for k in COUNT_KEYS], dtype=torch.int64, device=red_dev)
    per_rank = None
    if world > 1:
        allr = [torch.zeros_like(mine) for _ in range(world)]
        torch.distributed.all_gather(allr, mine)
        per_rank = [[round(float(v), 4) for v in t_] for t_ in allr]
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(cnt, op=torch.distributed.ReduceOp.SUM)
    elapsed, kernel_ms, primary_ms, paths_ms = (float(v) for v in tt[:4])
    frame = dict(zip(COUNT_KEYS, (int(v) for v in cnt)))

    # Beside the headline: the direct-light-only frame (path_seg_range (1, 1)).  Its counters are also what
    # render_kernel<MODE 2> performs inside the headline frame (same camera rays, vertices and shadow rays).
    also, direct_counts = None, None
    if world == 1 and seg[1] > 1 and not args.no_secondary:
        scene.path_seg_min, scene.path_seg_max = 1, 1
        rt.apply_scene(scene)
        rt.set_params(flags=_lib.F_COUNT_STATS)
        direct_counts = step()
        rt.set_params(flags=0)
        step()
        e2, k2 = timed(3)
        also = {"path_seg_range": [1, 1], "value": round(W * H * spp / (e2 / 3) / 1e6, 2), "unit": "Mrays/s",
                "ms_per_step": round(e2 / 3 * 1e3, 3), "kernel_ms": round(k2["kernel_ms"], 3),
                "roofline_frac_strict": round(algorithmic_bytes(direct_counts, W, H) / (k2["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "note": "direct light only (round 1's headline): one launch of render_kernel<64, false, wide, 0, false>; the reference never runs this"}
        scene.path_seg_min, scene.path_seg_max = seg
        rt.apply_scene(scene)
        rt.set_params(flags=base_flags)

    # ... and the close-up the reference's user actually looks at most of the time: the same frame size and DEM, camera
    # turned onto the terminator, field of view 0.7 deg (every pixel of the frame is on the Moon)
    zoom = None
    if world == 1 and seg[1] > 1 and not args.no_secondary:
        from moonrtx_amd.scene import zoomed_on_terminator
        zs = zoomed_on_terminator(args.scene, W, H, vfov_deg=0.7, spp_per_launch=S)
        zs.max_spp = spp
        zs.path_seg_min, zs.path_seg_max = seg
        rt.apply_scene(zs)
        rt.set_params(flags=base_flags)
        step()
        e3, k3 = timed(3)
        zoom = {"scene": f"{args.scene} zoomed on the terminator, vfov 0.7 deg, path_seg_range {list(seg)}",
                "value": round(W * H * spp / (e3 / 3) / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(e3 / 3 * 1e3, 3),
                "kernel_ms": round(k3["kernel_ms"], 3), "primary_ms": round(k3["primary_ms"], 3), "paths_ms": round(k3["paths_ms"], 3)}
        rt.apply_scene(scene)
        rt.set_params(flags=base_flags)

    # ... and the headline frame with the environment the reference binds by default (moon_renderer.py:604-607: a 16384x8192
    # star map as "TextureEnvironment"): the sky is no longer black (every tile is dispatched; the sky-only ones as
    # render_kernel<MODE 3>), every path that leaves the Moon looks its texel up
    starmap = None
    if world == 1 and seg[1] > 1 and not args.no_secondary and args.workload == "cfg3":
        rt.upload_background(synth_starmap(8192, 16384))
        step()
        e4, k4 = timed(3)
        starmap = {"scene": f"{args.scene} + synthetic 16384x8192 RGBA8 star map bound as the environment, path_seg_range {list(seg)}",
                   "value": round(W * H * spp / (e4 / 3) / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(e4 / 3 * 1e3, 3),
                   "kernel_ms": round(k4["kernel_ms"], 3), "primary_ms": round(k4["primary_ms"], 3), "paths_ms": round(k4["paths_ms"], 3)}
        rt.upload_background(None)

    facade = None
    if rank == 0 and world == 1 and args.workload == "cfg3" and not args.no_secondary:
        try:
            facade = facade_rates(dem_buf, dem_h, dem_w, col_buf, col_shape, W, H)
        except Exception as e:     # a side figure must not take the headline down
            facade = {"error": repr(e)}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        rays = W * H * spp
        cfg_used = rt.config()               # the tile size the context actually runs with (16 x 16 up to two ranks, 32 x 32 beyond)
        wide = "true" if (dem_h + 4) * (dem_w + 4) * 8 > 0xFFFFFFFF else "false"
        queue = seg[1] > 1 and not args.inwave_paths and paths_ms > 0.0   # small launches keep their paths in the wave (MOONRT_PATH_QUEUE_MIN)
        # ---- roofline.  Bytes are ALGORITHMIC; time is the HIP-event duration of the kernels.
        px_adj = -32 * W * H + 32 * W * H // world
        frame_bytes = algorithmic_bytes(counted, W, H) + px_adj
        frame_bytes_nom = algorithmic_bytes(counted, W, H, nominal=True) + px_adj
        kernels = {}
        if queue and world == 1:
            # the counted frame says what EACH kernel evaluated (MrtxStats::camera_* = render_kernel's own share, trial segment
            # included; the rest is path_kernel's) -- round 3 borrowed the direct frame's counts for the render kernel
            cam = {"dem_fetches": counted["camera_dem_fetches"], "mip_fetches": counted["camera_mip_fetches"],
                   "colour_fetches": counted["camera_colour_fetches"], "background_fetches": counted["camera_background_fetches"],
                   "height_samples": counted["camera_height_samples"]}
            a_bytes = algorithmic_bytes(cam, W, H)
            kernels["render_kernel<%d, false, %s, 2, false>" % (S, wide)] = {
                "ms": round(primary_ms, 3), "algorithmic_bytes": int(a_bytes),
                "achieved": round(a_bytes / (primary_ms * 1e-3) / 1e9, 1), "frac": round(a_bytes / (primary_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "counts": {k: int(v) for k, v in cam.items()},
                "does": "camera ray, march, first vertex, light sample + shadow march, roulette + continuation ray + its first "
                        "segment, hand-over records; algorithmic bytes from this kernel's OWN counters in the counted frame"}
            p_bytes = frame_bytes - a_bytes
            kernels["path_kernel<false, %s> + resolve_paths_kernel<%d>" % (wide, S)] = {
                "ms": round(paths_ms, 3), "algorithmic_bytes": int(p_bytes),
                "achieved": round(p_bytes / (paths_ms * 1e-3) / 1e9, 1), "frac": round(p_bytes / (paths_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "does": "continuation-ray marches, later vertices and their shadow rays (persistent waves fed from the record queue), per-pixel sums",
                "limited_by_note": "its traffic beyond the L2s is ~0.74 of the HBM peak, but it is not bound by those bytes: a bit-exact build that "
                                   "proves 74 % of its march steps unnecessary from a finer max-mip (-DMRTX_PATH_MIP2=1) moves 14 GB instead of 26 GB per "
                                   "frame and takes 4.53 ms instead of 4.35 (round 4, DESIGN.md section 4.6): the chain of dependent memory rounds a "
                                   "persistent wave goes through is the bound, with the memory system close to saturation at the same time"}
        if kernels:
            dom = max(kernels, key=lambda k: kernels[k]["ms"])
            dom_bytes, dom_ms = kernels[dom]["algorithmic_bytes"], kernels[dom]["ms"]
        else:
            if queue:
                dom = "render_kernel<%d, false, %s, 2, false> + path_kernel<false, %s> + resolve_paths_kernel<%d> (the frame's kernels together)" % (S, wide, wide, S)
            else:
                dom = "render_kernel<%d, false, %s, %d, false>" % (S, wide, 1 if seg[1] > 1 else 0)
            dom_bytes, dom_ms = frame_bytes, kernel_ms
        ach = dom_bytes / (dom_ms * 1e-3) / 1e9
        # PMC evidence comes from SEPARATE rocprofv3 passes of this same command (tools/profile_bench.sh): accepted only
        # if it was taken with these very kernel sources
        prof, prof_note = None, "no profile file"
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if world == 1 and args.workload == "cfg3" and os.path.isfile(tpath):
            cand = json.load(open(tpath))
            if cand.get("source_hash") == source_hash() and cand.get("path_seg") == list(seg):
                prof, prof_note = cand, f"profiles/{cand['tag']}_summary.md (separate --pmc passes of this command, kernel sources {cand['source_hash']})"
            else:
                prof_note = (f"profiles/traffic_latest.json is stale (taken with kernel sources {cand.get('source_hash')}, "
                             f"path_seg {cand.get('path_seg')}; these are {source_hash()}, {list(seg)})")
        def prof_entry(kernel):
            """Counters of exactly this instantiation (profiles written before round 3 carry no full name: base name only)."""
            if not prof:
                return None
            base = kernel.split("<")[0]
            for name, v in prof["kernels"].items():
                full = v.get("name")
                if name == base and (full is None or kernel.replace(" ", "") in full.replace("void mrtx::", "").replace(" ", "")):
                    return v
            return None

        def measured(v):
            if v is None:
                return None
            util = v["hbm_bytes"] / (v["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS if v.get("hbm_bytes") and v.get("avg_ms") else None
            lim = None
            if util is not None:
                # "memory-system": traffic beyond the L2s (Infinity-Cache hits included) >= 0.6 of the HBM peak.  Round 4 showed for
                # path_kernel that this does NOT mean "bound by bytes" (halving them left its time unchanged): dependent memory rounds
                # against a memory system close to saturation -- see the kernel's limited_by_note
                lim = "memory-system" if util >= 0.6 else ("valu+latency" if (v.get("valu_issue_frac") or 0) >= 0.3 else "latency")
            return {"hbm_bytes": int(v["hbm_bytes"]) if v.get("hbm_bytes") else None, "hbm_utilisation": None if util is None else round(util, 4),
                    "hbm_bytes_note": "FETCH_SIZE / WRITE_SIZE = traffic leaving the L2s: it INCLUDES what the 256 MB Infinity Cache serves (MI355X_MICROARCH.md), so 'hbm' here means the memory side of the L2, an upper bound on DRAM traffic",
                    "valu_issue_frac": v.get("valu_issue_frac"), "lanes_per_valu_inst": v.get("lanes_per_valu"), "l2_hit": v.get("l2_hit"),
                    "profile_kernel_ms": v.get("avg_ms"), "limited_by": lim}

        pk = prof_entry(dom)
        for kname, kv in kernels.items():        # what the counters say, kernel by kernel (the path stage is named by its first kernel)
            kv["measured"] = measured(prof_entry(kname.split(" + ")[0]))
        roof = {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 4),
                "traffic": None if pk is None else int(pk["hbm_bytes"]),
                "kernel": dom, "kernel_ms": round(dom_ms, 3), "algorithmic_bytes": int(dom_bytes),
                "limited_by": (measured(pk) or {}).get("limited_by"),
                "limiter": "`bound` names the roofline the fraction is priced against (the HBM read roofline BASELINE.json asks for); what the PMC "
                           "counters say limits each kernel is `limited_by` (here and per kernel under `kernels`): memory-system = measured traffic beyond the L2s "
                           "(FETCH_SIZE: Infinity-Cache hits included, so an upper bound on DRAM traffic) >= 0.6 of the HBM peak -- which for path_kernel "
                           "is dependent memory rounds against a nearly saturated memory system, not bytes (limited_by_note) --, valu+latency = VALU "
                           "issue >= 0.3 with that traffic far from the peak (VALU issue + dependent-load rounds)",
                "valu_issue_frac": None if pk is None else pk.get("valu_issue_frac"),
                "hbm_utilisation": None if pk is None else round(pk["hbm_bytes"] / (pk["avg_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "lanes_per_valu_inst": None if pk is None else pk.get("lanes_per_valu"),
                "l2_hit": None if pk is None else pk.get("l2_hit"),
                "profile": prof_note,
                "profile_kernel_ms": None if pk is None else pk["avg_ms"],
                # SURVEY.md 8(d) read literally: 16 B for every DEM evaluation the march DEFINES (the oracle's count), whether or not a
                # kernel had to perform it.  > 1 means the kernels did not do the counted work: the max-mip / horizon bounds prove
                # `skip_ratio` of the defined evaluations unnecessary (results and counters identical with MRTX_F_NO_SKIP, tested)
                "frac_literal": round(frame_bytes_nom / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                "frac_literal_note": "whole frame, every evaluation the march defines at 16 B (SURVEY.md 8(d) literally) / frame kernel time / 8 TB/s; "
                                     "exceeds 1 because only dem_fetches of the height_samples are performed",
                "skip_ratio": round(1.0 - frame["dem_fetches"] / max(1, frame["height_samples"]), 4),
                "frame": {"kernel_ms": round(kernel_ms, 3), "algorithmic_bytes": int(frame_bytes),
                          "achieved": round(frame_bytes / (kernel_ms * 1e-3) / 1e9, 1),
                          "frac": round(frame_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                          "achieved_nominal": round(frame_bytes_nom / (kernel_ms * 1e-3) / 1e9, 1),
                          "frac_nominal": round(frame_bytes_nom / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)},
                "kernels": kernels or None,
                "note": "achieved = ALGORITHMIC bytes / HIP-event duration: 16 B per DEM evaluation PERFORMED + 4 B per max-mip / "
                        "medium-mip texel read to prove the others unnecessary + 16 B per colour fetch + 4 B per background texel + "
                        "32 B per pixel (SURVEY.md 8(d)); it is a cache-served throughput, NOT an HBM utilisation (that is "
                        "hbm_utilisation, from the PMC traffic); *_nominal counts every evaluation the march defines (the skips "
                        "prove skip_ratio of them unnecessary, results unchanged) and can exceed 1.  A kernel change that REMOVES "
                        "evaluations lowers `frac` while the frame gets faster (round 4: 0.60 -> 0.5 for the frame, 20.4 -> 17.6 ms); "
                        "frac_literal prices the defined evaluations and moves with the time"}
        out = {
            "metric": "Mrays/s (primary camera samples) at 3840x2160, 64 spp, downscale-2 DEM" if args.workload == "cfg3"
                      else f"Mrays/s (primary camera samples), {args.workload}",
            "value": round(rays / (ms_per_step * 1e-3) / 1e6, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {W}x{H}, {spp} spp, DEM {dem_h}x{dem_w} f32"
                                   + (f", colour {col_shape[0]}x{col_shape[1]} RGBA8" if col_shape else ", grey albedo")
                                   + f", scene {args.scene}" + (f" zoomed on the terminator (vfov {args.zoom} deg)" if args.zoom > 0 else "")
                                   + f", path_seg_range {seg}"
                                   + (" = the reference's setting (moon_renderer.py:583)" if seg == (2, 4) else ""),
                       "parallelism": f"image tiles {cfg_used['tile_w']}x{cfg_used['tile_h']} dealt round-robin (2-D lattice) over {world} GPU(s)"
                                      + ("" if world == 1 else (", packed float4 radiance" + (" + hit" if gather.with_hits else "") + " tiles of the active tiles ("
                                         + ("32" if gather.with_hits else "16") + " B per pixel) gathered to rank 0 by torch.distributed.gather over ")
                                         + ("RCCL (xGMI)" if backend == "nccl" else f"{backend} (staged through host memory: a rehearsal, not the production transport)")
                                         + f" in {int((getattr(gather, 'last_timing', None) or {}).get('parts', 1))} part(s), all but the last overlapped with rendering"),
                       "march": "step 5e-3, eps 3e-4, scene_epsilon 1e-4, 1 light sample + shadow ray per path vertex, "
                                f"path_seg_range {seg}" + (" (direct light only)" if seg[1] <= 1 else
                                                          (", paths continued by persistent waves behind a record queue" if queue else
                                                           ", paths continued inside the render wave"))},
            "distributed": {"backend": backend, "world_size": dist_world,
                            "per_rank_[wall_s, kernel_ms, primary_ms, paths_ms, pack_ms, gather_wait_ms, unpack_ms]": per_rank,
                            "gather_bytes_per_rank": getattr(gather, "last_bytes", None),
                            "note": "per step, means over the timed steps: kernel_ms = HIP-event time of this rank's kernels; pack_ms = "
                                    "mrtx_pack_part calls; gather_wait_ms = what of the exchange the rendering did not hide (wait for the "
                                    "asynchronous gathers + device synchronise); unpack_ms = mrtx_unpack_all on rank 0"},
            "kernel_ms": round(kernel_ms, 3), "primary_ms": round(primary_ms, 3), "paths_ms": round(paths_ms, 3),
            "frame_counts": frame,
            "hit_samples_per_s": round(frame["primary_hits"] / (ms_per_step * 1e-3), 1),
            "value_note": "1 ray = 1 primary camera sample (SURVEY.md 8(d)): the %.0f %% of them that the host-side sky cull proves black are "
                          "counted and never traced; hit_samples_per_s counts the samples that met the Moon" % (100.0 * (1.0 - frame["primary_hits"] / rays)),
            "bytes_per_ray": round(algorithmic_bytes(frame, W, H) / rays, 2),
            "bytes_per_ray_nominal": round(algorithmic_bytes(frame, W, H, nominal=True) / rays, 2),
            "roofline": roof,
            "inputs_s": round(t_inputs, 2),
            "hip_runtime": _lib.hip_runtimes_loaded(),     # the ONE libamdhip64 this process runs on (moonrtx_amd/_lib.py)
        }
        if also is not None:
            out["also"] = also
        if zoom is not None:
            out["also_zoomed"] = zoom
        if starmap is not None:
            out["also_starmap"] = starmap
        if facade is not None:
            out["facade"] = facade
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, dem_buf, (dem_h, dem_w), col_buf, col_shape, frame,
                                               args.cpu_budget_s)
            try:
                out["cpu_baseline_numpy"] = numpy_baselines(args, scene, dem_buf, dem_h, dem_w, W, H, dev, direct_counts, frame)
            except Exception as e:     # a side figure must not take the headline down
                out["cpu_baseline_numpy"] = {"error": repr(e)}
        print(json.dumps(out), flush=True)

    rt.close()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
