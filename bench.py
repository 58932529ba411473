#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MoonRTX hot path on MI355X.

Metric (BASELINE.json): Mrays/s + ms/frame at 3840x2160, 64 spp, --downscale-2-sized DEM; 1 ray = 1 primary
camera sample.  A "step" is one full frame: restart the accumulation cycle
(`refresh_scene`), render all samples of every pixel, and -- on N > 1 GPUs -- gather the tiles to rank 0.
Inputs (synthetic LOLA-like DEM + colour map) are generated on the device and are
resident in HBM before the timed region.

  python bench.py --gpus 1 --steps 5 --warmup 1
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.
"""
import argparse
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


def algorithmic_bytes(st, width, height, nominal=False):
    """16 B per DEM bilinear evaluation, 16 B per colour fetch, 4 B per background texel,
    32 B per pixel (one float4 radiance + one float4 hit write).

    Strict (default): the DEM evaluations the kernel actually PERFORMS (`dem_fetches`) plus the max-mip texels
    it reads to prove the others unnecessary (4 B each).  nominal=True: the evaluations the march DEFINES
    (`height_samples`, the oracle's count) -- what a kernel without the result-preserving skip would read."""
    dem = st["height_samples"] if nominal else st["dem_fetches"]
    mip = 0 if nominal else st["mip_fetches"]
    return 16 * dem + 4 * mip + 16 * st["colour_fetches"] + 4 * st["background_fetches"] + 32 * width * height


def cpu_baseline(scene, dem_buf, dem_shape, col_buf, col_shape, frame_stats, budget_s=20.0):
    """Time the CPU oracle (kind "port") on a bounded sample of the same workload, on this host's cores.

    A short probe crop gives the host's DEM-samples/s; the timed sample is then the largest centred crop of
    the SAME frame (same scene, DEM, spp) predicted to fit `budget_s` -- the whole frame when it fits.  The
    rate is taken in DEM samples per second and converted to whole-frame Mrays/s with the frame's own
    deterministic sample counts, so cheap sky pixels are not mis-priced."""
    import numpy as np
    from oracle import orc
    dem = dem_buf.download(np.float32, dem_shape)
    col = col_buf.download(np.uint8, col_shape + (4,)) if col_buf is not None else None
    threads = orc.set_threads(min(os.cpu_count() or 1, 16))   # the box's CPU share for one GPU
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
        "cores": threads, "kind": "port",
        "sample": (f"oracle/mrtx_oracle.c, OpenMP x{threads}: " + ("the WHOLE frame" if whole else
                   f"centred {region[2]-region[0]}x{region[3]-region[1]} crop of the frame")
                   + f" at {scene.spp_per_launch} spp, same scene/DEM/colour map: {st['primary_rays']} rays, "
                   f"{st['height_samples']} DEM samples in {dt:.2f} s = {hs_per_s/1e6:.1f} M DEM samples/s"
                   + ("" if whole else f"; scaled to the frame's {frame_stats['height_samples']} DEM samples")),
        "seconds": round(dt, 2),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cfg3", choices=sorted(WORKLOADS))
    ap.add_argument("--scene", default="S1", choices=["S1", "S2", "S3"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the path_seg_range (2,4) side measurement")
    ap.add_argument("--cpu-budget-s", type=float, default=20.0)
    ap.add_argument("--dem-scale", type=int, default=1, help="debug: shrink the DEM by this factor")
    ap.add_argument("--path-seg", type=int, nargs=2, default=(1, 1), metavar=("MIN", "MAX"),
                    help="path length in segments; (1,1) = direct light only (headline), the reference sets (2,4)")
    ap.add_argument("--inwave-paths", action="store_true", help="A/B: keep D6 paths inside the render wave (MRTX_F_INWAVE_PATHS)")
    args = ap.parse_args()

    import torch
    from moonrtx_amd import build, dist as mdist
    from moonrtx_amd.renderer import MoonRT, synth_ldem, synth_color, dem_from_ldem
    from moonrtx_amd.scene import named_scene
    from moonrtx_amd import _lib

    build.build_native()
    rank, world, local = mdist.init_process_group()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    dev = local
    torch.cuda.set_device(dev)
    backend = torch.distributed.get_backend() if world > 1 else None

    W, H, spp, dem_h, dem_w, col_shape = WORKLOADS[args.workload]
    dem_h //= args.dem_scale
    dem_w //= args.dem_scale
    S = min(spp, 64)
    n_blocks = spp // S

    # ---- inputs, resident in HBM before anything is timed
    t0 = time.perf_counter()
    src = synth_ldem(dem_h, dem_w, device=dev)
    dem_buf, radius_scale = dem_from_ldem(src, dem_h, dem_w, 1, device=dev)
    src.free()
    col_buf = synth_color(col_shape[0], col_shape[1], device=dev) if col_shape else None
    t_inputs = time.perf_counter() - t0

    scene = named_scene(args.scene, W, H, spp_per_launch=S)
    scene.max_spp = spp
    scene.path_seg_min, scene.path_seg_max = args.path_seg
    rt = MoonRT(W, H, device=dev, rank=rank, world=world)
    rt.bind_dem(dem_buf, dem_h, dem_w)
    if col_buf is not None:
        rt.bind_color(col_buf, col_shape[0], col_shape[1])
    rt.apply_scene(scene)
    gather = mdist.FrameGather(rt, torch.device("cuda", dev))

    def step():
        rt.reset()
        return gather.render_and_gather(n_blocks)     # world 1: plain render; world > 1: render in parts, gather overlapped

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # one counted frame (deterministic sample counts for the roofline), untimed
    base_flags = _lib.F_INWAVE_PATHS if args.inwave_paths else 0
    rt.set_params(flags=_lib.F_COUNT_STATS | base_flags)
    counted = step()
    rt.set_params(flags=base_flags)
    for _ in range(args.warmup):
        step()

    barrier()
    t = time.perf_counter()
    kernel_ms = primary_ms = paths_ms = 0.0
    for _ in range(args.steps):
        st_ = step()
        kernel_ms += st_["kernel_ms"]; primary_ms += st_["primary_ms"]; paths_ms += st_["paths_ms"]
    barrier()
    elapsed = time.perf_counter() - t
    kernel_ms /= max(1, args.steps); primary_ms /= max(1, args.steps); paths_ms /= max(1, args.steps)

    # whole-job numbers: max time over ranks, counts summed over ranks
    red_dev = "cpu" if backend == "gloo" else "cuda"
    tt = torch.tensor([elapsed, kernel_ms], dtype=torch.float64, device=red_dev)
    keys = ("primary_rays", "primary_hits", "shadow_rays", "height_samples", "colour_fetches", "background_fetches",
            "dem_fetches", "mip_fetches")
    cnt = torch.tensor([counted[k] for k in keys], dtype=torch.int64, device=red_dev)
    if world > 1:
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(cnt, op=torch.distributed.ReduceOp.SUM)
    elapsed, kernel_ms = float(tt[0]), float(tt[1])
    frame = dict(zip(keys, (int(v) for v in cnt)))

    # Beside the headline (direct light): the same frame with the reference's own default
    # path_seg_range (2, 4) (moon_renderer.py:583), a few untimed-region steps on one GPU, reported as a secondary figure.
    also = None
    if world == 1 and tuple(args.path_seg) == (1, 1) and not args.no_secondary:
        scene.path_seg_min, scene.path_seg_max = 2, 4
        rt.apply_scene(scene)
        rt.set_params(flags=base_flags)
        step()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        k2 = sum(step()["kernel_ms"] for _ in range(3)) / 3.0
        torch.cuda.synchronize()
        e2 = (time.perf_counter() - t2) / 3.0
        also = {"path_seg_range": [2, 4], "value": round(W * H * spp / e2 / 1e6, 2), "unit": "Mrays/s",
                "ms_per_step": round(e2 * 1e3, 3), "kernel_ms": round(k2, 3),
                "note": "the reference's default path length (D6: up to 3 further segments with next-event estimation)"}
        scene.path_seg_min, scene.path_seg_max = args.path_seg
        rt.apply_scene(scene)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        rays = W * H * spp
        # roofline of the dominant (render) kernel: this rank's algorithmic bytes / its launch duration
        px_adj = -32 * W * H + 32 * W * H // world
        ach = (algorithmic_bytes(counted, W, H) + px_adj) / (kernel_ms * 1e-3) / 1e9
        ach_nom = (algorithmic_bytes(counted, W, H, nominal=True) + px_adj) / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        if world == 1 and args.workload == "cfg3" and os.path.isfile(tpath):
            traffic = json.load(open(tpath))   # from the rocprofv3 --pmc passes of this same command (tools/profile_bench.sh)
        out = {
            "metric": "Mrays/s (primary camera samples) at 3840x2160, 64 spp, downscale-2 DEM" if args.workload == "cfg3"
                      else f"Mrays/s (primary camera samples), {args.workload}",
            "value": round(rays / (ms_per_step * 1e-3) / 1e6, 2), "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.workload}: {W}x{H}, {spp} spp, DEM {dem_h}x{dem_w} f32"
                                   + (f", colour {col_shape[0]}x{col_shape[1]} RGBA8" if col_shape else ", grey albedo")
                                   + f", scene {args.scene}",
                       "parallelism": f"image tiles 32x32 dealt round-robin (2-D lattice) over {world} GPU(s), active tiles gathered to rank 0"
                                      + (", RCCL gather of float4 radiance+hits to rank 0" if world > 1 else ""),
                       "march": "step 5e-3, eps 3e-4, scene_epsilon 1e-4, 1 light sample + shadow ray per path vertex, "
                                f"path_seg_range {tuple(args.path_seg)}" + (" (direct light only)" if args.path_seg[1] <= 1 else "")},
            "kernel_ms": round(kernel_ms, 3), "primary_ms": round(primary_ms, 3), "paths_ms": round(paths_ms, 3),
            "frame_counts": frame,
            "bytes_per_ray": round(algorithmic_bytes(frame, W, H) / rays, 2),
            "bytes_per_ray_nominal": round(algorithmic_bytes(frame, W, H, nominal=True) / rays, 2),
            "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(ach / HBM_PEAK_GBS, 4),
                         "traffic": None if traffic is None else round(traffic["hbm_bytes_per_launch"]),
                         "traffic_source": None if traffic is None else f"profiles/{traffic['tag']}_summary.md: FETCH_SIZE x {traffic['fetch_factor']} (calibrated on a known 8-byte-per-lane stream) + WRITE_SIZE",
                         "algorithmic_bytes": int(algorithmic_bytes(counted, W, H) + px_adj),
                         "kernel": "mrtx::render_kernel<%d, false, %s, %s, false>" % (
                             S, "true" if (dem_h + 4) * (dem_w + 4) * 8 > 0xFFFFFFFF else "false",
                             "true" if args.path_seg[1] > 1 else "false"),
                         "achieved_nominal": round(ach_nom, 1), "frac_nominal": round(ach_nom / HBM_PEAK_GBS, 4),
                         "note": "achieved = algorithmic bytes / HIP-event launch duration, bytes = 16 B per DEM "
                                 "evaluation PERFORMED + 4 B per max-mip texel + 16 B per colour fetch + 4 B per "
                                 "background texel + 32 B per pixel; *_nominal counts every evaluation the march "
                                 "defines (the max-mip skip proves most of them unnecessary, results unchanged); "
                                 f"strict frac vs the 6290 GB/s measured-copy peak: {round(ach / 6290.0, 4)}"},
            "inputs_s": round(t_inputs, 2),
        }
        if also is not None:
            out["also"] = also
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(scene, dem_buf, (dem_h, dem_w), col_buf, col_shape, frame,
                                               args.cpu_budget_s)
        print(json.dumps(out), flush=True)

    rt.close()
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
