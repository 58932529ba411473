"""Worker of tests/test_gpu_multiproc.py: one rank of a torch.distributed.run group, the REAL HIP renderer behind
moonrtx_amd.dist.FrameGather (gloo transport, every rank on device 0: MOONRT_DIST_BACKEND=gloo MOONRT_ONE_DEVICE=1 -- a one-GPU
box cannot give RCCL a device per rank).  Rank 0 checks the gathered frame against its own single-rank render, bit for bit."""
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE)); sys.path.insert(0, HERE)
import numpy as np
import synth_np
from moonrtx_amd import dist as mdist
from moonrtx_amd.renderer import MoonRT
from moonrtx_amd.scene import named_scene


def main():
    rank, world, local = mdist.init_process_group()
    import torch
    W, H = 320, 192
    dem = synth_np.dem(360, 720, seed=7, craters=40)
    col = synth_np.colour_map(180, 360)
    bg = np.random.default_rng(3).integers(0, 255, (64, 128, 4), dtype=np.uint8)
    with_hits = os.environ.get("MP_WITH_HITS", "0") == "1"
    failures = []
    for idx, (name, seg, env) in enumerate((("S1", (2, 4), None), ("S3", (1, 1), bg))):
        s = named_scene(name, W, H, spp_per_launch=64)
        s.path_seg_min, s.path_seg_max = seg
        rt = MoonRT(W, H, device=local, rank=rank, world=world)
        rt.upload_dem(dem); rt.upload_color(col); rt.upload_background(env); rt.apply_scene(s)
        g = mdist.FrameGather(rt, torch.device("cuda", local), with_hits=with_hits)
        rt.reset()
        g.render_and_gather(1, parts=1 if idx == 0 else 2)     # the second scene in two parts: the overlapped gather (device transports only)
        picks = [(W // 2, H // 2), (W // 2 + 37, H // 2 - 20), (3, 5), (W - 1, H - 1)]
        got_hits = [g.hit_at(x, y) for x, y in picks]          # a collective: every rank calls it
        if rank == 0:
            lin = rt.read_linear()
            one = MoonRT(W, H, device=local)
            one.upload_dem(dem); one.upload_color(col); one.upload_background(env); one.apply_scene(s)
            one.reset(); one.render(1)
            ref = one.read_linear()
            same = np.ascontiguousarray(lin).view(np.uint32) == np.ascontiguousarray(ref).view(np.uint32)
            if not same.all():
                failures.append(f"{name} {seg}: {int((~same).sum())} of {same.size} radiance words differ from the single-rank frame")
            if ref[..., :3].max() <= 0.0:
                failures.append(f"{name} {seg}: empty reference frame")
            for (x, y), h in zip(picks, got_hits):
                r = one.read_hit(x, y)
                if tuple(np.float32(v) for v in h) != tuple(np.float32(v) for v in r):
                    failures.append(f"{name} {seg}: hit_at({x}, {y}) = {h} but the single-rank frame holds {r}")
            if with_hits:
                hs = np.ascontiguousarray(rt.read_hits()).view(np.uint32) == np.ascontiguousarray(one.read_hits()).view(np.uint32)
                if not hs.all():
                    failures.append(f"{name} {seg}: {int((~hs).sum())} hit words differ")
            one.close()
        rt.close()
    import torch.distributed as td
    td.barrier()
    if rank == 0:
        print("MP_GATHER " + ("OK world %d" % world if not failures else "FAIL " + "; ".join(failures)), flush=True)
    td.destroy_process_group()
    sys.exit(1 if failures else 0)


if __name__ == "__main__":
    main()
