#!/usr/bin/env python3
"""Per-rank kernel time of the sharded cfg3 frame (PATH_SEG=2,4 by default; 1,1 = direct light), measured one rank after another on ONE GPU: the load balance of the
tile deal (tile t -> rank t % world) and the bytes each rank would send.  python tools/rank_balance.py [world ...]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from bench import WORKLOADS
from moonrtx_amd.renderer import MoonRT, synth_ldem, synth_color, dem_from_ldem
from moonrtx_amd.scene import named_scene

W, H, spp, dem_h, dem_w, col_shape = WORKLOADS["cfg3"]
src = synth_ldem(dem_h, dem_w, device=0)
dem_buf, _ = dem_from_ldem(src, dem_h, dem_w, 1, device=0)
src.free()
col = synth_color(col_shape[0], col_shape[1], device=0)
scene = named_scene(os.environ.get("SCENE", "S1"), W, H, spp_per_launch=64)
scene.path_seg_min, scene.path_seg_max = (int(t) for t in os.environ.get("PATH_SEG", "2,4").split(","))
for world in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    ms, nbytes, split, wall = [], 0, [], []
    for r in range(world):
        tile = tuple(int(t) for t in os.environ.get("TILE", "32,32").split(","))
        rt = MoonRT(W, H, device=0, rank=r, world=world, tile=tile)
        rt.bind_dem(dem_buf, dem_h, dem_w)
        rt.bind_color(col, col_shape[0], col_shape[1])
        rt.apply_scene(scene)
        rt.set_params(flags=int(os.environ.get("MRTX_FLAGS", "0")))
        rt.set_gather_hits(os.environ.get("HITS", "0") == "1")       # the exchange moves the linear framebuffer only by default (round 4)
        rt.reset(); rt.render(1)
        t = []
        P = rt.shard_parts(int(os.environ.get("PARTS", "1"))) if world > 1 else 1     # PARTS=2: as FrameGather.render_and_gather drives it
        for _ in range(3):
            rt.reset()
            if P > 1:
                sts = [rt.render_part(1, k, P) for k in range(P)]
                st = {key: sum(s_[key] for s_ in sts) for key in ("kernel_ms", "primary_ms", "paths_ms")}
            else:
                st = rt.render(1)
            t.append((st["kernel_ms"], st["primary_ms"], st["paths_ms"]))
        ms.append(min(t)[0]); split.append(min(t)[1:])
        if os.environ.get("WALL") == "1" and world > 1:
            # what a step of bench.py costs this rank on the HOST clock before the collective: reset + render + pack into a send buffer
            # (launch overheads, event bookkeeping and the stream synchronisations included); 20 steps, after 2 to warm up
            import time
            from moonrtx_amd.renderer import DeviceBuffer
            send = DeviceBuffer(rt.shard_bytes(), device=0)
            for i in range(22):
                if i == 2:
                    t0 = time.perf_counter(); ksum = 0.0
                rt.reset(); st = rt.render(1); rt.pack_shard(send.ptr)
                if i >= 2:
                    ksum += st["kernel_ms"]
            wall.append(((time.perf_counter() - t0) / 20 * 1e3, ksum / 20))
            send.free()
        nbytes = rt.shard_bytes_active() if world > 1 else 0
        full = rt.shard_bytes()
        rt.close()
    print(f"world {world} ({P} part(s)): kernel ms per rank min {min(ms):.3f} mean {sum(ms) / len(ms):.3f} max {max(ms):.3f} "
          f"(imbalance {max(ms) / (sum(ms) / len(ms)) - 1:.1%}); sum {sum(ms):.2f}; shard {nbytes / 1e6:.1f} MB active of {full / 1e6:.1f} MB; "
          f"slowest rank: render {max(split)[0]:.3f} + paths {max(split)[1]:.3f}")
    if wall:
        print(f"   host clock per step (reset + render + pack, mean of 20): slowest rank {max(wall)[0]:.3f} ms with kernels {max(wall)[1]:.3f} ms; "
              f"mean over ranks {sum(w for w, _ in wall) / len(wall):.3f} ms with kernels {sum(k for _, k in wall) / len(wall):.3f} ms")
