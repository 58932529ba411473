"""Image-tile sharding across the GPUs of one node + the one exchange step of the path.

New in this build (the reference is single-GPU): every pixel sample is independent
given the read-only scene, so the frame is cut into tile_w x tile_h tiles dealt round-robin to ranks
(tile t -> rank t % world, t = raster order with a per-row cyclic shift so ranks form a 2-D lattice: sky, limb,
terminator and night side are spread evenly -- include/moonrt.h), each rank renders
its tiles for all samples, and ONE gather brings the packed float4 radiance + hit tiles to rank 0 over
xGMI (RCCL `gather` = grouped send/recv: every peer uses its own direct link to the root).  Only tiles the
sky cull keeps travel (mrtx_shard_bytes_active: ~38 % of a whole-disc 16:9 frame), each rank deriving the layout
from its own copy of the scene.  No
reduction is needed -- tiles are disjoint -- and the RNG is keyed by (pixel, sample), so 1-, 2-, 4- and
8-GPU frames are bit-identical.

One process per GPU, `torch.distributed` (backend "nccl" == RCCL on ROCm; "gloo" for the CPU tests).
torch is plumbing here: process group, device buffers for the collective, streams.
"""
import os


def tiles_of(rank, world, width, height, tile=(32, 32)):
    """Raster indices of the tiles a rank owns, and the padded slot count every rank packs."""
    tx = (width + tile[0] - 1) // tile[0]
    ty = (height + tile[1] - 1) // tile[1]
    n = tx * ty
    return list(range(rank, n, world)), (n + world - 1) // world


def tile_shift(world):
    """Columns of cyclic shift per tile row in the tile numbering (mrtx_tile_shift, csrc/mrtx_device.h)."""
    import math
    s = 3
    while math.gcd(s, world) != 1:
        s += 2
    return s


def tile_xy(t, tiles_x, shift):
    """Tile number -> (tx, ty) tile coordinates (mrtx_tile_xy)."""
    ty, c = divmod(t, tiles_x)
    return (c - shift * ty) % tiles_x, ty


def tile_owner(x, y, width, tile, world):
    """Rank that renders pixel (x, y): its tile's number (mrtx_tile_id: raster order with the per-row cyclic shift) mod world."""
    tiles_x = (width + tile[0] - 1) // tile[0]
    tx, ty = x // tile[0], y // tile[1]
    return (ty * tiles_x + (tx + tile_shift(world) * ty) % tiles_x) % world


def env_rank_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init_process_group(backend=None):
    """Join the group torchrun described in the environment (MASTER_ADDR must be 127.0.0.1 on one node).

    MOONRT_DIST_BACKEND=gloo + MOONRT_ONE_DEVICE=1 rehearse the multi-rank path on a single-GPU box (every rank
    renders on device 0, the gather is staged through host memory); production is one rank per GPU over RCCL."""
    from . import _lib
    _lib.load()                      # settles which HIP runtime this process uses (one copy, see _lib._preload_torch_hip_runtime)
    import torch
    import torch.distributed as dist
    _lib.assert_single_hip_runtime()
    rank, world, local = env_rank_world()
    if os.environ.get("MOONRT_ONE_DEVICE") == "1":
        local = 0
    if world == 1:
        return rank, world, local
    if backend is None:
        backend = os.environ.get("MOONRT_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if backend == "nccl":
        torch.cuda.set_device(local)
    if not dist.is_initialized():
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def gather_parts(world, shard_bytes, requested=0):
    """In how many parts a rank renders + gathers its tile list.  `requested` > 0 (MOONRT_GATHER_PARTS) wins; else two parts from
    32 MB per rank up, one below: a part costs a render + path-stage + resolve launch of its own (+0.3 ms per rank at cfg3, measured
    rank after rank on one GPU: profiles/r03_rank_balance.md) and hides the transfer of the OTHER part only (~50 GB/s per xGMI link
    assumed, unmeasured here) -- at cfg3 that pays for two ranks (50 MB each), not for four (25 MB) or eight (12.6 MB)."""
    if world <= 1:
        return 1
    if requested > 0:
        return int(requested)
    return 2 if shard_bytes >= (32 << 20) else 1


class FrameGather:
    """Reusable buffers + the gather of packed shards to rank 0.

    `renderer` needs shard_bytes(), pack_shard(ptr), unpack_shard(src_rank, ptr), rank, world -- the
    MoonRT surface (mrtx_shard_bytes / mrtx_pack_shard / mrtx_unpack_shard in include/moonrt.h)."""

    def __init__(self, renderer, device, with_hits=False):
        """`with_hits` = False (default): the exchange moves the final linear framebuffer only, 16 bytes per pixel -- what the
        north star asks the gather for; the facade reads ONE hit texel per mouse event (moon_renderer.py:1137-1142), which
        `hit_at()` fetches from the rank that owns the pixel.  True: the float4 hit tiles travel with the radiance (32 bytes per
        pixel) and rank 0 holds the whole hit buffer after every frame."""
        import torch
        self.torch = torch
        self.r = renderer
        self.rank, self.world = renderer.rank, renderer.world
        self.with_hits = bool(with_hits) or self.world == 1
        if hasattr(renderer, "set_gather_hits"):
            renderer.set_gather_hits(self.with_hits)
        self.nbytes = renderer.shard_bytes()
        self.send = torch.empty(self.nbytes // 4, dtype=torch.float32, device=device)
        self.recv = None
        if self.rank == 0 and self.world > 1:
            self.recv = [torch.empty_like(self.send) for _ in range(self.world)]
        # gloo cannot gather device tensors: stage through host memory (rehearsal / CPU tests only)
        self.host_staged = False
        if self.world > 1 and self.send.is_cuda:
            import torch.distributed as dist
            self.host_staged = dist.get_backend() == "gloo"
        if self.host_staged:
            self.send_h = torch.empty(self.nbytes // 4, dtype=torch.float32, pin_memory=True)
            self.recv_h = [torch.empty(self.nbytes // 4, dtype=torch.float32) for _ in range(self.world)] if self.rank == 0 else None

    def render_and_gather(self, n_blocks=1, parts=None):
        """One block of samples on every rank + the exchange, overlapped: the tile list renders in `parts` pieces
        (MOONRT_GATHER_PARTS, default: two pieces when a rank's shard is 32 MB or more, else one) and the RCCL gather of piece k
        runs on the collective's stream while
        piece k+1 renders.  Falls back to render() + gather() when the scene needs the full layout, on a single
        rank, and with host-staged (gloo) transport.  Returns the summed render statistics."""
        r = self.r
        if parts is None:
            parts = int(os.environ.get("MOONRT_GATHER_PARTS", "0"))
        if parts <= 0:
            parts = gather_parts(self.world, r.shard_bytes_active() if hasattr(r, "shard_bytes_active") else 0)
        P = r.shard_parts(parts) if (self.world > 1 and parts > 1 and hasattr(r, "shard_parts") and not self.host_staged) else 1
        if P == 1:
            st = r.render(n_blocks)
            self.gather()
            return st
        import time
        import torch.distributed as dist
        works, total = [], None
        t_pack = t_wait = t_unpack = 0.0
        for k in range(P):
            st = r.render_part(n_blocks, k, P)                      # returns when the piece is rendered
            t0 = time.perf_counter()
            off, ln = r.pack_part(self.send.data_ptr(), k, P)       # synchronous on the renderer's stream
            t_pack += time.perf_counter() - t0
            a, b = off // 4, (off + ln) // 4
            if b > a:                                               # same cut on every rank
                works.append(dist.gather(self.send[a:b], [t[a:b] for t in self.recv] if self.rank == 0 else None,
                                         dst=0, async_op=True))
            if total is None:
                total = dict(st)
            else:
                for key, v in st.items():
                    total[key] += v
        t0 = time.perf_counter()
        for w in works:
            w.wait()
        self.torch.cuda.synchronize()
        t_wait = time.perf_counter() - t0                           # what of the exchange the rendering did not hide
        self.last_bytes = r.shard_bytes_active()
        if self.rank == 0:
            t0 = time.perf_counter()
            r.unpack_all([t.data_ptr() for t in self.recv])
            t_unpack = time.perf_counter() - t0
        self.last_timing = {"parts": P, "pack_ms": t_pack * 1e3, "gather_wait_ms": t_wait * 1e3, "unpack_ms": t_unpack * 1e3}
        return total

    def hit_at(self, x, y):
        """rt._get_hit_at(x, y) on a sharded frame: (hx, hy, hz, hd) of pixel (x, y), on EVERY rank (a collective: all ranks call
        it with the same pixel).  The owner of the pixel's tile reads the one texel (mrtx_read_hit, 16 bytes over PCIe) and
        broadcasts it; with `with_hits` rank 0 already holds it but the owner's copy is the same bits."""
        r = self.r
        if self.world == 1:
            return r.read_hit(x, y)
        import torch.distributed as dist
        cfg = r.config()
        owner = tile_owner(int(x), int(y), cfg["width"], (cfg["tile_w"], cfg["tile_h"]), self.world)
        dev = "cpu" if dist.get_backend() == "gloo" else self.send.device
        t = self.torch.zeros(4, dtype=self.torch.float32, device=dev)
        if self.rank == owner:
            t.copy_(self.torch.tensor(r.read_hit(x, y), dtype=self.torch.float32))
        dist.broadcast(t, src=owner)
        return tuple(float(v) for v in t.cpu())

    def gather(self):
        """After every rank has rendered: bring all tiles to rank 0's framebuffer."""
        if self.world == 1:
            return
        import torch.distributed as dist
        torch = self.torch
        n = self.nbytes // 4
        if hasattr(self.r, "shard_bytes_active"):
            n = self.r.shard_bytes_active() // 4          # same on every rank: a function of the (identical) scene
        self.last_bytes = n * 4
        self.r.pack_shard(self.send.data_ptr())          # synchronous on the renderer's stream
        import time
        t0 = time.perf_counter()
        if n == 0:
            pass
        elif self.host_staged:
            self.send_h[:n].copy_(self.send[:n])
            dist.gather(self.send_h[:n], [t[:n] for t in self.recv_h] if self.rank == 0 else None, dst=0)
            if self.rank == 0:
                for src in range(1, self.world):
                    self.recv[src][:n].copy_(self.recv_h[src][:n])
        else:
            dist.gather(self.send[:n], [t[:n] for t in self.recv] if self.rank == 0 else None, dst=0)
        if self.send.is_cuda:
            torch.cuda.synchronize()
        t1 = time.perf_counter()
        if self.rank == 0:
            if hasattr(self.r, "unpack_all"):
                self.r.unpack_all([t.data_ptr() for t in self.recv])
            else:
                for src in range(1, self.world):
                    self.r.unpack_shard(src, self.recv[src].data_ptr())
        self.last_timing = {"parts": 1, "pack_ms": 0.0, "gather_wait_ms": (t1 - t0) * 1e3, "unpack_ms": (time.perf_counter() - t1) * 1e3}
