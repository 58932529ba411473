"""The N > 1 path on CPU: `gloo` groups of 2, 3, 4 and 8 ranks run the product's tile ownership + gather
(moonrtx_amd/dist.py) around a stand-in renderer whose pixels come from the oracle, and rank 0 must end up
with exactly the single-rank frame -- radiance through the gather (hit-less by default: the final linear framebuffer
is what travels), hit texels through FrameGather.hit_at from the rank that owns the pixel."""
import ctypes as C
import os
import socket

import numpy as np
import pytest

import synth_np
from moonrtx_amd import dist as mdist
from moonrtx_amd.scene import named_scene

TILE = (16, 16)


class OracleShardRenderer:
    """Stand-in with MoonRT's sharding surface (shard_bytes / pack_shard / unpack_shard), numpy inside."""

    def __init__(self, scene, dem, rank, world):
        from oracle import orc
        self.rank, self.world = rank, world
        self.W, self.H = scene.width, scene.height
        self.o = orc.Oracle(scene, dem)
        self.mine, self.slots = mdist.tiles_of(rank, world, self.W, self.H, TILE)
        self.tx = (self.W + TILE[0] - 1) // TILE[0]
        self.with_hits = True

    def set_gather_hits(self, on):
        self.with_hits = bool(on)

    def config(self):
        return {"device": 0, "width": self.W, "height": self.H, "rank": self.rank, "world": self.world, "tile_w": TILE[0], "tile_h": TILE[1]}

    def read_hit(self, x, y):
        return tuple(float(v) for v in self.o.hits[y, x])

    def _box(self, t):
        tx, ty = mdist.tile_xy(t, self.tx, mdist.tile_shift(self.world))
        x0, y0 = tx * TILE[0], ty * TILE[1]
        return x0, y0, min(self.W, x0 + TILE[0]), min(self.H, y0 + TILE[1])

    def render(self, n_blocks=1):
        for t in self.mine:
            self.o.blocks_done = 0
            self.o.render(n_blocks, self._box(t))
        return {"kernel_ms": 0.0}

    def shard_bytes(self):
        return self.slots * TILE[0] * TILE[1] * (32 if self.with_hits else 16)

    def _view(self, ptr):
        """The packed shard as the library lays it out: per slot one tile of sums, then (with hits) one tile of hits."""
        per = 2 if self.with_hits else 1
        n = self.slots * TILE[0] * TILE[1] * 4 * per
        buf = (C.c_float * n).from_address(ptr)
        a = np.frombuffer(buf, np.float32).reshape(self.slots, per, TILE[1], TILE[0], 4)
        return a[:, 0], (a[:, 1] if self.with_hits else None)

    def pack_shard(self, ptr, stream=None):
        acc, hit = self._view(ptr)
        acc[:] = 0
        if hit is not None:
            hit[:] = 0
        for k, t in enumerate(self.mine):
            x0, y0, x1, y1 = self._box(t)
            acc[k, :y1 - y0, :x1 - x0] = self.o.accum[y0:y1, x0:x1]
            if hit is not None:
                hit[k, :y1 - y0, :x1 - x0] = self.o.hits[y0:y1, x0:x1]

    def unpack_shard(self, src, ptr, stream=None):
        acc, hit = self._view(ptr)
        tiles, _ = mdist.tiles_of(src, self.world, self.W, self.H, TILE)
        for k, t in enumerate(tiles):
            x0, y0, x1, y1 = self._box(t)
            self.o.accum[y0:y1, x0:x1] = acc[k, :y1 - y0, :x1 - x0]
            if hit is not None:
                self.o.hits[y0:y1, x0:x1] = hit[k, :y1 - y0, :x1 - x0]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


PICKS = [(3, 2), (35, 26), (69, 51), (17, 40), (50, 9), (33, 33), (64, 20), (5, 47)]   # pixels whose hit texel every rank asks for


def _worker(rank, world, port, out_path, with_hits):
    import torch
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from oracle import orc
    orc.set_threads(2 if world <= 4 else 1)
    r, w, _ = mdist.init_process_group("gloo")
    assert (r, w) == (rank, world)
    dem = synth_np.dem(90, 180, seed=5, craters=10)
    scene = named_scene("S1", 70, 52, spp_per_launch=4)          # ragged: not a multiple of the tile
    rend = OracleShardRenderer(scene, dem, rank, world)
    g = mdist.FrameGather(rend, torch.device("cpu"), with_hits=with_hits)
    assert rend.with_hits == with_hits and g.nbytes == rend.slots * TILE[0] * TILE[1] * (32 if with_hits else 16)
    g.render_and_gather(1)            # what bench.py's step() calls (no part support here: render + one gather)
    picks = [g.hit_at(x, y) for x, y in PICKS]      # a collective: every rank asks, the owner answers
    if rank == 0:
        np.savez(out_path, accum=rend.o.accum, hits=rend.o.hits, picks=np.array(picks, np.float32))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world,with_hits", [(2, True), (2, False), (3, False), (4, False), (8, False), (8, True)])
def test_gather_reassembles_the_single_rank_frame(tmp_path, world, with_hits):
    import torch.multiprocessing as mp
    from oracle import orc
    out = str(tmp_path / "frame.npz")
    mp.spawn(_worker, args=(world, _free_port(), out, with_hits), nprocs=world, join=True)
    got = np.load(out)
    dem = synth_np.dem(90, 180, seed=5, craters=10)
    scene = named_scene("S1", 70, 52, spp_per_launch=4)
    ref = orc.Oracle(scene, dem)
    ref.render(1)
    assert np.array_equal(got["accum"].view(np.uint32), ref.accum.view(np.uint32))          # the final linear framebuffer
    if with_hits:
        assert np.array_equal(got["hits"].view(np.uint32), ref.hits.view(np.uint32))
    else:                               # the root holds the hit records of its OWN tiles; the others are served on demand
        own = np.zeros((52, 70), bool)
        tx = (70 + TILE[0] - 1) // TILE[0]
        for t in mdist.tiles_of(0, world, 70, 52, TILE)[0]:
            x, y = mdist.tile_xy(t, tx, mdist.tile_shift(world))
            own[y * TILE[1]:(y + 1) * TILE[1], x * TILE[0]:(x + 1) * TILE[0]] = True
        assert np.array_equal(got["hits"][own].view(np.uint32), ref.hits[own].view(np.uint32))
        assert not got["hits"][~own].any()
    for (x, y), h in zip(PICKS, got["picks"]):     # _get_hit_at on a sharded frame, whoever owns the pixel
        assert np.array_equal(np.asarray(h, np.float32).view(np.uint32), ref.hits[y, x].view(np.uint32)), (x, y)
    owners = {mdist.tile_owner(x, y, 70, TILE, world) for x, y in PICKS}
    assert len(owners) >= min(world, 3)
    assert ref.accum[..., :3].max() > 0 and (np.array([h[3] for h in got["picks"]]) > 0).any()


def test_tile_ownership_is_a_partition():
    for world in (1, 2, 3, 4, 8):
        seen = []
        for r in range(world):
            tiles, slots = mdist.tiles_of(r, world, 3840, 2160, (32, 32))
            assert len(tiles) <= slots
            seen += tiles
        assert sorted(seen) == list(range(120 * 68))
        # interleave => balanced: every rank within one tile of the others
        counts = [len(mdist.tiles_of(r, world, 3840, 2160)[0]) for r in range(world)]
        assert max(counts) - min(counts) <= 1


def test_gather_parts_follow_the_shard_size():
    """One part below 32 MB per rank, two from there up (cfg3: 50.1 / 25.1 / 12.6 MB at world 2 / 4 / 8); an explicit request wins."""
    from moonrtx_amd.dist import gather_parts
    assert gather_parts(1, 500 << 20) == 1
    assert gather_parts(2, 50069504) == 2 and gather_parts(4, 25100000) == 1 and gather_parts(8, 12600000) == 1
    assert gather_parts(8, 12600000, requested=2) == 2 and gather_parts(2, 50069504, requested=1) == 1
    assert gather_parts(8, 32 << 20) == 2 and gather_parts(8, (32 << 20) - 1) == 1
