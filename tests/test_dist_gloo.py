"""The N > 1 path on CPU: world_size-2 (and 3) `gloo` groups run the product's tile ownership + gather
(moonrtx_amd/dist.py) around a stand-in renderer whose pixels come from the oracle, and rank 0 must end up
with exactly the single-rank frame."""
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
        return self.slots * TILE[0] * TILE[1] * 32

    def _view(self, ptr):
        n = self.slots * TILE[0] * TILE[1] * 4
        buf = (C.c_float * (2 * n)).from_address(ptr)
        a = np.frombuffer(buf, np.float32)
        return a[:n].reshape(self.slots, TILE[1], TILE[0], 4), a[n:].reshape(self.slots, TILE[1], TILE[0], 4)

    def pack_shard(self, ptr, stream=None):
        acc, hit = self._view(ptr)
        acc[:] = 0; hit[:] = 0
        for k, t in enumerate(self.mine):
            x0, y0, x1, y1 = self._box(t)
            acc[k, :y1 - y0, :x1 - x0] = self.o.accum[y0:y1, x0:x1]
            hit[k, :y1 - y0, :x1 - x0] = self.o.hits[y0:y1, x0:x1]

    def unpack_shard(self, src, ptr, stream=None):
        acc, hit = self._view(ptr)
        tiles, _ = mdist.tiles_of(src, self.world, self.W, self.H, TILE)
        for k, t in enumerate(tiles):
            x0, y0, x1, y1 = self._box(t)
            self.o.accum[y0:y1, x0:x1] = acc[k, :y1 - y0, :x1 - x0]
            self.o.hits[y0:y1, x0:x1] = hit[k, :y1 - y0, :x1 - x0]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out_path):
    import torch
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from oracle import orc
    orc.set_threads(2)
    r, w, _ = mdist.init_process_group("gloo")
    assert (r, w) == (rank, world)
    dem = synth_np.dem(90, 180, seed=5, craters=10)
    scene = named_scene("S1", 70, 52, spp_per_launch=4)          # ragged: not a multiple of the tile
    rend = OracleShardRenderer(scene, dem, rank, world)
    g = mdist.FrameGather(rend, torch.device("cpu"))
    g.render_and_gather(1)            # what bench.py's step() calls (no part support here: render + one gather)
    if rank == 0:
        np.savez(out_path, accum=rend.o.accum, hits=rend.o.hits)
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_gather_reassembles_the_single_rank_frame(tmp_path, world):
    import torch.multiprocessing as mp
    from oracle import orc
    out = str(tmp_path / "frame.npz")
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = np.load(out)
    dem = synth_np.dem(90, 180, seed=5, craters=10)
    scene = named_scene("S1", 70, 52, spp_per_launch=4)
    ref = orc.Oracle(scene, dem)
    ref.render(1)
    assert np.array_equal(got["accum"].view(np.uint32), ref.accum.view(np.uint32))
    assert np.array_equal(got["hits"].view(np.uint32), ref.hits.view(np.uint32))
    assert ref.accum[..., :3].max() > 0


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
