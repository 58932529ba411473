"""Every BASELINE.json config at its TRUE size on the GPU (cfg 3 lives in test_gpu_fullsize.py):

  cfg 1  512x512, 1 spp, 5760x11520 DEM, grey albedo          -- whole frame against the oracle, bit for bit
  cfg 2  1920x1080, 16 spp, 11520x23040 DEM (the largest DEM on the 32-bit-offset addressing path: 2.1 GB of row
         pairs), scenes S1 + S2                                -- three oracle crops each + size-independent properties
  cfg 4  7680x4320, 256 spp, 46080x92160 DEM (17 GB + 34 GB of row pairs), ONE rank's shard of the 8-GPU job on one GPU
         -- properties + one oracle crop of a tile that rank owns

Inputs are generated on the device (synthetic LDEM through the a1 arithmetic) and downloaded
for the oracle."""
import numpy as np
import pytest

from common import STAT_KEYS, assert_bit_equal
from moonrtx_amd import _lib, dist as mdist
from moonrtx_amd.renderer import MoonRT, synth_ldem, dem_from_ldem
from moonrtx_amd.scene import named_scene
from oracle import orc

pytestmark = pytest.mark.gpu


def device_dem(h, w):
    src = synth_ldem(h, w)
    dem, scale = dem_from_ldem(src, h, w, 1)
    src.free()
    return dem, scale


def oracle_crops(scene, dem_np, lin, hits, crops, col=None):
    o = orc.Oracle(scene, dem_np, col)
    for (x0, y0, cw, ch) in crops:
        o.reset()
        o.render(1, (x0, y0, x0 + cw, y0 + ch))
        assert_bit_equal(lin[y0:y0 + ch, x0:x0 + cw], o.linear()[y0:y0 + ch, x0:x0 + cw], f"radiance crop at {x0},{y0}")
        assert_bit_equal(hits[y0:y0 + ch, x0:x0 + cw], o.hits[y0:y0 + ch, x0:x0 + cw], f"hits crop at {x0},{y0}")
    assert orc.quad_out_of_range() == 0


def test_cfg1_true_size_whole_frame_matches_the_oracle(native_lib):
    W = H = 512
    dem_b, scale = device_dem(5760, 11520)
    dem = dem_b.download(np.float32, (5760, 11520))
    assert dem.max() == 1.0 and 1.005 < scale < 1.0075
    for name, seg in (("S1", (1, 1)), ("S1", (2, 4)), ("S3", (1, 1))):
        s = named_scene(name, W, H, spp_per_launch=1)
        s.path_seg_min, s.path_seg_max = seg
        rt = MoonRT(W, H)
        rt.bind_dem(dem_b, 5760, 11520)
        rt.apply_scene(s); rt.set_params(flags=_lib.F_COUNT_STATS)
        st = rt.render(1)
        lin, hits = rt.read_linear(), rt.read_hits()
        rt.close()
        o = orc.Oracle(s, dem)
        st_o = o.render(1)
        assert_bit_equal(lin, o.linear(), f"cfg1 {name} {seg} radiance")
        assert_bit_equal(hits, o.hits, f"cfg1 {name} {seg} hits")
        assert {k: st[k] for k in STAT_KEYS} == {k: st_o[k] for k in STAT_KEYS}
        assert st["primary_rays"] == W * H and 0.55 < st["primary_hits"] / (W * H) < 0.70   # disc = 0.9 of the frame height
    dem_b.free()


@pytest.fixture(scope="module")
def cfg2_inputs(native_lib):
    dem_b, _ = device_dem(11520, 23040)
    dem = dem_b.download(np.float32, (11520, 23040))
    yield dem_b, dem
    dem_b.free()


@pytest.mark.parametrize("name,seg", [("S1", (1, 1)), ("S2", (1, 1)), ("S1", (2, 4))])
def test_cfg2_full_size(cfg2_inputs, name, seg):
    W, H, S = 1920, 1080, 16
    dem_b, dem = cfg2_inputs
    s = named_scene(name, W, H, spp_per_launch=S)
    s.path_seg_min, s.path_seg_max = seg
    frames = {}
    for flags in (_lib.F_COUNT_STATS, 0):
        rt = MoonRT(W, H)
        rt.bind_dem(dem_b, 11520, 23040)
        rt.apply_scene(s); rt.set_params(flags=flags)
        st = rt.render(1)
        frames[flags] = (rt.read_linear(), rt.read_hits(), st)
        rt.reset(); st2 = rt.render(1)
        assert_bit_equal(rt.read_linear(), frames[flags][0], "same frame twice")
        if flags:
            assert st2["height_samples"] == st["height_samples"]
        rt.close()
    lin, hits, st = frames[_lib.F_COUNT_STATS]
    assert_bit_equal(frames[0][0], lin, "production vs counting kernels, radiance")
    assert_bit_equal(frames[0][1], hits, "production vs counting kernels, hits")
    # size-independent properties
    assert st["primary_rays"] == W * H * S
    cov = float(lin[..., 3].astype(np.float64).sum())
    assert abs(cov / (np.pi / 4 * (0.9 * H) ** 2) - 1.0) < 0.03 and abs(cov * S - st["primary_hits"]) < 1.0
    hd = hits[..., 3]
    r = np.linalg.norm(hits[..., :3], axis=-1)[hd > 0]
    assert 9.88 < r.min() and r.max() <= 10.0 + 1e-5
    assert lin[:20].max() == 0.0 and lin[..., :3].max() > 0.1
    if seg[1] > 1:
        assert st["bounce_rays"] >= st["primary_hits"] and st["paths_ms"] > 0
    # the oracle on three crops: terminator, limb, disc centre
    oracle_crops(s, dem, lin, hits, [(700, 500, 64, 48), (1150, 180, 64, 48), (930, 520, 64, 48)])


def test_cfg4_one_rank_of_eight_at_full_size(native_lib):
    """7680x4320, 256 spp (4 blocks of 64), the full-resolution DEM, the colour map of cfg 3 ("colour as cfg 3"):
    rank 3 of 8 renders its tile lattice; the root (rank 0) renders its own and takes rank 3's shard in."""
    W, H, S, DH, DW, RANK, WORLD = 7680, 4320, 64, 46080, 92160, 3, 8
    CH, CW = 13680, 27360
    dem_b, scale = device_dem(DH, DW)
    assert 1.005 < scale < 1.0075
    from moonrtx_amd.renderer import synth_color, DeviceBuffer
    col_b = synth_color(CH, CW)
    s = named_scene("S1", W, H, spp_per_launch=S)
    s.max_spp = 256

    def render(seg, flags, blocks=4):
        s.path_seg_min, s.path_seg_max = seg
        rt = MoonRT(W, H, rank=RANK, world=WORLD)
        rt.bind_dem(dem_b, DH, DW)
        rt.bind_color(col_b, CH, CW)
        rt.apply_scene(s); rt.set_params(flags=flags)
        st = rt.render(blocks)
        out = rt.read_linear(), rt.read_hits(), st
        rt.close()
        return out

    lin, hits, st = render((1, 1), _lib.F_COUNT_STATS)
    lin2, hits2, _ = render((1, 1), 0)
    assert_bit_equal(lin2, lin, "cfg4 production vs counting radiance")
    assert_bit_equal(hits2, hits, "cfg4 production vs counting hits")
    # ownership: only this rank's tiles carry data (tile t -> rank t % world on the shifted lattice)
    tiles_x = (W + 31) // 32
    shift = mdist.tile_shift(WORLD)
    ty, tx = np.divmod(np.arange(tiles_x * ((H + 31) // 32)), tiles_x)
    tid = ty * tiles_x + (tx + shift * ty) % tiles_x          # tile number of the tile at (tx, ty): mrtx_tile_id()
    own = (tid % WORLD == RANK).reshape(-1, tiles_x)
    own_px = np.kron(own, np.ones((32, 32), bool))[:H, :W]
    assert lin[~own_px].max() == 0.0 and hits[~own_px].max() == 0.0
    # one eighth of the disc's coverage, 256 samples per owned pixel
    cov = float(lin[..., 3].astype(np.float64).sum())
    disc = np.pi / 4 * (0.9 * H) ** 2
    assert abs(cov / (disc / WORLD) - 1.0) < 0.05
    assert abs(cov * 256 - st["primary_hits"]) < 64.0
    assert st["primary_rays"] == int(own_px.sum()) * 256
    hd = hits[..., 3]
    r = np.linalg.norm(hits[..., :3], axis=-1)[hd > 0]
    assert 9.88 < r.min() and r.max() <= 10.0 + 1e-5
    # the reference's own path length through the queue-based stage, one block, production vs in-wave
    a, ha, sta = render((2, 4), 0, blocks=1)
    b, hb, _ = render((2, 4), _lib.F_INWAVE_PATHS, blocks=1)
    assert_bit_equal(a, b, "cfg4 shard: path queue vs in-wave radiance")
    assert sta["paths_ms"] > 0
    # the ROOT of the eight: rank 0 renders its own lattice (one block, (2, 4)), rank 3 packs what it rendered above (hit-less: the
    # final linear framebuffer is what the exchange moves) and the root unpacks it beside its own tiles
    r3 = MoonRT(W, H, rank=RANK, world=WORLD); r3.set_gather_hits(False)
    r3.bind_dem(dem_b, DH, DW); r3.bind_color(col_b, CH, CW); r3.apply_scene(s); r3.set_params(flags=0); r3.render(1)
    r0 = MoonRT(W, H, rank=0, world=WORLD); r0.set_gather_hits(False)
    r0.bind_dem(dem_b, DH, DW); r0.bind_color(col_b, CH, CW); r0.apply_scene(s); r0.set_params(flags=0); r0.render(1)
    assert r0.shard_bytes_active() == r3.shard_bytes_active() < r0.shard_bytes() == (W // 32) * (H // 32) // WORLD * 32 * 32 * 16
    buf = DeviceBuffer(r3.shard_bytes())
    r3.pack_shard(buf.ptr)
    own0 = np.kron((tid % WORLD == 0).reshape(-1, tiles_x), np.ones((32, 32), bool))[:H, :W]
    root_own = r0.read_linear()
    assert root_own[~own0].max() == 0.0
    r0.unpack_shard(RANK, buf.ptr)
    root = r0.read_linear()
    assert_bit_equal(root[own_px], a[own_px], "cfg4: rank 3's tiles on the root after the exchange")
    assert_bit_equal(root[own0], root_own[own0], "cfg4: the root's own tiles untouched by the unpack")
    assert root[~(own_px | own0)].max() == 0.0
    r0.close(); r3.close(); buf.free()
    # one oracle crop: an owned tile on the terminator side of the disc, first block of 64 spp, direct light
    dem = dem_b.download(np.float32, (DH, DW))
    dem_b.free()
    col = col_b.download(np.uint8, (CH, CW, 4))
    col_b.free()
    s.path_seg_min, s.path_seg_max = 1, 1
    cand = [(x, y) for y in range(60, 75) for x in range(90, 150) if own[y, x]]
    tx0, ty0 = cand[len(cand) // 2]
    rt = MoonRT(W, H, rank=RANK, world=WORLD)
    rt.upload_dem(dem)                                   # the host-upload path at 17 GB as well
    rt.upload_color(col)
    rt.apply_scene(s); rt.set_params(flags=0)
    rt.render(1)
    lin1, hits1 = rt.read_linear(), rt.read_hits()
    rt.close()
    oracle_crops(s, dem, lin1, hits1, [(tx0 * 32, ty0 * 32, 32, 32)], col=col)
    # ... and the same owned tile of the (2, 4) frame rendered above through the path queue (first block of 64 spp)
    s.path_seg_min, s.path_seg_max = 2, 4
    oracle_crops(s, dem, a, ha, [(tx0 * 32, ty0 * 32, 32, 32)], col=col)
