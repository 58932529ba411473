"""HIP path (through the C ABI) vs the CPU oracle on identical seeded inputs.

Bar: bit-exact linear radiance, hit buffer and sample counters (the north star asks for
L_inf < 2^-10 in linear radiance; both sides follow one arithmetic spec, so we hold them to 0).
The tone-mapped 8- and 16-bit read-backs are exact as well.
"""
import numpy as np
import pytest

import synth_np
from common import STAT_KEYS, assert_bit_equal, render_hip, render_oracle
from moonrtx_amd.scene import named_scene
from moonrtx_amd import renderer
from oracle import orc

pytestmark = pytest.mark.gpu

L_INF_BAR = 2.0 ** -10


@pytest.fixture(scope="module")
def dem_small():
    return synth_np.dem(360, 720, seed=5, craters=60)


def check(scene, dem, color=None, bg=None, blocks=(1,)):
    lin_h, hits_h, st_h, _ = render_hip(scene, dem, color, bg, blocks)
    lin_o, hits_o, st_o = render_oracle(scene, dem, color, bg, blocks)
    assert np.abs(lin_h - lin_o).max() < L_INF_BAR
    assert_bit_equal(lin_h, lin_o, "linear radiance")
    assert_bit_equal(hits_h, hits_o, "hit buffer")
    assert {k: st_h[k] for k in STAT_KEYS} == {k: st_o[k] for k in STAT_KEYS}
    assert lin_o[..., :3].max() > 0.01, "scene rendered black: test is vacuous"
    return lin_h, st_h


def test_math_primitives_bit_exact(native_lib):
    rng = np.random.default_rng(3)
    n = 20000
    pts = rng.standard_normal((n, 3)).astype(np.float32) * 10
    special = np.array([[0, 0, 10], [0, 0, -10], [0, 10, 0], [0, -10, 0], [10, 0, 0], [-10, 0, 0], [0, 0, 0],
                        [1e-30, 0, 1], [0, 1e-30, -1], [1e-20, -1e-20, 5], [3, -3, 3], [-7, -7, 0]], np.float32)
    pts = np.concatenate([pts, special])
    lat_d, lon_d = renderer.probe_latlon(pts[:, 0], pts[:, 1], pts[:, 2])
    lat_o, lon_o = orc.latlon(pts[:, 0], pts[:, 1], pts[:, 2])
    assert_bit_equal(lat_d, lat_o, "lat polynomial")
    assert_bit_equal(lon_d, lon_o, "lon polynomial")
    p64 = pts[:n].astype(np.float64)
    assert np.abs(lat_o[:n] - np.arctan2(p64[:, 2], np.hypot(p64[:, 0], p64[:, 1]))).max() < 6e-7
    assert np.abs(lon_o[:n] - np.arctan2(p64[:, 0], p64[:, 1])).max() < 6e-7


def test_domain_restricted_reciprocal_and_sqrt_are_ieee_exact(native_lib):
    """The kernels take 1/x as v_rcp_f32 + one Newton step and sqrt as v_sqrt_f32 + a residual fix where the argument's range is
    known (mrtx_kernels.hip: rcp_cr, sqrt_cr); the oracle uses the C compiler's IEEE division and sqrtf.  Equality is checked
    EXHAUSTIVELY on the device against the compiler's IEEE expansions: every normal float of either sign whose reciprocal is normal
    (exponents -126 .. 125), and every float from 2^-104 up plus zero for the square root."""
    import ctypes as C
    def probe(which, lo_bits, n):
        bad, first = C.c_uint64(), C.c_uint32()
        assert native_lib.mrtx_probe_cr(0, which, lo_bits, n, C.byref(bad), C.byref(first)) == 0
        return bad.value, first.value
    for sign in (0, 1):                                                  # reciprocal, one Newton step (what ships)
        lo = (sign << 31) | (1 << 23)                                    # exponent field 1 (2^-126) ... 252 (2^125), all mantissas
        assert probe(0, lo, (252 << 23)) == (0, 0)
        assert probe(1, lo, (252 << 23)) == (0, 0)                       # two steps: the same
    bad, first = probe(0, 253 << 23, 1 << 23)                            # 2^126 and beyond: 1/x is subnormal -- outside the domain
    assert bad > 0 and first >= (253 << 23)
    assert probe(2, 0, 1) == (0, 0)                                      # sqrt_cr(+0) = 0
    assert probe(2, (127 - 104) << 23, (255 << 23) - ((127 - 104) << 23)) == (0, 0)      # 2^-104 ... the largest finite float
    assert probe(2, 1 << 23, 1 << 23)[0] > 0                             # 2^-126: outside sqrt_cr's domain (never passed in)


@pytest.mark.parametrize("name", ["S1", "S2", "S3"])
def test_first_light_1spp(native_lib, dem_small, name):
    """BASELINE config 1 shape: 1 spp, grey albedo (reduced image so the oracle takes seconds)."""
    check(named_scene(name, 192, 160, spp_per_launch=1), dem_small)


@pytest.mark.parametrize("spp", [2, 4, 8, 16, 32, 64])
def test_wave_packing_all_spp(native_lib, dem_small, spp):
    check(named_scene("S1", 72, 56, spp_per_launch=spp), dem_small)


def test_accumulation_blocks_match_oracle_and_single_launch(native_lib, dem_small):
    s = named_scene("S1", 64, 48, spp_per_launch=16)
    lin_a, _ = check(s, dem_small, blocks=(1, 1, 2))
    lin_b, _ = check(s, dem_small, blocks=(4,))
    assert_bit_equal(lin_a, lin_b, "1+1+2 blocks vs 4 blocks in one launch")


def test_colour_texture_background_sun_disk(native_lib, dem_small):
    col = synth_np.colour_map(180, 360)
    rng = np.random.default_rng(9)
    bg = rng.integers(0, 255, (64, 128, 4), dtype=np.uint8)
    s = named_scene("S2", 160, 96, spp_per_launch=4)
    s.sun_pos, s.sun_radius = (40.0, 3100.0 - 300.0, 25.0), 30.0   # in view, beside the disk
    _, st = check(s, dem_small, col, bg)
    assert st["colour_fetches"] > 0 and st["background_fetches"] > 0


def test_odd_sizes_and_zoomed_camera(native_lib, dem_small):
    s = named_scene("S1", 77, 45, spp_per_launch=8)
    s.vfov_deg = 0.9
    s.target = (3.0, 0.0, 4.0)
    check(s, dem_small)


def test_ragged_dem_shapes(native_lib):
    """Non power-of-two, non 2:1 DEMs, and the smallest legal one."""
    check(named_scene("S2", 48, 48, spp_per_launch=4), synth_np.dem(181, 359, seed=2, craters=10))
    tiny = np.array([[1.0, 0.99], [0.98, 1.0]], np.float32)
    check(named_scene("S2", 32, 32, spp_per_launch=4), tiny)


def test_maxmip_skip_is_result_preserving(native_lib, dem_small):
    """The max-mip step skip must change nothing but the number of DEM evaluations performed."""
    from moonrtx_amd import _lib
    from moonrtx_amd.renderer import MoonRT
    for name, spp in (("S1", 16), ("S3", 4)):
        s = named_scene(name, 120, 90, spp_per_launch=spp)
        out = {}
        for tag, flags in (("skip", _lib.F_COUNT_STATS), ("full", _lib.F_COUNT_STATS | _lib.F_NO_SKIP)):
            rt = MoonRT(s.width, s.height)
            rt.upload_dem(dem_small); rt.apply_scene(s); rt.set_params(flags=flags)
            st = rt.render(1)
            out[tag] = (rt.read_linear(), rt.read_hits(), st)
            rt.close()
        assert_bit_equal(out["skip"][0], out["full"][0], "skip vs full radiance")
        assert_bit_equal(out["skip"][1], out["full"][1], "skip vs full hits")
        a, b = out["skip"][2], out["full"][2]
        assert {k: a[k] for k in STAT_KEYS} == {k: b[k] for k in STAT_KEYS}
        assert b["dem_fetches"] >= b["height_samples"] and b["mip_fetches"] == 0
        assert a["dem_fetches"] < 0.7 * b["dem_fetches"] and a["mip_fetches"] > 0


def seam_ridge_case(width=96, height=64, spp=4):
    """A meridian ridge (D = 1) just WEST of the +-180 seam, in the last, partial fine-mip cell of a DEM whose width is not
    a multiple of the cell (731 = 45 x 16 + 11; the ridge, columns 723..727, lies in no neighbouring cell's two-texel
    border), a low Sun in the west and a close-up of the seam: the ridge's shadow falls across the seam onto columns 0..3,
    whose shadow rays travel westward through column 0 -- the horizon-mip cell (i, 0) must know about the ridge."""
    from moonrtx_amd import scene as sc
    h, w = 366, 731
    rng = np.random.default_rng(4)
    dem = np.full((h, w), 0.992, np.float32) + rng.random((h, w)).astype(np.float32) * np.float32(2e-4)
    dem[:, 723:728] = 1.0
    s = sc.make_scene(width, height, 82.0, 90.0, spp_per_launch=spp, libration=(0.0, 0.0))
    s.u, s.v = (0.0, 0.0, 1.0), (0.0, 1.0, 0.0)       # lon 180 faces the camera: the seam runs down the middle of the frame
    s.vfov_deg = 0.6
    return s, dem


@pytest.mark.parametrize("segs", [(1, 1), (2, 4)])
def test_horizon_mip_sees_the_partial_cell_across_the_seam(native_lib, segs):
    """Round-2 advisor finding: hmip_build_kernel stepped texel columns by the cell size across the seam and never visited
    the last, partial fine cell when dem_w % cell != 0, so horizon_kend could end a shadow ray below a peak next to the
    seam (lit pixels where the spec says shadow).  HIP == oracle, with the skip logic and without."""
    from moonrtx_amd import _lib
    from moonrtx_amd.renderer import MoonRT
    s, dem = seam_ridge_case()
    s.path_seg_min, s.path_seg_max = segs
    lin_o, hits_o, st_o = render_oracle(s, dem)
    # the case is what it claims to be: the ridge shadows pixels just east of the seam
    flat = dem.copy(); flat[:, 723:728] = flat[:, 700:705]; flat[0, 0] = 1.0
    lin_f, hits_f, _ = render_oracle(s, flat)
    lon = np.degrees(np.arctan2(-hits_f[..., 0], hits_f[..., 1]))
    col = lon * (731 / 360.0) + 731 / 2.0 - 0.5
    east_of_seam = (col > -0.5) & (col < 4.0)
    assert ((lin_o[..., 0] < 0.5 * lin_f[..., 0]) & (lin_f[..., 0] > 0.01) & east_of_seam).sum() > 200
    for flags in (_lib.F_COUNT_STATS, 0, _lib.F_COUNT_STATS | _lib.F_NO_SKIP):
        lin_h, hits_h, st_h, _ = render_hip(s, dem, flags=flags)
        assert_bit_equal(lin_h, lin_o, f"radiance, flags {flags}")
        assert_bit_equal(hits_h, hits_o, f"hit buffer, flags {flags}")
        if flags & _lib.F_COUNT_STATS:
            assert {k: st_h[k] for k in STAT_KEYS} == {k: st_o[k] for k in STAT_KEYS}


def test_sky_tile_cull_is_result_preserving(native_lib, dem_small):
    """Host-side cull of tiles that cannot see the Moon or the Sun disk: same frame, same counters."""
    from moonrtx_amd import _lib
    from moonrtx_amd.renderer import MoonRT
    s = named_scene("S1", 400, 300, spp_per_launch=4)
    s.sun_pos, s.sun_radius = (150.0, 3100.0 - 300.0, -80.0), 25.0        # a visible Sun disk off to the side
    out = {}
    for tag, flags in (("cull", _lib.F_COUNT_STATS), ("all", _lib.F_COUNT_STATS | _lib.F_NO_CULL)):
        rt = MoonRT(s.width, s.height, tile=(16, 16))
        rt.upload_dem(dem_small); rt.apply_scene(s); rt.set_params(flags=flags)
        st = rt.render(1)
        out[tag] = (rt.read_linear(), rt.read_hits(), st)
        # a second view through the same context: the cull set changes, stale pixels must not survive
        rt.set_camera((0.0, -300.0, 0.0), (6.0, 0.0, -5.0), (0, 0, 1), 1.5)
        rt.reset(); st2 = rt.render(1)
        out[tag + "2"] = (rt.read_linear(), rt.read_hits(), st2)
        rt.close()
    for a, b in (("cull", "all"), ("cull2", "all2")):
        assert_bit_equal(out[a][0], out[b][0], "cull vs all radiance")
        assert_bit_equal(out[a][1], out[b][1], "cull vs all hits")
        assert {k: out[a][2][k] for k in STAT_KEYS} == {k: out[b][2][k] for k in STAT_KEYS}
    assert out["cull"][0][..., :3].max() >= 2.0 - 1e-6      # the Sun disk is there (radiance 2.0)


@pytest.mark.parametrize("seg", [(2, 2), (2, 4), (1, 3)])
def test_multi_bounce_paths_match_oracle(native_lib, dem_small, seg):
    """D6: path continuation with Russian roulette, next-event estimation at every vertex, environment on escape."""
    col = synth_np.colour_map(90, 180)
    rng = np.random.default_rng(11)
    bg = rng.integers(0, 255, (32, 64, 4), dtype=np.uint8)
    rough = synth_np.corrugated_dem(720, 1440)      # steep relief: bounce rays really do hit terrain again
    s = named_scene("S1", 96, 72, spp_per_launch=16)
    s.path_seg_min, s.path_seg_max = seg
    _, st = check(s, rough, col, bg)
    assert st["bounce_rays"] > 0 and st["shadow_rays"] > 0
    if seg[0] >= 2:
        assert st["bounce_rays"] >= st["primary_hits"]             # the first continuation is guaranteed
        if seg[1] > 2:
            assert st["bounce_rays"] > st["primary_hits"]          # some paths went on to a third segment
    s2 = named_scene("S3", 64, 48, spp_per_launch=4)
    s2.path_seg_min, s2.path_seg_max = seg
    check(s2, dem_small, blocks=(2, 1))


def test_edge_case_parameters_match_oracle(native_lib, dem_small):
    """Degenerate but legal inputs: the spec must stay defined and both sides must agree."""
    # 1x1 and sliver frames
    for (w, h, spp) in ((1, 1, 64), (3, 1, 16), (1, 5, 4), (17, 3, 1)):
        s = named_scene("S2", w, h, spp_per_launch=spp)
        lin_h, hits_h, st_h, _ = render_hip(s, dem_small)
        lin_o, hits_o, st_o = render_oracle(s, dem_small)
        assert_bit_equal(lin_h, lin_o, f"{w}x{h} radiance"); assert_bit_equal(hits_h, hits_o, f"{w}x{h} hits")
        assert {k: st_h[k] for k in STAT_KEYS} == {k: st_o[k] for k in STAT_KEYS}
    # no bisection (eps >= step), a huge step, no lift of the shadow origin
    s = named_scene("S1", 64, 48, spp_per_launch=4); s.marching_step_eps = 1.0e-2
    check(s, dem_small)
    s = named_scene("S1", 64, 48, spp_per_launch=4); s.marching_step = 0.05; s.marching_step_eps = 2.0e-3
    check(s, dem_small)
    s = named_scene("S1", 64, 48, spp_per_launch=4); s.scene_epsilon = 0.0
    check(s, dem_small)
    # a point light (radius 0) still lights the surface; zero brightness renders black but hits are recorded
    s = named_scene("S2", 48, 48, spp_per_launch=4); s.light_radius = 0.0
    lin_h, hits_h, _, _ = render_hip(s, dem_small); lin_o, hits_o, _ = render_oracle(s, dem_small)
    assert_bit_equal(lin_h, lin_o, "point light"); assert lin_o[..., :3].max() == 0.0      # zero solid angle, zero power
    s = named_scene("S2", 48, 48, spp_per_launch=4); s.light_radiance = 0.0
    lin_h, hits_h, _, _ = render_hip(s, dem_small); lin_o, hits_o, _ = render_oracle(s, dem_small)
    assert_bit_equal(lin_h, lin_o, "dark"); assert_bit_equal(hits_h, hits_o, "dark hits"); assert (hits_o[..., 3] > 0).any()
    # eye INSIDE the bounding sphere (between the sphere and the terrain), looking along the surface
    s = named_scene("S2", 64, 48, spp_per_launch=4, libration=(0.0, 0.0))
    s.eye = (0.0, -9.999, 0.0); s.target = (3.0, -9.6, 1.0); s.vfov_deg = 60.0
    check(s, dem_small)
    # a wide field of view with the Moon small in the frame, and a camera roll
    s = named_scene("S1", 80, 60, spp_per_launch=4); s.vfov_deg = 40.0; s.up = (0.3, 0.0, 1.0)
    check(s, dem_small)


def test_overlay_tubes_match_oracle(native_lib, dem_small):
    """D11 (set_graph): flat, non-shadowing tubes outside the bounding sphere; binned per tile on the host, so the
    nearest hit over a tile's bin must equal the oracle's search over every capsule."""
    from moonrtx_amd import overlays, _lib
    from moonrtx_amd.renderer import MoonRT
    s = named_scene("S1", 200, 150, spp_per_launch=8)
    pos, edges, r, c = overlays.graticule(rotation=s.rotation, tube=0.02)
    caps = overlays.graph_to_capsules(pos, edges, r, c)
    # a "label" graph with per-vertex radii (night-side labels hidden by zero radii, renderer_labels.py:126-128)
    lp = np.array([[2.0, -10.3, 1.0], [2.6, -10.25, 1.0], [2.6, -10.25, 1.5], [-12.0, 0.0, 3.0], [-12.5, 0.0, 3.5]])
    caps = np.concatenate([caps, overlays.graph_to_capsules(lp, [[0, 1], [1, 2], [3, 4]], np.array([0.03, 0.03, 0.03, 0.0, 0.0]),
                                                            [1.0, 0.9, 0.3])])
    lin_h, hits_h, st_h, _ = render_hip(s, dem_small, capsules=caps, tile=(16, 16))
    lin_o, hits_o, st_o = render_oracle(s, dem_small, capsules=caps)
    assert_bit_equal(lin_h, lin_o, "overlay radiance"); assert_bit_equal(hits_h, hits_o, "overlay hits")
    assert {k: st_h[k] for k in STAT_KEYS} == {k: st_o[k] for k in STAT_KEYS}
    base, _, st_b, _ = render_hip(s, dem_small)
    touched = (lin_h != base).any(-1)
    assert touched.sum() > 1500 and st_h["primary_hits"] < st_b["primary_hits"]    # tubes hide some terrain samples
    night = (base[..., 3] > 0.99) & (base[..., 0] == 0.0)                           # night-side pixels: tubes still show
    assert (lin_h[night][:, 0] > 0).sum() > 50
    yellow = (lin_h[..., 0] - lin_h[..., 2]) - (base[..., 0] - base[..., 2])        # the label is (1, 0.9, 0.3)
    assert yellow.max() > 0.05
    # zoomed view + sharding + removal
    s2 = named_scene("S2", 96, 64, spp_per_launch=4); s2.vfov_deg = 0.8; s2.target = (2.3, 0.0, 1.2)
    for world in (1, 2):
        rts = []
        for rk in range(world):
            rt = MoonRT(s2.width, s2.height, rank=rk, world=world, tile=(16, 16))
            rt.upload_dem(dem_small); rt.apply_scene(s2); rt.set_capsules(caps); rt.render(1)
            rts.append(rt)
        if world == 2:
            from moonrtx_amd.renderer import DeviceBuffer
            buf = DeviceBuffer(rts[1].shard_bytes()); rts[1].pack_shard(buf.ptr); rts[0].unpack_shard(1, buf.ptr)
        lin2 = rts[0].read_linear()
        lin2_o, _, _ = render_oracle(s2, dem_small, capsules=caps)
        assert_bit_equal(lin2, lin2_o, f"zoomed overlay, world={world}")
        if world == 1:
            rts[0].set_capsules(None); rts[0].reset(); rts[0].render(1)
            plain, _, _ = render_oracle(s2, dem_small)
            assert_bit_equal(rts[0].read_linear(), plain, "overlay removed")
        for rt in rts:
            rt.close()


def test_reference_moon_grid_overlay_matches_oracle(native_lib, dem_small):
    """The reference's own overlay graphs (golden from moonrtx.moon_grid: 3267 grid-line edges at r = 0.006 + 633 label
    edges at r = 0.012, colour 0.5; renderer_labels.py:24-28, :291-300) rotated by the libration, over the terrain."""
    import os
    from moonrtx_amd import overlays
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "moon_grid_graphs.npz"))
    s = named_scene("S1", 240, 160, spp_per_launch=8)
    R = np.asarray(s.rotation, float)
    caps = np.concatenate([overlays.graph_to_capsules(g["lines_pos"] @ R.T, g["lines_edges"], 0.006, [0.5, 0.5, 0.5]),
                           overlays.graph_to_capsules(g["labels_pos"] @ R.T, g["labels_edges"], 0.012, [0.5, 0.5, 0.5])])
    assert caps.shape == (3900, 12)
    for scene, tile in ((s, (32, 32)), (named_scene("S2", 160, 120, spp_per_launch=16, libration=(40.0, 25.0)), (16, 16))):
        if scene is not s:
            scene.vfov_deg = 1.1; scene.target = (1.0, 0.0, 2.0)          # zoom on a grid crossing with labels
            Rz = np.asarray(scene.rotation, float)
            caps = np.concatenate([overlays.graph_to_capsules(g["lines_pos"] @ Rz.T, g["lines_edges"], 0.006, [0.5, 0.5, 0.5]),
                                   overlays.graph_to_capsules(g["labels_pos"] @ Rz.T, g["labels_edges"], 0.012, [0.5, 0.5, 0.5])])
        lin_h, hits_h, st_h, _ = render_hip(scene, dem_small, capsules=caps, tile=tile)
        lin_o, hits_o, st_o = render_oracle(scene, dem_small, capsules=caps)
        assert_bit_equal(lin_h, lin_o, "grid overlay radiance"); assert_bit_equal(hits_h, hits_o, "grid overlay hits")
        assert {k: st_h[k] for k in STAT_KEYS} == {k: st_o[k] for k in STAT_KEYS}
        base, _, _, _ = render_hip(scene, dem_small, tile=tile)
        assert ((lin_h != base).any(-1)).sum() > 300                          # the grid is visible


def test_other_radius_centre_step_and_deep_relief(native_lib, dem_small):
    """Nothing may be tied to R = 10 at the origin, step 5e-3 or ~1 % relief (skip bounds, mip cell, guards)."""
    from moonrtx_amd import overlays
    s = named_scene("S1", 96, 72, spp_per_launch=8)
    k = 0.37
    off = np.array([3.0, -2.0, 1.5])
    s.radius = 10.0 * k
    s.center = tuple(off)
    s.eye = tuple(off + k * np.array(s.eye)); s.target = tuple(off)
    s.light_pos = tuple(off + np.array(s.light_pos)); s.sun_pos = tuple(off + np.array(s.sun_pos))
    s.marching_step = 2.2e-3; s.marching_step_eps = 1.0e-4; s.scene_epsilon = 4.0e-5
    check(s, dem_small)
    pos, edges, r, c = overlays.graticule(radius=10.25 * k, rotation=s.rotation, tube=0.02 * k)
    caps = overlays.graph_to_capsules(pos + off, edges, r, c)
    lin_h, hits_h, st_h, _ = render_hip(s, dem_small, capsules=caps, tile=(16, 16))
    lin_o, hits_o, st_o = render_oracle(s, dem_small, capsules=caps)
    assert_bit_equal(lin_h, lin_o, "scaled scene with overlay"); assert_bit_equal(hits_h, hits_o, "scaled hits")
    # 10 % relief, coarse DEM, big steps relative to the texel
    rng = np.random.default_rng(5)
    deep = (0.9 + 0.1 * rng.random((40, 80))).astype(np.float32); deep[7, 9] = 1.0
    s2 = named_scene("S1", 80, 60, spp_per_launch=8); s2.path_seg_min, s2.path_seg_max = 2, 3
    check(s2, deep)


def test_wide_addressing_path_matches(native_lib, dem_small):
    """DEMs above 4 GiB (downscale 1: 17 GB) take 64-bit byte offsets; force that path on a small DEM."""
    from moonrtx_amd import _lib
    s = named_scene("S1", 80, 64, spp_per_launch=16)
    col = synth_np.colour_map(90, 180)
    lin_w, hits_w, st_w, _ = render_hip(s, dem_small, col, flags=_lib.F_COUNT_STATS | _lib.F_FORCE_WIDE)
    lin_o, hits_o, st_o = render_oracle(s, dem_small, col)
    assert_bit_equal(lin_w, lin_o, "wide path radiance")
    assert_bit_equal(hits_w, hits_o, "wide path hits")
    assert {k: st_w[k] for k in STAT_KEYS} == {k: st_o[k] for k in STAT_KEYS}


def test_polar_and_seam_views_use_exact_segments(native_lib, dem_small):
    """Cameras over the poles and over the +/-180 seam: the segments there fall back to exact evaluation."""
    from oracle import orc
    for lib in ((0.0, 89.0), (0.0, -88.0), (179.5, 5.0), (-179.8, -40.0)):
        s = named_scene("S2", 64, 64, spp_per_launch=4, libration=lib)
        check(s, dem_small)
    assert orc.quad_out_of_range() == 0


def test_sharded_ranks_reassemble_bit_exact(native_lib, dem_small):
    """world=2 and world=3 on one GPU: each rank renders its tiles, rank 0 unpacks the peers' shards."""
    from moonrtx_amd.renderer import MoonRT, DeviceBuffer
    s = named_scene("S1", 100, 70, spp_per_launch=4)
    lin_1, hits_1, _, _ = render_hip(s, dem_small)
    for world in (2, 3):
        rts = []
        for r in range(world):
            rt = MoonRT(s.width, s.height, rank=r, world=world, tile=(16, 16))
            rt.upload_dem(dem_small); rt.apply_scene(s); rt.render(1)
            rts.append(rt)
        bufs = []
        for rt in rts[1:]:
            b = DeviceBuffer(rt.shard_bytes())
            rt.pack_shard(b.ptr)
            bufs.append(b)
        for r, b in enumerate(bufs, start=1):
            rts[0].unpack_shard(r, b.ptr)
        assert_bit_equal(rts[0].read_linear(), lin_1, f"world={world} reassembled radiance")
        assert_bit_equal(rts[0].read_hits(), hits_1, f"world={world} reassembled hits")
        for rt in rts:
            rt.close()


def test_gather_moves_only_active_tiles_and_clears_stale_ones(native_lib, dem_small):
    """The exchange's active layout: only tiles the sky cull keeps travel; after a view change the root's copies of
    peer tiles that became sky read as zero; with the cull disabled the full layout is used.  All bit-exact."""
    from dataclasses import replace
    from moonrtx_amd.renderer import MoonRT, DeviceBuffer
    from moonrtx_amd import _lib
    a = named_scene("S1", 192, 96, spp_per_launch=4)                       # disc in the middle of a wide frame
    b = replace(a, target=(14.0, 0.0, 5.0))                                # disc pushed towards a corner
    for world in (2, 3):
        rts = [MoonRT(a.width, a.height, rank=r, world=world, tile=(16, 16)) for r in range(world)]
        bufs = [DeviceBuffer(rt.shard_bytes()) for rt in rts]
        for rt in rts:
            rt.upload_dem(dem_small)
        for scene, flags in ((a, 0), (b, 0), (a, _lib.F_NO_CULL), (b, 0)):
            want = MoonRT(a.width, a.height, tile=(16, 16))
            want.upload_dem(dem_small); want.apply_scene(scene); want.render(1)
            sizes = []
            for rt, buf in zip(rts, bufs):
                rt.apply_scene(scene); rt.set_params(flags=flags); rt.reset(); rt.render(1)
                sizes.append(rt.shard_bytes_active())
                rt.pack_shard(buf.ptr)
            assert len(set(sizes)) == 1
            if flags == 0:
                assert 0 < sizes[0] < 0.7 * rts[0].shard_bytes(), (sizes, rts[0].shard_bytes())
            else:
                assert sizes[0] == rts[0].shard_bytes()
            rts[0].unpack_all([buf.ptr for buf in bufs])
            assert_bit_equal(rts[0].read_linear(), want.read_linear(), f"world={world} radiance")
            assert_bit_equal(rts[0].read_hits(), want.read_hits(), f"world={world} hits")
            # the same exchange in parts (what FrameGather.render_and_gather overlaps with rendering)
            for P in (2, 3):
                if flags != 0:
                    assert rts[0].shard_parts(P) == 1            # full layout: no parts
                    continue
                assert rts[0].shard_parts(P) == P
                pieces = []
                for rt, buf in zip(rts, bufs):
                    rt.reset()
                    buf.upload(np.zeros(buf.nbytes, np.uint8))
                    cover = []
                    for k in range(P):
                        rt.render_part(1, k, P)
                        cover.append(rt.pack_part(buf.ptr, k, P))
                    assert rt.samples_done() == scene.spp_per_launch
                    pieces.append(cover)
                assert all(p == pieces[0] for p in pieces)        # every rank cuts at the same bytes
                assert pieces[0][0][0] == 0 and sum(ln for _, ln in pieces[0]) == sizes[0]
                assert all(pieces[0][k][0] + pieces[0][k][1] == pieces[0][k + 1][0] for k in range(P - 1))
                rts[0].unpack_all([buf.ptr for buf in bufs])
                assert_bit_equal(rts[0].read_linear(), want.read_linear(), f"world={world} parts={P} radiance")
                assert_bit_equal(rts[0].read_hits(), want.read_hits(), f"world={world} parts={P} hits")
            want.close()
        for rt in rts:
            rt.close()


def test_hitless_exchange_config_and_stage_counters(native_lib, dem_small):
    """ABI 7: mrtx_set_gather_hits(0) halves the shard (the final linear framebuffer is all that travels) and leaves the root's
    hit buffer to its own tiles, one texel of a peer's tile coming from mrtx_read_hit on that peer; mrtx_get_config reports the
    tiling in force; MrtxStats::camera_* split the counters between render_kernel and path_kernel."""
    from moonrtx_amd.renderer import MoonRT, DeviceBuffer
    from moonrtx_amd import _lib, dist as mdist
    s = named_scene("S1", 100, 70, spp_per_launch=4)
    s.path_seg_min, s.path_seg_max = 2, 4
    lin_1, hits_1, st_1, _ = render_hip(s, dem_small, tile=(16, 16))
    # stage counters: with the queue the camera stage holds part of every total, in the wave all of it
    for k in ("height_samples", "dem_fetches", "mip_fetches", "colour_fetches", "background_fetches"):
        assert 0 < st_1["camera_" + k] <= st_1[k] if st_1[k] else st_1["camera_" + k] == 0, (k, st_1)
    assert st_1["camera_dem_fetches"] < st_1["dem_fetches"] and st_1["camera_height_samples"] < st_1["height_samples"]
    _, _, st_w, _ = render_hip(s, dem_small, tile=(16, 16), flags=_lib.F_COUNT_STATS | _lib.F_INWAVE_PATHS)
    assert all(st_w["camera_" + k] == st_w[k] == st_1[k] for k in ("height_samples", "colour_fetches"))
    # the evaluations actually PERFORMED are an implementation count: path_kernel cuts every march's skip intervals with the medium
    # mip, the in-wave kernel only those of its camera and shadow rays (round 4), so the two modes agree in the spec counters only
    assert st_w["camera_dem_fetches"] == st_w["dem_fetches"] > 0
    d = MoonRT(64, 48); cfg1 = d.config(); d.close()
    d = MoonRT(64, 48, rank=1, world=4); cfg4 = d.config(); d.close()
    assert (cfg1["tile_w"], cfg1["tile_h"], cfg1["world"]) == (16, 16, 1) and (cfg4["tile_w"], cfg4["tile_h"], cfg4["rank"], cfg4["world"]) == (32, 32, 1, 4)
    for world in (2, 3):
        rts = []
        for r in range(world):
            rt = MoonRT(s.width, s.height, rank=r, world=world, tile=(16, 16))
            full = rt.shard_bytes()
            rt.set_gather_hits(False)
            assert rt.shard_bytes() * 2 == full
            rt.upload_dem(dem_small); rt.apply_scene(s); rt.render(1)
            rts.append(rt)
        assert rts[0].shard_bytes_active() < rts[0].shard_bytes()
        bufs = [DeviceBuffer(rt.shard_bytes()) for rt in rts]
        for rt, b in zip(rts, bufs):
            rt.pack_shard(b.ptr)
        rts[0].unpack_all([b.ptr for b in bufs])
        assert_bit_equal(rts[0].read_linear(), lin_1, f"world={world} radiance through the hit-less exchange")
        own = np.zeros((s.height, s.width), bool)
        for y in range(0, s.height, 16):
            for x in range(0, s.width, 16):
                own[y:y + 16, x:x + 16] = mdist.tile_owner(x, y, s.width, (16, 16), world) == 0
        got = rts[0].read_hits()
        assert np.array_equal(got[own].view(np.uint32), hits_1[own].view(np.uint32)) and not got[~own].any()
        for (x, y) in ((50, 35), (20, 30), (70, 40), (45, 20)):          # a pick: the owner's one texel
            o = mdist.tile_owner(x, y, s.width, (16, 16), world)
            assert np.array_equal(np.array(rts[o].read_hit(x, y), np.float32).view(np.uint32), hits_1[y, x].view(np.uint32)), (x, y, o)
        assert hits_1[35, 50, 3] > 0
        for rt in rts:
            rt.close()
        for b in bufs:
            b.free()


_RCCL_SNIPPET = r"""
import os, sys, tempfile, torch, torch.distributed as dist
with tempfile.TemporaryDirectory() as d:
    dist.init_process_group("nccl", init_method="file://" + os.path.join(d, "store"), rank=0, world_size=1)
    torch.cuda.set_device(0)
    send = torch.arange(4096, dtype=torch.float32, device="cuda")
    recv = [torch.zeros_like(send)]
    works = [dist.gather(send[a:b], [t[a:b] for t in recv], dst=0, async_op=True) for a, b in ((0, 1000), (1000, 4096))]
    for w in works:
        w.wait()
    torch.cuda.synchronize()
    ok = torch.equal(recv[0], send)
    dist.destroy_process_group()
sys.exit(0 if ok else 3)
"""


def test_rccl_async_gather_of_views():
    """The collective calls FrameGather.render_and_gather() makes -- async gathers of slices of one buffer into slices of
    the receive buffers -- on the real RCCL backend (one rank is all a one-GPU box can host).  Own process: torch
    must bring up its HIP runtime before anything else touches the device, as in bench.py."""
    import subprocess, sys
    r = subprocess.run([sys.executable, "-c", _RCCL_SNIPPET], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]


def test_one_hip_runtime_whatever_the_import_order(native_lib, dem_small):
    """libmoonrt.so FIRST (this test session loaded it long ago), then torch + an RCCL group, in ONE process: the loader
    preloads the torch wheel's own libamdhip64 before libmoonrt.so, so both sides share a single HIP/HSA runtime --
    with two copies the later one sees no GPUs ("ProcessGroupNCCL is only supported with GPUs, no GPUs found")."""
    import os
    from moonrtx_amd import _lib, dist as mdist
    from moonrtx_amd.renderer import MoonRT
    s = named_scene("S1", 96, 64, spp_per_launch=8)
    rt = MoonRT(s.width, s.height)                      # the HIP runtime is initialised through libmoonrt.so here
    rt.upload_dem(dem_small); rt.apply_scene(s)
    rt.render(1)
    ref = rt.read_linear()
    import torch
    import torch.distributed as dist
    assert len(_lib.assert_single_hip_runtime()) == 1
    assert torch.cuda.is_available() and torch.cuda.device_count() >= 1
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        assert dist.get_backend() == "nccl"
        g = mdist.FrameGather(rt, torch.device("cuda", 0))
        rt.reset()
        st = g.render_and_gather(1)                     # world 1: render + (no-op) gather through the same entry point
        assert st["primary_rays"] == s.width * s.height * 8
        assert_bit_equal(rt.read_linear(), ref, "frame after the RCCL group came up in the same process")
        t = torch.ones(4, device="cuda")
        dist.all_reduce(t)                              # RCCL really talks to the device
        assert float(t.sum()) == 4.0
    finally:
        if created:
            dist.destroy_process_group()
        rt.close()


def test_tonemapped_read_backs_equal_the_oracle(native_lib, dem_small):
    """The 8-bit image the GUI shows / save_image writes and the 16-bit save_image path (moon_renderer.py:598-600,
    renderer_dialogs.py:1222-1224): byte for byte the oracle's, for three gammas and two exposures, with an overlay."""
    from moonrtx_amd.renderer import MoonRT
    from oracle import orc
    s = named_scene("S1", 96, 64, spp_per_launch=8)
    s.path_seg_min, s.path_seg_max = 2, 4
    o = orc.Oracle(s, dem_small)
    o.render(1)
    rt = MoonRT(s.width, s.height)
    try:
        rt.upload_dem(dem_small); rt.apply_scene(s); rt.render(1)
        assert_bit_equal(rt.read_linear(), o.linear(), "linear radiance")
        rng = np.random.default_rng(3)
        ov = rng.integers(0, 256, size=(s.height, s.width, 4), dtype=np.uint8)
        ov[:, : s.width // 2, 3] = 0
        n8 = n16 = 0
        for gamma, expo in [(2.2, 0.9), (0.5, 0.9), (5.0, 0.9), (2.2, 4.0), (1.0, 0.25)]:
            rt.set_params(tonemap_gamma=gamma, tonemap_exposure=expo)
            got8, want8 = rt.read_rgba8(), o.rgba8(expo, gamma)
            assert np.array_equal(got8, want8), (gamma, expo, int((got8 != want8).sum()))
            got16, want16 = rt.read_rgb16(), o.rgb16(expo, gamma)
            assert got16.dtype == np.uint16 and np.array_equal(got16, want16), (gamma, expo)
            n8 += len(np.unique(got8[..., :3])); n16 += len(np.unique(got16))
            rt.upload_overlay(ov)
            assert np.array_equal(rt.read_rgba8(), o.rgba8(expo, gamma, overlay=ov)), ("overlay", gamma, expo)
            rt.upload_overlay(None)
        assert n8 > 300 and n16 > 5000           # real images, not constants
        # the spec's level is round(N x^(1/gamma)) up to the rounding of a threshold: never more than one level from numpy
        rt.set_params(tonemap_gamma=2.2, tonemap_exposure=0.9)
        lin = rt.read_linear()[..., :3]
        ref = np.floor(np.clip((0.9 * np.maximum(lin.astype(np.float64), 0)) ** (1 / 2.2), 0, 1) * 255 + 0.5)
        d = np.abs(rt.read_rgba8()[..., :3].astype(np.int64) - ref.astype(np.int64))
        assert d.max() <= 1 and (d != 0).mean() < 1e-3
    finally:
        rt.close()


def test_overlay_postprocess_is_exact_alpha_compositing(native_lib, dem_small):
    """D12 (renderer_video.py:21-25): an opaque black patch renders 0, a 50 % black patch over 46 renders 23."""
    from moonrtx_amd.renderer import MoonRT
    s = named_scene("S2", 64, 48, spp_per_launch=4)
    rt = MoonRT(s.width, s.height)
    rt.upload_dem(dem_small); rt.apply_scene(s); rt.render(1)
    plain = rt.read_rgba8()
    ov = np.zeros((48, 64, 4), np.uint8)
    ov[:10, :, 3] = 255                                   # opaque black bar
    ov[10:20, :, 3] = 128                                 # 50 % black
    ov[20:30] = (255, 255, 255, 255)                      # opaque white
    ov[30:40] = (200, 100, 50, 77)
    rt.upload_overlay(ov)
    got = rt.read_rgba8()
    a = ov[..., 3:4].astype(np.uint32)
    want = (plain[..., :3].astype(np.uint32) * (255 - a) + ov[..., :3].astype(np.uint32) * a + 127) // 255
    assert np.array_equal(got[..., :3], want.astype(np.uint8)) and (got[..., 3] == 255).all()
    assert (got[:10, :, :3] == 0).all() and (got[20:30, :, :3] == 255).all() and np.array_equal(got[40:], plain[40:])
    assert (46 * (255 - 128) + 127) // 255 == 23           # the reference's own check value
    rt.upload_overlay(None)
    assert np.array_equal(rt.read_rgba8(), plain)
    rt.close()


def test_device_ldem_pipeline_matches_oracle(native_lib):
    """a1 on the device (data_loader.py:166-247): block mean, scale, +1, /max -- bit exact."""
    from moonrtx_amd.renderer import DeviceBuffer, dem_from_ldem
    for d in (1, 2, 3, 4, 8):
        src = synth_np.ldem_source(24 * d, 48 * d, seed=d)
        buf = DeviceBuffer(src.nbytes); buf.upload(src)
        dst, scale = dem_from_ldem(buf, 24, 48, d)
        got = dst.download(np.float32, (24, 48))
        want, wscale = orc.dem_from_ldem(src, d)
        assert_bit_equal(got, want, f"device LDEM pipeline d={d}")
        assert np.float32(scale) == np.float32(wscale)
        assert got.max() == 1.0


def test_synthetic_device_dem_is_lola_like_and_renders(native_lib):
    from moonrtx_amd.renderer import synth_ldem, dem_from_ldem
    h, w = 360, 720
    src = synth_ldem(h, w)
    dst, scale = dem_from_ldem(src, h, w, 1)
    dem = dst.download(np.float32, (h, w))
    raw = src.download(np.int16, (h, w))
    assert raw.min() >= -18200 and raw.max() <= 21600 and raw.std() > 500
    assert dem.max() == 1.0 and dem.min() > 0.985
    check(named_scene("S1", 64, 64, spp_per_launch=4), dem)


def test_errors_are_reported_not_fatal(native_lib):
    from moonrtx_amd.renderer import MoonRT, MoonRTError
    rt = MoonRT(32, 32)
    with pytest.raises(MoonRTError, match="displacement"):
        rt.render(1)
    with pytest.raises(MoonRTError, match="spp_per_launch"):
        rt.set_params(spp_per_launch=3)
    with pytest.raises(MoonRTError):
        rt.set_camera((0, 0, 0), (0, 1, 0), (0, 2, 0), 4.0)
    rt.close()
    with pytest.raises(MoonRTError):
        MoonRT(0, 10)


def test_render_and_gather_async_branch_with_a_stand_in_collective(native_lib, dem_small, monkeypatch):
    """FrameGather.render_and_gather()'s overlapped branch -- parts, slice offsets, async works, unpack -- with `world`
    ranks as threads on one GPU and a stand-in for torch.distributed.gather that does what the collective does
    (rendezvous of all ranks per call, copy of every rank's slice into the root's list).  The real RCCL call pattern
    is covered by test_rccl_async_gather_of_views; a multi-rank RCCL run needs the 8-GPU node."""
    import subprocess, sys, textwrap
    code = textwrap.dedent(r"""
        import sys, threading
        import numpy as np, torch, torch.distributed as dist
        sys.path.insert(0, "tests")
        import synth_np
        from moonrtx_amd import dist as mdist
        from moonrtx_amd.renderer import MoonRT
        from moonrtx_amd.scene import named_scene

        world = int(sys.argv[1]); with_hits = sys.argv[2] == "1"
        lock = threading.Lock()
        calls = {}                                   # call index -> {"bar": Barrier, "src": {rank: tensor}, "dst": list}
        counters = [0] * world
        tls = threading.local()

        class Work:
            def __init__(self, slot, rank): self.slot, self.rank = slot, rank
            def wait(self):
                self.slot["bar"].wait()              # every rank has deposited
                if self.rank == 0:
                    for r, t in self.slot["src"].items():
                        self.slot["dst"][r].copy_(t)
                    torch.cuda.synchronize()
                self.slot["bar2"].wait()

        def fake_gather(tensor, gather_list=None, dst=0, async_op=False):
            rank = tls.rank
            k = counters[rank]; counters[rank] += 1
            with lock:
                slot = calls.setdefault(k, {"bar": threading.Barrier(world), "bar2": threading.Barrier(world), "src": {}, "dst": None})
                slot["src"][rank] = tensor
                if rank == 0: slot["dst"] = gather_list
            w = Work(slot, rank)
            if async_op: return w
            w.wait()

        dist.gather = fake_gather
        dist.get_backend = lambda *a, **k: "nccl"
        dem = synth_np.dem(90, 180, seed=5, craters=10)
        scene = named_scene("S1", 192, 96, spp_per_launch=8)
        ref = MoonRT(scene.width, scene.height, tile=(16, 16)); ref.upload_dem(dem); ref.apply_scene(scene); ref.render(1)
        want, want_h = ref.read_linear(), ref.read_hits(); ref.close()
        out, errs = {}, []
        def run(rank):
            try:
                tls.rank = rank
                rt = MoonRT(scene.width, scene.height, rank=rank, world=world, tile=(16, 16))
                rt.upload_dem(dem); rt.apply_scene(scene)
                g = mdist.FrameGather(rt, torch.device("cuda", 0), with_hits=with_hits)
                assert rt.shard_parts(2) == 2
                assert rt.shard_bytes() == rt.config()["tile_w"] * rt.config()["tile_h"] * (32 if with_hits else 16) * -(-(12 * 6) // world)
                for _ in range(2):                   # twice: buffers and counters are reused
                    rt.reset()
                    st = g.render_and_gather(1, parts=2)
                assert g.last_bytes == rt.shard_bytes_active() < rt.shard_bytes()
                if rank == 0: out["lin"], out["hits"] = rt.read_linear(), rt.read_hits()
                rt.close()
            except Exception as e:
                errs.append((rank, repr(e)))
                for s in list(calls.values()): s["bar"].abort(); s["bar2"].abort()
        th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        [t.start() for t in th]; [t.join(120) for t in th]
        assert not errs, errs
        assert np.array_equal(out["lin"].view(np.uint32), want.view(np.uint32)), "radiance"
        if with_hits:
            assert np.array_equal(out["hits"].view(np.uint32), want_h.view(np.uint32)), "hits"
        else:       # the final linear framebuffer is what travelled: the root's hit buffer holds its own tiles, nothing else
            own = np.zeros((scene.height, scene.width), bool)
            for y in range(0, scene.height, 16):
                for x in range(0, scene.width, 16):
                    own[y:y + 16, x:x + 16] = mdist.tile_owner(x, y, scene.width, (16, 16), world) == 0
            assert np.array_equal(out["hits"][own].view(np.uint32), want_h[own].view(np.uint32)), "own hits"
            assert not out["hits"][~own].any(), "peer hits must not have travelled"
        print("ok", world, len(calls))
    """)
    for world, with_hits in ((2, 0), (3, 1), (4, 0), (8, 0), (8, 1)):
        r = subprocess.run([sys.executable, "-c", code, str(world), str(with_hits)], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "ok" in r.stdout, (world, with_hits, r.stdout[-500:], r.stderr[-2500:])
