"""D6 (set_uint("path_seg_range", 2, 4), moon_renderer.py:583) through the queue-based path stage:
render_kernel<MODE 2> hands every primary terrain hit to the persistent path_kernel, resolve_paths_kernel sums the
samples.  It must equal the in-wave path loop (MRTX_F_INWAVE_PATHS) and the oracle bit for bit -- radiance, hit
records and every spec counter -- for every wave packing, across accumulation blocks, with overlays, an environment
map and the Sun disk."""
import numpy as np
import pytest

import synth_np
from common import STAT_KEYS, assert_bit_equal, render_hip, render_oracle
from moonrtx_amd import _lib
from moonrtx_amd.scene import named_scene

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def rough():
    return synth_np.corrugated_dem(720, 1440)      # steep relief: continuation rays really do hit terrain again


def three_way(scene, dem, color=None, bg=None, blocks=(1,), capsules=None, tile=(32, 32), extra_flags=0):
    lin_o, hits_o, st_o = render_oracle(scene, dem, color, bg, blocks, capsules=capsules)
    out = {}
    for tag, fl in (("queue", 0), ("inwave", _lib.F_INWAVE_PATHS)):
        for count in (_lib.F_COUNT_STATS, 0):      # counting and production instantiations
            lin, hits, st, _ = render_hip(scene, dem, color, bg, blocks, capsules=capsules, tile=tile,
                                          flags=fl | count | extra_flags)
            assert_bit_equal(lin, lin_o, f"{tag} (count={count}) radiance vs oracle")
            assert_bit_equal(hits, hits_o, f"{tag} (count={count}) hits vs oracle")
            if count and len(blocks) == 1:
                assert {k: st[k] for k in STAT_KEYS} == {k: st_o[k] for k in STAT_KEYS}, tag
            out[(tag, count)] = st
    assert out[("queue", 1)]["paths_ms"] > 0.0 and out[("inwave", 1)]["paths_ms"] == 0.0
    return st_o


@pytest.mark.parametrize("spp", [1, 2, 4, 8, 16, 32, 64])
def test_path_queue_every_wave_packing(native_lib, rough, spp):
    s = named_scene("S1", 70, 50, spp_per_launch=spp)
    s.path_seg_min, s.path_seg_max = 2, 4
    st = three_way(s, rough)
    assert st["bounce_rays"] >= st["primary_hits"] > 0


@pytest.mark.parametrize("seg", [(2, 2), (1, 3), (3, 4), (4, 4), (1, 2)])
def test_path_queue_segment_ranges_with_textures(native_lib, rough, seg):
    col = synth_np.colour_map(90, 180)
    bg = np.random.default_rng(11).integers(0, 255, (32, 64, 4), dtype=np.uint8)
    s = named_scene("S1", 96, 72, spp_per_launch=16)
    s.path_seg_min, s.path_seg_max = seg
    st = three_way(s, rough, col, bg)
    assert st["bounce_rays"] > 0 and st["background_fetches"] > 0


def test_path_queue_across_blocks_and_tiles(native_lib, rough):
    """Accumulation over several launches (the running sum is read back) and a tile size that does not divide the frame."""
    s = named_scene("S3", 75, 53, spp_per_launch=8)
    s.path_seg_min, s.path_seg_max = 2, 4
    three_way(s, rough, blocks=(2, 1, 1), tile=(16, 48))
    three_way(s, rough, blocks=(3,), extra_flags=_lib.F_NO_CULL | _lib.F_FORCE_WIDE)


def test_sun_disk_lights_the_moon_through_continuation_rays(native_lib, rough):
    """The flat Sun-disk sphere is visible to continuation rays (moon_renderer.py:109-111: "the stray light the disk
    bounces onto the Moon"; :757-760): a huge disk in front of the night side brightens it, and only through bounces."""
    s = named_scene("S3", 64, 48, spp_per_launch=16)              # crescent: most of the disc is night
    s.sun_pos, s.sun_radius, s.sun_radiance = (0.0, -2500.0, 0.0), 1500.0, 2.0     # behind the camera, facing the night side
    s.path_seg_min, s.path_seg_max = 2, 4
    st = three_way(s, rough)
    assert st["bounce_sun_hits"] > 0
    lit = render_oracle(s, rough)[0]
    s.path_seg_min, s.path_seg_max = 1, 1
    dark = render_oracle(s, rough)[0]
    s.path_seg_min, s.path_seg_max = 2, 4
    s.sun_radius = 0.01                                           # parked (moon_renderer.py:115)
    parked, _, stp = render_oracle(s, rough)
    assert stp["bounce_sun_hits"] == 0
    night = dark[..., :3].sum(-1) == 0.0
    assert night.sum() > 200
    assert lit[..., :3][night].mean() > 20 * max(parked[..., :3][night].mean(), 1e-6)


def test_path_queue_with_overlay_tubes(native_lib, rough):
    from moonrtx_amd import overlays
    s = named_scene("S1", 120, 90, spp_per_launch=8)
    s.path_seg_min, s.path_seg_max = 2, 3
    pos, edges, r, c = overlays.graticule(rotation=s.rotation, tube=0.02)
    caps = overlays.graph_to_capsules(pos, edges, r, c)
    three_way(s, rough, capsules=caps)


def test_hand_over_buffers_are_bounded_by_rendering_in_sub_parts(native_lib, rough, monkeypatch):
    """A frame whose hand-over records would exceed the budget (MOONRT_PATH_MAX_GB; 64 GB by default, e.g. a cfg4 frame with
    every pixel on the Moon needs 129 GB) is rendered in sub-parts of its tile list through the same buffers -- same frame."""
    s = named_scene("S1", 160, 128, spp_per_launch=64)
    s.path_seg_min, s.path_seg_max = 2, 4
    s.vfov_deg = 2.0                                     # every tile on the Moon: 20 tiles x 1024 wave-jobs
    lin_o, hits_o, st_o = render_oracle(s, rough)
    monkeypatch.setenv("MOONRT_PATH_MAX_GB", "0.001")    # -> 4096 wave-jobs per sub-part: five sub-parts
    for count in (_lib.F_COUNT_STATS, 0):
        lin, hits, st, _ = render_hip(s, rough, flags=count)
        assert_bit_equal(lin, lin_o, "sub-parts radiance"); assert_bit_equal(hits, hits_o, "sub-parts hits")
        if count:
            assert {k: st[k] for k in STAT_KEYS} == {k: st_o[k] for k in STAT_KEYS}
        assert st["launches"] >= 3 * 4


def test_small_launches_keep_their_paths_in_the_wave(native_lib, rough, monkeypatch):
    """Below MOONRT_PATH_QUEUE_MIN samples (8 M by default) a launch runs ONE kernel with the paths inside the wave; above it
    the three-kernel queue.  Same frame either way."""
    s = named_scene("S1", 96, 64, spp_per_launch=16)
    s.path_seg_min, s.path_seg_max = 2, 4
    lin_o, hits_o, _ = render_oracle(s, rough)
    monkeypatch.delenv("MOONRT_PATH_QUEUE_MIN", raising=False)
    lin, hits, st, _ = render_hip(s, rough, flags=0)
    assert st["launches"] == 1 and st["paths_ms"] == 0.0
    assert_bit_equal(lin, lin_o, "automatic in-wave radiance"); assert_bit_equal(hits, hits_o, "automatic in-wave hits")
    monkeypatch.setenv("MOONRT_PATH_QUEUE_MIN", "1000")
    lin, hits, st, _ = render_hip(s, rough, flags=0)
    assert st["launches"] == 3 and st["paths_ms"] > 0.0
    assert_bit_equal(lin, lin_o, "queue radiance"); assert_bit_equal(hits, hits_o, "queue hits")


def test_sky_tiles_get_their_own_launch_when_an_environment_is_bound(native_lib, rough):
    """moon_renderer.py:604-607 binds a star map: nothing can be culled any more, but the tiles that can only see the sky are
    rendered by render_kernel<MODE 3> (environment texel only) and stay out of the path queue -- same frame as the oracle."""
    bg = np.random.default_rng(5).integers(0, 255, (48, 96, 4), dtype=np.uint8)
    bg[::3] = 0                                          # black texels: an escaping path adds exactly nothing there
    s = named_scene("S1", 160, 128, spp_per_launch=16)
    s.vfov_deg = 12.0                                    # the disc covers the middle tiles only
    for seg, launches in (((2, 4), 4), ((1, 1), 2)):
        s.path_seg_min, s.path_seg_max = seg
        lin_o, hits_o, st_o = render_oracle(s, rough, None, bg)
        for count in (_lib.F_COUNT_STATS, 0):
            lin, hits, st, _ = render_hip(s, rough, None, bg, flags=count)
            assert_bit_equal(lin, lin_o, f"sky split {seg} radiance"); assert_bit_equal(hits, hits_o, f"sky split {seg} hits")
            if count:
                assert {k: st[k] for k in STAT_KEYS} == {k: st_o[k] for k in STAT_KEYS}
            assert st["launches"] == launches, st["launches"]
        assert st_o["background_fetches"] > st_o["primary_hits"] > 0


def test_two_ranks_with_environment_and_paths(native_lib, rough):
    """What FrameGather drives on two GPUs when an environment map is bound (full gather layout, no parts), here on two
    contexts of one GPU: (2,4) paths behind the queue, the sky tiles at the end of every rank's list rendered by their own
    launch -- the reassembled frame equals the oracle's."""
    from moonrtx_amd.renderer import MoonRT, DeviceBuffer
    bg = np.random.default_rng(9).integers(0, 255, (40, 80, 4), dtype=np.uint8)
    s = named_scene("S1", 192, 128, spp_per_launch=16)
    s.vfov_deg = 9.0
    s.path_seg_min, s.path_seg_max = 2, 4
    lin_o, hits_o, _ = render_oracle(s, rough, None, bg)
    world = 2
    rts = [MoonRT(s.width, s.height, rank=r, world=world, tile=(16, 16)) for r in range(world)]
    bufs = [DeviceBuffer(rt.shard_bytes()) for rt in rts]
    try:
        for rt, buf in zip(rts, bufs):
            rt.upload_dem(rough); rt.upload_background(bg); rt.apply_scene(s); rt.set_params(flags=0)
            assert rt.shard_parts(2) == 1                 # an environment map needs every tile: the full layout, no parts
            rt.reset()
            st = rt.render(1)
            assert st["launches"] == 4                    # render + paths + resolve for the Moon's tiles, one for the sky
            rt.pack_shard(buf.ptr)
        rts[0].unpack_all([buf.ptr for buf in bufs])
        assert_bit_equal(rts[0].read_linear(), lin_o, "2 ranks, environment, paths: radiance")
        assert_bit_equal(rts[0].read_hits(), hits_o, "2 ranks, environment, paths: hits")
    finally:
        for rt in rts:
            rt.close()


def test_no_memory_for_the_records_falls_back_to_in_wave_paths(native_lib, rough, monkeypatch, capfd):
    """Round-2 advisor finding: a render call must not fail where an identical result exists.  When the hand-over buffers
    cannot be allocated (test hook MOONRT_TEST_PATH_NOMEM) the frame is rendered with its paths inside the render wave:
    one launch, the oracle's frame, one line on stderr."""
    s = named_scene("S1", 96, 64, spp_per_launch=16)
    s.path_seg_min, s.path_seg_max = 2, 4
    lin_o, hits_o, st_o = render_oracle(s, rough)
    monkeypatch.setenv("MOONRT_TEST_PATH_NOMEM", "1")
    lin, hits, st, _ = render_hip(s, rough, flags=_lib.F_COUNT_STATS, blocks=(1, 1))
    lin_o2, hits_o2, st_o2 = render_oracle(s, rough, blocks=(1, 1))
    assert_bit_equal(lin, lin_o2, "fallback radiance"); assert_bit_equal(hits, hits_o2, "fallback hits")
    assert st["launches"] == 2 and st["paths_ms"] == 0.0
    assert capfd.readouterr().err.count("keep their paths inside the render wave") == 1      # said once per context


def test_path_stage_launch_parameters_are_validated(native_lib, rough, monkeypatch):
    """Round-2 advisor finding: tuning values that would silently drop paths are not accepted -- MOONRT_PATH_GRP above 5 (a group
    must fit the 64 lanes that hold its chunk counts) and MOONRT_PATH_WAVES below 8 x MOONRT_PATH_NSUB (a work counter without
    a consumer) are ignored by mrtx_create, and the frame is the oracle's."""
    s = named_scene("S1", 96, 64, spp_per_launch=64)
    s.path_seg_min, s.path_seg_max = 2, 4
    lin_o, hits_o, _ = render_oracle(s, rough)
    for env in ({"MOONRT_PATH_GRP": "9"}, {"MOONRT_PATH_WAVES": "8"}, {"MOONRT_PATH_WAVES": "16", "MOONRT_PATH_NSUB": "4"},
                {"MOONRT_PATH_GRP": "5", "MOONRT_PATH_WAVES": "64", "MOONRT_PATH_NSUB": "8"}):
        for k in ("MOONRT_PATH_GRP", "MOONRT_PATH_WAVES", "MOONRT_PATH_NSUB"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        lin, hits, st, _ = render_hip(s, rough, flags=0)
        assert_bit_equal(lin, lin_o, f"radiance with {env}"); assert_bit_equal(hits, hits_o, f"hits with {env}")
        assert st["launches"] == 3
