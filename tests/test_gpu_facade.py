"""The PlotOptiX-named facade on the real HIP backend: driven with moon_renderer.py's call sequence it must
produce the oracle's frame for the same scene."""
import numpy as np
import pytest

import synth_np
from common import assert_bit_equal, render_oracle
from test_facade_cpu import drive_like_init_renderer
from moonrtx_amd import scene as sc
from moonrtx_amd.tkoptix import TkOptiX

pytestmark = pytest.mark.gpu


def test_facade_frame_equals_oracle(native_lib):
    W, H = 96, 64
    dem = synth_np.dem(180, 360, seed=5, craters=30)
    col = synth_np.colour_map(90, 180)
    launches = []
    rt = TkOptiX(width=W, height=H, on_launch_finished=lambda r: launches.append(1))
    drive_like_init_renderer(rt, dem, col, gamma=2.2, brightness=80)
    rt.set_param(max_accumulation_frames=16)
    # update_view (moon_renderer.py:840-860) for scene S1
    s = sc.named_scene("S1", W, H, spp_per_launch=16)
    with rt._padlock:
        rt.update_camera("cam1", eye=list(s.eye))
        rt.update_data("moon", u=s.u, v=s.v)
        rt.update_data("sun_disk", pos=[list(s.sun_pos)], r=s.sun_radius)
        rt.update_light("sun", pos=list(s.light_pos), radius=s.light_radius)
    img = rt.render_cycle()
    lin = rt._rt.read_linear()
    hits = rt._rt.read_hits()
    s.vfov_deg = 4.2422
    s.path_seg_min, s.path_seg_max = 2, 4          # what init_renderer sets (moon_renderer.py:583)
    lin_o, hits_o, _ = render_oracle(s, dem, col)
    assert_bit_equal(lin, lin_o, "facade frame vs oracle")
    assert_bit_equal(hits, hits_o, "facade hits vs oracle")
    assert img.shape == (H, W, 4) and img[..., :3].max() > 50
    # picking path: renderer_navigation.py:452-492 on the facade's hit buffer
    hx, hy, hz, hd = rt._get_hit_at(W // 2 + 10, H // 2)
    assert hd > 280 and 9.8 < np.linalg.norm([hx, hy, hz]) <= 10.0
    lat, lon = sc.selenographic((hx, hy, hz), s.rotation)
    assert -90 <= lat <= 90 and -180 <= lon <= 180
    rt.close()


def test_facade_render_thread_on_gpu(native_lib):
    import threading
    dem = synth_np.dem(90, 180, seed=1, craters=5)
    done = threading.Event()
    rt = TkOptiX(width=48, height=32)
    drive_like_init_renderer(rt, dem, synth_np.colour_map(45, 90))
    rt.set_accum_done_cb(lambda r: done.set())
    rt.start()
    assert done.wait(30.0)
    assert rt.get_image()[..., :3].max() > 0
    rt.close()


def test_rendered_phase_matches_the_ephemeris(native_lib):
    """Date + observer -> ephemeris -> scene -> HIP render of a smooth sphere: the lit fraction of the disc equals the
    illuminated fraction k = (1 + cos i) / 2, and the lit side points along the bright-limb angle (measured from 'up'
    towards celestial east, which is to the LEFT in the view)."""
    import math
    from datetime import datetime, timedelta, timezone
    from moonrtx_amd import ephemeris as E
    from moonrtx_amd.renderer import MoonRT
    E.init(E.Observer(52.2, 21.0, 100))
    dem = np.ones((16, 32), np.float32)
    W = H = 256
    for days in (3.0, 7.4, 11.0, 18.5, 24.0):
        dt = datetime(2025, 3, 29, 11, 0, tzinfo=timezone.utc) + timedelta(days=days)      # new moon 2025-03-29 10:58 UTC
        for mode in (True, False):
            e = E.calculate_moon_ephemeris(dt, mode)
            s = E.scene_from_ephemeris(e, W, H, spp_per_launch=16)
            rt = MoonRT(W, H)
            rt.upload_dem(dem); rt.apply_scene(s); rt.render(1)
            lin = rt.read_linear(); rt.close()
            disc = lin[..., 3] > 0.5
            lit = disc & (lin[..., 0] > 0.0)                  # the Sun's 0.27 deg radius softens the terminator by ~0.2 % of the disc
            k = (1.0 + math.cos(math.radians(e.phase_angle))) / 2.0
            assert abs(lit.sum() / disc.sum() - k) < 0.02, (days, mode, lit.sum() / disc.sum(), k)
            ys, xs = np.nonzero(lit)
            cx, cy = xs.mean() - (W - 1) / 2.0, (H - 1) / 2.0 - ys.mean()                 # image x right, y up
            got = math.degrees(math.atan2(-cx, cy))                                      # from up towards the left
            if 0.05 < k < 0.95:
                assert abs(E.wrap_signed_degrees(got - e.bright_limb_angle)) < 3.0, (days, mode, got, e.bright_limb_angle)
            # the disc fills 0.9 of the frame height at the reference distance and scales with the apparent radius
            # (moon_renderer.py:522-544: the camera distance follows the topocentric distance)
            scale = math.asin(1737.4 / e.distance) / math.asin(1737.4 / 384400.0)
            assert abs(disc.sum() / (math.pi * (0.45 * H * scale) ** 2) - 1.0) < 0.02, (days, disc.sum(), scale)


def test_video_export_on_gpu_holds_the_rendered_frames(native_lib, tmp_path):
    """renderer_video.py:219-340 on the HIP backend: three converged cycles with the Sun moving in between, each captured into the
    Motion-JPEG AVI before the accum-done callback moves the scene on; the decoded frames are the cycles' RGBA8 images."""
    import io
    import threading
    from PIL import Image
    from moonrtx_amd.video import read_avi_frames
    W, H = 128, 96
    dem = synth_np.dem(180, 360, seed=2, craters=20)
    rt = TkOptiX(width=W, height=H)
    drive_like_init_renderer(rt, dem, synth_np.colour_map(90, 180))
    s = sc.named_scene("S1", W, H, spp_per_launch=64)
    with rt._padlock:
        rt.update_camera("cam1", eye=list(s.eye))
        rt.update_data("moon", u=s.u, v=s.v)
        rt.update_light("sun", pos=list(s.light_pos), radius=s.light_radius)
    rt.encoder_create(fps=30, bitrate=60)
    rt.encoder_start(str(tmp_path / "phase.avi"), 3)
    shots, finished = [], threading.Event()
    lp = np.array(s.light_pos, float)

    def accum_done(r):
        shots.append(r.get_image().copy())
        if len(shots) < 3:
            k = len(shots)
            c, sn = np.cos(0.6 * k), np.sin(0.6 * k)
            r.update_light("sun", pos=[c * lp[0] - sn * lp[1], sn * lp[0] + c * lp[1], lp[2]])
            r.refresh_scene()
        else:
            finished.set()

    rt.set_accum_done_cb(accum_done)
    rt.start()
    assert finished.wait(60.0)
    assert not rt.encoder_is_open() and rt.encoded_frames() == 3
    rt.close()
    info, frames = read_avi_frames(str(tmp_path / "phase.avi"))
    assert (info["width"], info["height"], info["total_frames"]) == (W, H, 3)
    means = []
    for shot, jp in zip(shots, frames):
        dec = np.asarray(Image.open(io.BytesIO(jp)).convert("RGB"), np.float32)
        ref = shot[..., :3].astype(np.float32)
        assert ref.max() > 50
        assert 10 * np.log10(255.0 ** 2 / max(1e-9, np.mean((dec - ref) ** 2))) > 32.0
        means.append(ref.mean())
    assert len({round(m, 2) for m in means}) == 3          # the Sun moved: three different frames
