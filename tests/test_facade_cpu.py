"""The PlotOptiX-named facade driven the way moon_renderer.py drives `self.rt`, on a recording backend
(no GPU): parameter mapping, scene plumbing, render-thread callback contract."""
import threading
import time

import numpy as np
import pytest

from moonrtx_amd.tkoptix import TkOptiX
from moonrtx_amd.materials import m_diffuse, m_flat
from moonrtx_amd import scene as sc


class RecordingBackend:
    def __init__(self, w, h):
        self.width, self.height, self.rank, self.world = w, h, 0, 1
        self.calls = []
        self.blocks = 0
        self.closed = False

    def __getattr__(self, name):
        def rec(*a, **k):
            self.calls.append((name, a, k))
            if name == "render":
                self.blocks += a[0] if a else 1
                return {"kernel_ms": 0.0}
            if name == "reset":
                self.blocks = 0
            if name == "read_rgba8":
                return np.full((self.height, self.width, 4), min(255, self.blocks), np.uint8)
            if name == "read_hits":
                h = np.zeros((self.height, self.width, 4), np.float32); h[2, 3] = (1, 2, 3, 290.5); return h
            if name == "read_hit":
                return (1.0, 2.0, 3.0, 290.5) if tuple(a[:2]) == (3, 2) else (0.0, 0.0, 0.0, 0.0)
            if name == "read_linear":
                return np.zeros((self.height, self.width, 4), np.float32)
            if name == "close":
                self.closed = True
        return rec

    def last(self, name):
        return [c for c in self.calls if c[0] == name][-1]


def drive_like_init_renderer(rt, elevation, color, gamma=2.2, brightness=80):
    """The call sequence of MoonRenderer.init_renderer (moon_renderer.py:570-650)."""
    rt.set_param(min_accumulation_step=1, max_accumulation_frames=64)
    rt.set_uint("path_seg_range", 2, 4)
    rt.set_float("scene_epsilon", 1.0e-4)
    rt.set_float("marching_step", 5.0e-3)
    rt.set_float("marching_step_eps", 3.0e-4)
    rt.set_ambient(0)
    rt.set_float("tonemap_exposure", 0.9)
    rt.set_float("tonemap_gamma", gamma)
    rt.add_postproc("Gamma")
    rt.set_background(0)
    rt.set_texture_2d("moon_color", color)
    mat = m_diffuse.copy()
    mat["ColorTextures"] = ["moon_color"]
    rt.update_material("diffuse", mat)
    rt.set_data("moon", geom="ParticleSetTextured", geom_attr="DisplacedSurface", pos=[0, 0, 0], u=[0, 0, 1],
                v=[0, -1, 0], r=10.0)
    rt.set_displacement("moon", elevation, refresh=False)
    rt.setup_camera("cam1", cam_type="Pinhole", eye=[0, -300, 0], target=[0, 0, 0], up=[0, 0, 1], fov=4.2422,
                    aperture_radius=0.01, aperture_fract=0.2, focal_scale=0.7)
    rt.setup_light("sun", color=brightness * sc.SUN_BRIGHTNESS_SCALE, radius=100, in_geometry=False)
    flat = dict(m_flat); flat["OcclusionProgram"] = "x"; flat["VarFloat4"] = {"x": [1, 1, 1, 0]}
    rt.setup_material("flat", flat)
    rt.set_data("sun_disk", geom="ParticleSet", mat="flat", pos=[[0.0, 3100.0, 0.0]], r=0.01, c=2.0)


def make():
    be = RecordingBackend(16, 8)
    events = []
    rt = TkOptiX(width=16, height=8, on_launch_finished=lambda r: events.append(("launch", threading.current_thread().name)),
                 backend=be)
    return rt, be, events


def test_init_sequence_maps_onto_the_c_abi_surface():
    rt, be, _ = make()
    elev = np.ones((4, 8), np.float32)
    col = np.zeros((4, 8, 4), np.uint8)
    drive_like_init_renderer(rt, elev, col)
    assert be.last("upload_dem")[1][0].shape == (4, 8)
    assert be.last("upload_color")[1][0].shape == (4, 8, 4)
    assert be.last("upload_background")[1] == (None,)
    name, a, _ = be.last("set_moon_frame")
    assert np.allclose(a[0], 0) and a[1] == 10.0 and np.allclose(a[2], (0, 0, 1)) and np.allclose(a[3], (0, -1, 0))
    name, a, _ = be.last("set_camera")
    assert np.allclose(a[0], (0, -300, 0)) and abs(a[3] - 4.2422) < 1e-12
    name, a, _ = be.last("set_light")
    assert a[1] == 100 and abs(a[2] - 80 * 460.5316) < 1e-6
    name, a, _ = be.last("set_sun_disk")
    assert np.allclose(a[0], (0, 3100, 0)) and a[1] == 0.01 and a[2] == 2.0
    cam = rt.get_camera("cam1")
    assert cam["Eye"] == [0, -300, 0] and cam["Target"] == [0, 0, 0] and cam["Up"] == [0, 0, 1]
    assert rt._optix.get_camera_fov(0) == pytest.approx(4.2422)
    rt.close()
    assert be.closed


def test_update_view_sequence_and_cycle_restart():
    """moon_renderer.py:852-871: camera move + update_data + update_light under the padlock, then refresh."""
    rt, be, _ = make()
    drive_like_init_renderer(rt, np.ones((4, 8), np.float32), np.zeros((4, 8, 4), np.uint8))
    img = rt.render_cycle()
    assert img[0, 0, 0] == 1                     # 64 frames == ONE 64-spp launch
    assert be.last("set_params")[2]["spp_per_launch"] == 64
    R = sc.libration_rotation(3.0, -5.0)
    with rt._padlock:
        with rt._padlock:                        # re-entrant
            cam = rt.get_camera("cam1")
            rt.update_camera("cam1", eye=(np.array(cam["Eye"]) * 1.01).tolist())
        rt.update_data("moon", u=R[:, 2], v=-R[:, 1])
        rt.update_data("sun_disk", pos=[[10.0, 2800.0, 5.0]], r=33.0)
        rt.update_light("sun", pos=sc.light_position(77, -70), radius=99.8)
    assert np.allclose(be.last("set_moon_frame")[1][2], R[:, 2])
    assert be.last("set_sun_disk")[1][1] == 33.0
    assert np.allclose(be.last("set_light")[1][0], sc.light_position(77, -70)) and be.last("set_light")[1][1] == 99.8
    n_reset = sum(1 for c in be.calls if c[0] == "reset")
    rt.render_cycle()
    assert sum(1 for c in be.calls if c[0] == "reset") == n_reset + 1
    # preview switch (moon_renderer.py:475): one frame per cycle -> a 1-spp launch
    rt.set_param(max_accumulation_frames=1)
    rt.render_cycle()
    assert be.last("set_params")[2]["spp_per_launch"] == 1
    rt._optix.set_camera_fov(2.0)
    assert be.last("set_camera")[1][3] == 2.0
    assert rt._get_hit_at(3, 2) == (1.0, 2.0, 3.0, 290.5) and rt._get_hit_at(99, 99)[3] <= 0
    # the hit buffer is never pulled back wholesale: one texel per query (moon_renderer.py:1138 asks once per mouse event)
    assert not [c for c in be.calls if c[0] == "read_hits"] and be.last("read_hit")[1] == (3, 2)
    rt.close()


def test_render_thread_callbacks_and_padlock_contract():
    """on_launch_finished after every launch from the render thread; accum-done callback runs with the
    padlock held and may edit the scene re-entrantly (renderer_video.py:276-318)."""
    rt, be, events = make()
    drive_like_init_renderer(rt, np.ones((4, 8), np.float32), np.zeros((4, 8, 4), np.uint8))
    rt.set_param(max_accumulation_frames=128)    # 2 launches of 64 per cycle
    done = threading.Event()
    seen = {"cycles": 0, "owned": None}

    def accum_done(r):
        seen["cycles"] += 1
        seen["owned"] = r._padlock._is_owned()
        if seen["cycles"] < 3:
            r.update_light("sun", radius=100.0 + seen["cycles"])   # re-entrant edit, as update_view does
            r.refresh_scene()
        else:
            r.set_accum_done_cb(None)
            done.set()

    rt.set_accum_done_cb(accum_done)
    rt.start()
    assert rt._is_started
    assert done.wait(10.0), "render thread never completed three cycles"
    time.sleep(0.1)
    assert seen["owned"] is True
    launches = [e for e in events if e[0] == "launch"]
    assert len(launches) >= 6 and all(t == "moonrt-render" for _, t in launches)
    n = len(events)
    time.sleep(0.4)
    assert len(events) == n, "the loop must idle after a converged cycle until refresh_scene()"
    rt.refresh_scene()
    time.sleep(0.6)
    assert len(events) >= n + 2
    rt.close()
    assert not rt._is_started


def test_unsupported_pieces_are_explicit():
    rt, be, _ = make()
    # overlay graphs (renderer_labels.py:295-300, :324-325, :367-373): flattened into capsules; radius 0 hides
    pos = np.array([[0, -10.25, 0], [1, -10.2, 0], [1, -10.2, 1], [5, 5, 5]], float)
    rt.set_graph("grid", pos=pos, edges=np.array([[0, 1], [1, 2]], np.int32), r=0.006, c=[0.5, 0.5, 0.5], mat="grid_material")
    caps = be.last("set_capsules")[1][0]
    assert caps.shape == (2, 12) and np.allclose(caps[0, :4], [0, -10.25, 0, 0.006]) and np.allclose(caps[1, 8:11], 0.5)
    rt.set_graph("labels", pos=pos, edges=np.array([[0, 1], [2, 3]], np.int32), r=np.array([0.008, 0.008, 0.0, 0.0], np.float32),
                 c=[1.0, 0.9, 0.3], mat="spot_label_material")
    assert be.last("set_capsules")[1][0].shape == (3, 12)            # the zero-radius edge is dropped
    rt.update_graph("grid", r=0.0)                                   # hide (show_moon_grid(False))
    assert be.last("set_capsules")[1][0].shape == (1, 12)
    rt.update_graph("grid", pos=pos * 2.0, r=0.006)
    assert np.allclose(be.last("set_capsules")[1][0][0, :3], [0, -20.5, 0])
    rt.delete_geometry("labels"); rt.delete_geometry("grid")
    assert len(be.last("set_capsules")[1][0]) == 0
    assert rt.encoder_is_open() is False
    with pytest.raises(RuntimeError):
        rt.encoder_start("x.avi", 3)                                 # encoder_create first
    with pytest.raises(ValueError):
        rt.set_param(bogus=1)
    with pytest.raises(ValueError):
        rt.set_texture_2d("t", np.zeros((4, 4, 3), np.uint8))
    rt.close()


def test_reference_grid_graphs_become_capsules():
    """The graphs moon_grid.create_moon_grid() + merge_segments_to_graph() produce in the reference (golden:
    tests/golden/moon_grid_graphs.npz, made by importing moonrtx.moon_grid) through set_graph / update_graph the way
    renderer_labels.py:291-300, :209, :324-325 drive them."""
    import json, os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(here, "moon_grid_graphs.npz"))
    meta = json.load(open(os.path.join(here, "moon_grid_graphs.json")))
    assert list(g["lines_pos"].shape) == meta["lines"]["pos_shape"] == [3300, 3]
    assert list(g["lines_edges"].shape) == meta["lines"]["edges_shape"] == [3267, 2]
    assert list(g["labels_pos"].shape) == meta["labels"]["pos_shape"] == [1266, 3]
    assert list(g["labels_edges"].shape) == meta["labels"]["edges_shape"] == [633, 2]
    assert int(g["lines_edges"].sum()) == meta["lines"]["edge_sum"] and int(g["labels_edges"].sum()) == meta["labels"]["edge_sum"]
    assert np.abs(g["lines_pos"]).sum() == pytest.approx(meta["lines"]["pos_abs_sum"], rel=1e-6)
    # grid lines sit on the sphere of radius 10 * 1.02 (moon_grid.py:689-694), outside the bounding sphere
    assert np.allclose(np.linalg.norm(g["lines_pos"], axis=1), 10.2, atol=1e-4)
    assert np.linalg.norm(g["labels_pos"], axis=1).min() > 10.0

    rt, be, _ = make()
    rt.set_graph("grid_lines", pos=g["lines_pos"], edges=g["lines_edges"], r=meta["radii"]["grid_line"], c=meta["colour"],
                 mat="grid_material")
    caps = be.last("set_capsules")[1][0]
    assert caps.shape == (3267, 12)
    assert np.array_equal(caps[:, 0:3], g["lines_pos"][g["lines_edges"][:, 0]])
    assert np.array_equal(caps[:, 4:7], g["lines_pos"][g["lines_edges"][:, 1]])
    assert np.allclose(caps[:, 3], 0.006) and np.allclose(caps[:, 8:11], 0.5)
    rt.set_graph("grid_labels", pos=g["labels_pos"], edges=g["labels_edges"], r=meta["radii"]["grid_label"], c=meta["colour"],
                 mat="grid_material")
    assert be.last("set_capsules")[1][0].shape == (3267 + 633, 12)
    R = sc.libration_rotation(4.0, -6.0)                                  # update_view: labels follow the libration
    rt.update_graph("grid_labels", pos=g["labels_pos"] @ R.T)
    caps = be.last("set_capsules")[1][0]
    assert caps.shape == (3900, 12) and np.allclose(np.linalg.norm(caps[3267:, 0:3], axis=1), np.linalg.norm(g["labels_pos"][g["labels_edges"][:, 0]], axis=1), atol=1e-4)
    rt.update_graph("grid_lines", r=0.0); rt.update_graph("grid_labels", r=0.0)      # show_moon_grid(False)
    assert len(be.last("set_capsules")[1][0]) == 0
    rt.close()


def test_save_image_16_bits_is_really_16_bits(tmp_path):
    """renderer_dialogs.py:1222-1224: ".tiff" is saved with bps="Bps16" -- 16 bits per sample on disk, never a silent 8."""
    import struct
    from PIL import Image
    be = RecordingBackend(16, 8)
    want = np.zeros((8, 16, 3), np.uint16)          # the backend's 16-bit frame (mrtx_read_rgb16; values checked on the GPU
    want[..., 0] = np.linspace(0, 65535, 16).astype(np.uint16)[None, :]; want[..., 1] = 0x1234; want[..., 2] = 7   # against the oracle)
    be.read_rgb16 = lambda: want
    rt = TkOptiX(width=16, height=8, backend=be)
    rt.set_float("tonemap_exposure", 0.9); rt.set_float("tonemap_gamma", 2.2)
    p = tmp_path / "frame.tiff"
    rt.save_image(str(p), bps="Bps16")
    raw = p.read_bytes()
    assert raw[:4] == b"II*\x00"
    (n_tags,) = struct.unpack_from("<H", raw, 8)
    tags = {struct.unpack_from("<H", raw, 10 + 12 * i)[0]: struct.unpack_from("<HHII", raw, 10 + 12 * i) for i in range(n_tags)}
    assert tags[256][3] == 16 and tags[257][3] == 8 and tags[277][3] & 0xFFFF == 3
    assert struct.unpack_from("<HHH", raw, tags[258][3]) == (16, 16, 16)
    got = np.frombuffer(raw, "<u2", count=8 * 16 * 3, offset=tags[273][3]).reshape(8, 16, 3)
    assert np.array_equal(got, want) and len(np.unique(got[..., 0])) == 16          # 16 distinct levels: not 8-bit data
    assert be.last("set_params")[2]["tonemap_gamma"] == 2.2                          # the post-process parameters were pushed first
    assert Image.open(str(p)).size == (16, 8)
    q = tmp_path / "frame.png"
    rt.save_image(str(q), bps="Bps16")
    assert q.read_bytes()[24] == 16                                                  # IHDR bit depth
    with pytest.raises(ValueError):
        rt.save_image(str(tmp_path / "frame.jpg"), bps="Bps16")
    rt.close()


def test_one_callback_per_cycle_by_default_and_per_frame_when_progressive():
    """The documented deviation at the boundary (set_param's docstring): the reference's
    `min_accumulation_step=1, max_accumulation_frames=64` (moon_renderer.py:578) is ONE launch of 64 samples and ONE
    on_launch_finished here; TkOptiX(progressive=True) gives PlotOptiX's 64 launches of one frame (renderer_status.py:239)."""
    for progressive, launches, spp in ((False, 1, 64), (True, 64, 1)):
        be = RecordingBackend(16, 8)
        fired = []
        rt = TkOptiX(width=16, height=8, on_launch_finished=lambda r: fired.append(1), backend=be, progressive=progressive)
        drive_like_init_renderer(rt, np.ones((4, 8), np.float32), np.zeros((4, 8, 4), np.uint8))
        rt.render_cycle()
        renders = [c for c in be.calls if c[0] == "render"]
        assert len(renders) == launches and len(fired) == launches, (progressive, len(renders), len(fired))
        assert be.last("set_params")[2]["spp_per_launch"] == spp and be.last("set_params")[2]["max_spp"] == 64
        rt.set_param(min_accumulation_step=8)           # progressive: 8 launches of 8 frames; default: still one launch
        be.calls.clear(); fired.clear()
        rt.render_cycle()
        assert len([c for c in be.calls if c[0] == "render"]) == (8 if progressive else 1) == len(fired)
        rt.close()


def test_save_image_16_bits_after_changing_the_cycle_length(tmp_path):
    """render_cycle -> set_param(max_accumulation_frames=...) -> save_image(Bps16) (the headless export sequence): the save pushes
    exposure and gamma only -- re-pushing the cycle plan would ask the library to change spp_per_launch inside an accumulation
    cycle (MRTX_E_STATE; round-3 advisor finding)."""
    class Backend(RecordingBackend):
        def __getattr__(self, name):
            inner = RecordingBackend.__getattr__(self, name)

            def rec(*a, **k):
                if name == "set_params" and "spp_per_launch" in k and self.blocks and k["spp_per_launch"] != self.spp:
                    raise RuntimeError("spp_per_launch cannot change inside an accumulation cycle; reset first")
                if name == "set_params" and "spp_per_launch" in k:
                    self.spp = k["spp_per_launch"]
                if name == "read_rgb16":
                    self.calls.append((name, a, k))
                    return np.full((self.height, self.width, 3), 4660, np.uint16)
                return inner(*a, **k)
            return rec

    be = Backend(16, 8)
    be.spp = None
    rt = TkOptiX(width=16, height=8, backend=be)
    drive_like_init_renderer(rt, np.ones((4, 8), np.float32), np.zeros((4, 8, 4), np.uint8))
    rt.render_cycle()                                   # 64 samples accumulated
    rt.set_param(max_accumulation_frames=1)             # the preview setting, no new cycle rendered yet
    rt.set_float("tonemap_gamma", 1.8)
    out = tmp_path / "frame.tiff"
    rt.save_image(str(out), bps="Bps16")
    assert out.stat().st_size > 16 * 8 * 6
    pushed = be.last("set_params")[2]
    assert pushed == {"tonemap_exposure": 0.9, "tonemap_gamma": 1.8}
    rt.close()


def test_video_export_drives_the_encoder_like_renderer_video(tmp_path):
    """renderer_video.py:219-340: encoder_create, encoder_start(filename, n), one frame per finished accumulation cycle, the
    accum-done callback moving the scene on, the encoder closing by itself at the frame limit."""
    from moonrtx_amd.video import read_avi_frames
    rt, be, _ = make()
    drive_like_init_renderer(rt, np.zeros((8, 16), np.float32), np.zeros((4, 8, 4), np.uint8))
    rt.encoder_create(fps=25, bitrate=16)
    name = str(tmp_path / "moon.mp4")
    rt.encoder_start(name, 3)
    assert rt.encoder_is_open() and rt.encoder_file == name + ".avi" and rt.encoding_frames() == 3
    with pytest.raises(RuntimeError):
        rt.encoder_start(name, 3)                                    # already running
    seen = []

    def accum_done(r):
        seen.append(r.encoded_frames())                              # the frame of this cycle is already captured
        if len(seen) < 5:
            r.update_light("sun", pos=[100.0 * len(seen), 0.0, 0.0])
            r.refresh_scene()

    rt.set_accum_done_cb(accum_done)
    rt.start()
    t0 = time.time()
    while len(seen) < 5 and time.time() - t0 < 20.0:
        time.sleep(0.01)
    rt.set_accum_done_cb(None)
    assert seen == [1, 2, 3, 3, 3]                                   # closed by itself after three frames
    assert not rt.encoder_is_open() and rt.encoded_frames() == 3
    rt.encoder_stop()                                                # harmless after the automatic close (renderer_video.py:339-340)
    rt.close()
    info, frames = read_avi_frames(name + ".avi")
    assert (info["width"], info["height"], info["total_frames"], info["length"]) == (16, 8, 3, 3)
    assert info["handler"] == b"MJPG" and info["rate"] / info["scale"] == 25 and info["usec_per_frame"] == 40000
    assert info["riff_size"] == info["file_size"] - 8 and len(frames) == 3
    assert all(f[:2] == b"\xff\xd8" and f[-2:] == b"\xff\xd9" for f in frames)


def test_mjpeg_avi_round_trip_and_rate_control(tmp_path):
    import io
    from PIL import Image
    from moonrtx_amd.video import MjpegAviWriter, read_avi_frames, pick_quality
    rng = np.random.default_rng(5)
    yy, xx = np.mgrid[0:96, 0:160]
    base = (60 + 50 * np.sin(xx / 17.0) * np.cos(yy / 11.0)).astype(np.float32)
    imgs = []
    for k in range(4):                                                          # 40 grey levels apart: a swapped pair would show
        rgb = np.stack([base + 40 * k, base * 0.9 + 40 * k, base * 0.8 + 40 * k], -1) + rng.normal(0, 2.0, (96, 160, 3))
        imgs.append(np.clip(rgb, 0, 255).astype(np.uint8))
    path = str(tmp_path / "t.avi")
    w = MjpegAviWriter(path, 160, 96, fps=29.97, bitrate_mbps=50.0)          # ample budget: top quality; 3 coder threads
    for im in imgs[:3]:
        w.add_frame(im)
    w.add_frame(np.dstack([imgs[3], np.full((96, 160), 255, np.uint8)]))       # RGBA frames as the facade hands them over
    assert w.quality == 95 and w.frames == 4 and w.open
    w.close(); w.close()
    with pytest.raises(RuntimeError):
        w.add_frame(imgs[0])
    info, frames = read_avi_frames(path)
    assert info["total_frames"] == 4 and (info["rate"], info["scale"]) == (29970, 1000) and info["usec_per_frame"] == 33367
    for im, jp in zip(imgs, frames):
        dec = np.asarray(Image.open(io.BytesIO(jp)).convert("RGB"), np.float32)
        mse = np.mean((dec - im.astype(np.float32)) ** 2)
        assert 10 * np.log10(255.0 ** 2 / mse) > 30.0                            # the frame that went in
    # coded inside add_frame (workers=0): the same bytes as the coder threads wrote, in the same order
    w0 = MjpegAviWriter(str(tmp_path / "t0.avi"), 160, 96, fps=29.97, bitrate_mbps=50.0, n_frames=4, workers=0)
    for im in imgs:
        w0.add_frame(im)
    assert not w0.open                                                           # closed by itself at the frame limit
    assert read_avi_frames(str(tmp_path / "t0.avi"))[1] == frames
    # many frames through few coders: order and count survive the back-pressure
    w3 = MjpegAviWriter(str(tmp_path / "t3.avi"), 160, 96, fps=30, bitrate_mbps=50.0, workers=2)
    for k in range(23):
        w3.add_frame(imgs[k % 4])
    w3.close()
    f3 = read_avi_frames(str(tmp_path / "t3.avi"))[1]
    assert len(f3) == 23 and all(f3[k] == frames[k % 4] for k in range(23))
    # a tight budget lowers the quality, but never below the floor
    big = len(frames[0])
    q = pick_quality(imgs[0], big // 2, q_min=40, q_max=95)
    assert 40 <= q < 95
    assert pick_quality(imgs[0], 10, q_min=70, q_max=95) == 70
    with pytest.raises(ValueError):
        MjpegAviWriter(str(tmp_path / "bad.avi"), 160, 96, fps=0)
    w2 = MjpegAviWriter(str(tmp_path / "shape.avi"), 160, 96, fps=30)
    with pytest.raises(ValueError):
        w2.add_frame(np.zeros((96, 161, 3), np.uint8))
    w2.close()
    assert read_avi_frames(str(tmp_path / "shape.avi"))[0]["total_frames"] == 0


def test_headless_root_is_tks_timer_queue():
    """after / after_idle / after_cancel as the reference uses them (moon_renderer.py:398-404, :467-480): one thread, due order,
    cancelled jobs never run, an exception in a callback does not stop the queue."""
    from moonrtx_amd.headless_ui import HeadlessRoot, HeadlessCanvas
    root = HeadlessRoot(64, 48)
    ran, threads = [], set()

    def note(tag):
        ran.append(tag); threads.add(threading.current_thread().name)

    def boom():
        raise RuntimeError("callback error (expected in this test)")

    a = root.after(60, note, "late")
    root.after(20, note, "mid")
    root.after(0, boom)
    c = root.after(30, note, "cancelled")
    root.after_idle(note, "idle")
    root.after_cancel(c)
    assert a.startswith("after#") and root.pending() >= 3
    t0 = time.time()
    while len(ran) < 3 and time.time() - t0 < 5.0:
        time.sleep(0.005)
    assert ran == ["idle", "mid", "late"] and threads == {"moonrt-ui"}
    root.after_cancel(a); root.after_cancel("after#999")               # done / unknown ids: harmless, as in Tk
    assert root.wait_idle(2.0) and root.pending() == 0
    # the inert window calls renderer_status.py / renderer_dialogs.py make
    root.title("MoonRTX"); root.state("zoomed")
    assert root.title() == "MoonRTX" and root.state() == "zoomed" and (root.winfo_width(), root.winfo_x(), root.winfo_y()) == (64, 0, 0)
    seen = []
    root.bind("<F10>", lambda e: seen.append(("f10", threading.current_thread().name)))
    root.fire("<F10>"); assert root.wait_idle(2.0) and seen == [("f10", "moonrt-ui")]
    root.destroy()
    assert root.after(0, note, "after destroy") is None and root.winfo_exists() == 0
    # the measuring line of renderer_navigation.py:631-682
    cv = HeadlessCanvas(64, 48)
    line = cv.create_line(1, 2, 3, 4, fill="yellow", width=2)
    cv.coords(line, 1, 2, 30, 40)
    assert cv.coords(line) == [1.0, 2.0, 30.0, 40.0] and cv.items[line]["options"]["fill"] == "yellow"
    cv.delete(line); assert cv.items == {} and cv.coords(line) == []


def test_preview_restore_and_video_export_run_on_the_headless_queue(tmp_path):
    """The two flows of the reference that go through rt._root: (1) _begin_interactive_preview / _end_interactive_preview
    (moon_renderer.py:457-488): single-frame cycles during a burst of edits, converged rendering restored by a timer; (2) the video
    export's accum-done callback (renderer_video.py:276-363): per frame rt._root.after(0, progress), at the end
    rt._root.after(delay, finish) where finish stops the encoder on the UI thread."""
    from moonrtx_amd.video import read_avi_frames
    rt, be, _ = make()
    assert rt._root is not None and rt._canvas is not None
    drive_like_init_renderer(rt, np.zeros((8, 16), np.float32), np.zeros((4, 8, 4), np.uint8))
    rt.start()
    # (1) a burst of three edits, each re-arming the restore timer
    state = {"preview": False, "restore_id": None, "restored_on": None}

    def end_preview():
        state["restore_id"] = None; state["preview"] = False; state["restored_on"] = threading.current_thread().name
        rt.set_param(max_accumulation_frames=64); rt.refresh_scene()

    for k in range(3):
        if not state["preview"]:
            state["preview"] = True; rt.set_param(max_accumulation_frames=1)
        if state["restore_id"] is not None:
            rt._root.after_cancel(state["restore_id"])
        state["restore_id"] = rt._root.after(40, end_preview)
        rt.update_light("sun", pos=[10.0 * k, 0.0, 0.0]); rt.refresh_scene()
        time.sleep(0.01)
    t0 = time.time()
    while state["restored_on"] is None and time.time() - t0 < 5.0:
        time.sleep(0.005)
    assert state["restored_on"] == "moonrt-ui" and rt.get_param("max_accumulation_frames") == 64
    spps = [c[2].get("spp_per_launch") for c in be.calls if c[0] == "set_params" and "spp_per_launch" in c[2]]
    assert 1 in spps and spps[-1] == 64                                # preview launches, then the converged one again
    # (2) video export, three frames
    rt.encoder_create(fps=30, bitrate=16)
    rt.encoder_start(str(tmp_path / "v.avi"), 3)
    st = {"frame": 0, "progress": [], "finished": threading.Event()}

    def progress(frame, total):
        st["progress"].append((frame, total, threading.current_thread().name))

    def finish():
        if rt.encoder_is_open():
            rt.encoder_stop()
        st["finished"].set()

    def accum_done(r):                                                  # render thread, padlock held
        st["frame"] += 1
        if st["frame"] < 3:
            r.update_light("sun", pos=[5.0 * st["frame"], 1.0, 0.0]); r.refresh_scene()
            r._root.after(0, progress, st["frame"], 3)
        else:
            r.set_accum_done_cb(None)
            r._root.after(50, finish)

    rt.set_accum_done_cb(accum_done)
    rt.refresh_scene()
    assert st["finished"].wait(20.0)
    assert st["progress"] == [(1, 3, "moonrt-ui"), (2, 3, "moonrt-ui")]
    assert not rt.encoder_is_open() and rt.encoded_frames() == 3
    rt.close()
    assert rt._root.winfo_exists() == 0
    assert read_avi_frames(str(tmp_path / "v.avi"))[0]["total_frames"] == 3
    off = TkOptiX(width=16, height=8, backend=RecordingBackend(16, 8), headless_ui=False)
    assert off._root is None and off._canvas is None                   # what the reference reads as "no GUI"
    off.close()
