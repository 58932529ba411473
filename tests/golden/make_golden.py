"""Generate tests/golden/*.json by IMPORTING the reference modules that import cleanly here
(moonrtx.renderer_navigation).

Run in the build container only (needs /root/reference):   python tests/golden/make_golden.py
The outputs are data (inputs + expected outputs); no reference source text is stored.
"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, "/root/reference")
from moonrtx.renderer_navigation import NavigationMixin  # noqa: E402
from moonrtx import moon_grid  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def seeded_dem(h, w, seed):
    rng = np.random.default_rng(seed)
    e = (0.99 + 0.01 * rng.random((h, w))).astype(np.float32)
    e[rng.integers(0, h), rng.integers(0, w)] = 1.0
    return e


class _FakeRt:
    """Records what the mixin sends through the renderer boundary."""

    def __init__(self, eye, target):
        self.cam = {"Eye": list(eye), "Target": list(target), "Up": [0, 0, 1]}
        self.calls = []

    def get_camera(self, name):
        return self.cam

    def update_camera(self, name, **kw):
        self.calls.append(kw)


class Nav(NavigationMixin):
    MOON_RADIUS = 10.0
    MOON_RADIUS_KM = 1737.4
    CAMERA_NAME = "cam1"


def rot(ax, deg):
    a = np.radians(deg); c, s = np.cos(a), np.sin(a)
    return {"x": np.array([[1, 0, 0], [0, c, -s], [0, s, c]]),
            "y": np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]]),
            "z": np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]])}[ax]


def main():
    rng = np.random.default_rng(20260821)
    # 1. get_elevation_m on seeded DEMs: poles, seam, texel centres, random
    elev_cases = []
    for (h, w, seed) in [(8, 16, 1), (64, 128, 2), (90, 180, 3)]:
        nav = Nav()
        nav.elevation = seeded_dem(h, w, seed)
        nav.elevation_radius_scale = 1.0062
        pts = [(90.0, 0.0), (-90.0, 10.0), (89.99, -180.0), (0.0, 180.0), (0.0, -180.0), (0.0, 179.999),
               (0.0, -179.999), (12.5, 0.0), (-33.3, 77.7)]
        # texel centres
        for r in (0, 1, h // 2, h - 1):
            for c in (0, 1, w // 2, w - 1):
                pts.append((90.0 - (r + 0.5) * 180.0 / h, -180.0 + (c + 0.5) * 360.0 / w))
        for _ in range(40):
            pts.append((float(rng.uniform(-90, 90)), float(rng.uniform(-180, 180))))
        vals = [float(nav.get_elevation_m(la, lo)) for la, lo in pts]
        elev_cases.append({"h": h, "w": w, "seed": seed, "radius_scale": 1.0062,
                           "points": pts, "elevation_m": vals})
    json.dump({"source": "moonrtx.renderer_navigation.NavigationMixin.get_elevation_m (renderer_navigation.py:558-599)",
               "dem_generator": "tests/golden/make_golden.py::seeded_dem", "cases": elev_cases},
              open(os.path.join(HERE, "elevation_bilinear.json"), "w"), indent=1)

    # 2. body frame: center_on_lat_lon position <-> hit_to_selenographic
    frame_cases = []
    mats = [np.eye(3), rot("x", -5.0) @ rot("z", -3.0), rot("y", 21.0) @ rot("x", 6.5) @ rot("z", 7.9)]
    for R in mats:
        nav = Nav()
        nav.moon_rotation = R
        nav.moon_rotation_inv = R.T
        rows = []
        for la in (-88.0, -45.0, -5.0, 0.0, 12.0, 60.0, 89.0):
            for lo in (-179.0, -90.0, -3.0, 0.0, 45.0, 90.0, 178.0):
                nav.rt = _FakeRt([0.0, -300.0, 0.0], [0.0, 0.0, 0.0])
                nav.center_on_lat_lon(la, lo)
                scene_pos = nav.rt.calls[-1]["target"]
                back = nav.hit_to_selenographic(*scene_pos)
                rows.append({"lat": la, "lon": lo, "scene_pos": [float(t) for t in scene_pos],
                             "lat_back": float(back[0]), "lon_back": float(back[1])})
        off = nav.hit_to_selenographic(0.0, -20.0, 0.0)
        frame_cases.append({"rotation": R.tolist(), "rows": rows, "off_moon": [off[0], off[1]]})
    json.dump({"source": "NavigationMixin.center_on_lat_lon / hit_to_selenographic (renderer_navigation.py:27-73, :452-492)",
               "cases": frame_cases}, open(os.path.join(HERE, "body_frame.json"), "w"), indent=1)

    # 3. haversine
    nav = Nav()
    hv = []
    for _ in range(12):
        a = [float(rng.uniform(-90, 90)), float(rng.uniform(-180, 180)), float(rng.uniform(-90, 90)),
             float(rng.uniform(-180, 180))]
        hv.append({"args": a, "km": float(nav.calculate_great_circle_distance(*a))})
    json.dump({"source": "NavigationMixin.calculate_great_circle_distance (renderer_navigation.py:525-556)", "cases": hv},
              open(os.path.join(HERE, "haversine.json"), "w"), indent=1)

    print("golden vectors written to", HERE)



def grid_graphs():
    """4. The overlay graphs MoonRTX hands to set_graph (renderer_labels.py:172-177, :291-300): grid lines and grid
    labels from moon_grid.create_moon_grid() + merge_segments_to_graph().  Stored as data: vertex / edge arrays."""
    g = moon_grid.create_moon_grid()
    lines_pos, lines_edges = moon_grid.merge_segments_to_graph(g.lat_lines + g.lon_lines)
    segs = [seg for segs in g.lat_labels for seg in segs] + [seg for segs in g.lon_labels for seg in segs] + list(g.N)
    labels_pos, labels_edges = moon_grid.merge_segments_to_graph(segs)
    np.savez_compressed(os.path.join(HERE, "moon_grid_graphs.npz"),
                        lines_pos=np.asarray(lines_pos, np.float32), lines_edges=np.asarray(lines_edges, np.int32),
                        labels_pos=np.asarray(labels_pos, np.float32), labels_edges=np.asarray(labels_edges, np.int32))
    meta = {"source": "moonrtx.moon_grid.create_moon_grid + merge_segments_to_graph (moon_grid.py:27-46, :689-790)",
            "lines": {"pos_shape": list(np.shape(lines_pos)), "edges_shape": list(np.shape(lines_edges)),
                      "pos_sum": float(np.sum(np.asarray(lines_pos, np.float64))), "pos_abs_sum": float(np.abs(lines_pos).sum()),
                      "edge_sum": int(np.sum(lines_edges))},
            "labels": {"pos_shape": list(np.shape(labels_pos)), "edges_shape": list(np.shape(labels_edges)),
                       "pos_abs_sum": float(np.abs(labels_pos).sum()), "edge_sum": int(np.sum(labels_edges))},
            "radii": {"grid_line": 0.006, "grid_label": 0.012}, "colour": [0.5, 0.5, 0.5]}
    json.dump(meta, open(os.path.join(HERE, "moon_grid_graphs.json"), "w"), indent=1)
    print("grid graphs:", meta["lines"]["pos_shape"], meta["lines"]["edges_shape"], meta["labels"]["pos_shape"], meta["labels"]["edges_shape"])

if __name__ == "__main__":
    main()
    grid_graphs()
