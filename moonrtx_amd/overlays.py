"""Overlay graph geometry (D11): PlotOptiX `set_graph(name, pos=, edges=, r=, c=, mat=)` -> flat capsule list.

MoonRTX merges each overlay group (grid lines, grid labels, feature labels, pins) into one graph: vertices, edge
index pairs, a scalar or per-vertex radius (0 hides a vertex' edges) and one colour
(renderer_labels.py:295-300, :367-373; renderer_pins.py:54).  The renderer consumes capsules:
12 floats each -- ax ay az r  bx by bz 0  cr cg cb 0 -- in scene coordinates."""
import numpy as np


def graph_to_capsules(pos, edges, r, c):
    pos = np.asarray(pos, np.float64).reshape(-1, 3)
    edges = np.asarray(edges, np.int64).reshape(-1, 2)
    if edges.size == 0 or pos.size == 0:
        return np.zeros((0, 12), np.float32)
    rr = np.broadcast_to(np.asarray(r, np.float64).reshape(-1), (pos.shape[0],)) if np.ndim(r) else np.full(pos.shape[0], float(r))
    col = np.asarray(c, np.float64).reshape(-1)
    col = np.full(3, col[0]) if col.size == 1 else col[:3]
    ra, rb = rr[edges[:, 0]], rr[edges[:, 1]]
    rad = np.minimum(ra, rb)                       # an edge is visible only if both of its vertices are
    keep = rad > 0.0
    out = np.zeros((int(keep.sum()), 12), np.float32)
    out[:, 0:3] = pos[edges[keep, 0]]
    out[:, 3] = rad[keep]
    out[:, 4:7] = pos[edges[keep, 1]]
    out[:, 8:11] = col
    return out


def graticule(radius=10.25, step_deg=15.0, tube=0.012, colour=(0.5, 0.5, 0.5), points=64, rotation=None):
    """A latitude/longitude grid as (pos, edges) in scene coordinates -- a stand-in for moon_grid.create_moon_grid in
    tests and demos (body frame: +Z north, -Y longitude 0, +X longitude 90 E; renderer_navigation.py:47-53)."""
    lines = []
    t = np.linspace(0.0, 2 * np.pi, points + 1)
    for lat in np.arange(-90 + step_deg, 90, step_deg):
        la = np.radians(lat)
        lines.append(np.stack([np.cos(la) * np.sin(t), -np.cos(la) * np.cos(t), np.full_like(t, np.sin(la))], 1))
    u = np.linspace(-np.pi / 2, np.pi / 2, points // 2 + 1)
    for lon in np.arange(-180, 180, step_deg):
        lo = np.radians(lon)
        lines.append(np.stack([np.cos(u) * np.sin(lo), -np.cos(u) * np.cos(lo), np.sin(u)], 1))
    pos = np.concatenate(lines) * radius
    if rotation is not None:
        pos = pos @ np.asarray(rotation, float).T
    edges, off = [], 0
    for ln in lines:
        idx = np.arange(off, off + len(ln))
        edges.append(np.stack([idx[:-1], idx[1:]], 1))
        off += len(ln)
    return pos, np.concatenate(edges).astype(np.int32), tube, colour
