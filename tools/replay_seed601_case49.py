#!/usr/bin/env python3
"""CPU-only reconstruction of the one non-repeating fuzz failure of round 3 (seed 601 case 49; profiles/r03_fuzz_campaign.log:6):

    radiance: 708 of 15904 values differ bitwise; max |diff| = 3.666e+01; first at (13, 56, 0): 22.310598 vs 22.296503

The case is regenerated from its seed, rendered with the ORACLE ONLY under one hypothesis after the other, and each result is
compared with the oracle's own frame the way the failing run compared the HIP frame with it (count of differing values, largest
difference, first differing value).  A hypothesis that reproduces all of the record's numbers names what the HIP side was given.
Test infrastructure (uses oracle/); writes nothing.  Result of the run: profiles/r04_seed601_case49.md."""
import copy
import itertools
import os
import sys

import numpy as np

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_cases   # noqa: E402
from oracle import orc   # noqa: E402

RECORD = dict(values=708, maxdiff=36.66, first=(13, 56, 0), hip=np.float32(22.310598), oracle=np.float32(22.296503))

cs = list(itertools.islice(fuzz_cases.cases(601), 50))
desc, dem, col, bg, s, flags, tile, blocks, extra = cs[49]
prev = cs[48][4]
print(desc)


def frame(scene, dem_, col_, bg_):
    o = orc.Oracle(scene, dem_, col_, bg_)
    for nb in blocks:
        o.render(nb)
    return o.linear()


ref = frame(s, dem, col, bg)
print(f"oracle at {RECORD['first']}: {ref[RECORD['first']]!r} (the record's oracle value: {RECORD['oracle']!r})")
assert ref[RECORD["first"]] == RECORD["oracle"]


def signature(lin):
    d = lin.view(np.uint32) != ref.view(np.uint32)
    n = int(d.sum())
    if n == 0:
        return "identical"
    i = tuple(int(t) for t in np.argwhere(d)[0])
    md = float(np.abs(lin.astype(np.float64) - ref).max())
    hit = n == RECORD["values"] and i == RECORD["first"] and lin[i] == RECORD["hip"] and abs(md - RECORD["maxdiff"]) < 0.005
    return f"{n:5d} values, max |diff| {md:9.4g}, first at {i}: {lin[i]!r}" + ("   <== THE RECORD" if hit else "")


def hyp(name, **kw):
    s2 = copy.deepcopy(s)
    d2, c2, b2 = dem, col, bg
    for k, v in kw.items():
        if k == "col":
            c2 = v
        elif k == "bg":
            b2 = v
        else:
            setattr(s2, k, v)
    print(f"  {name:58s} {signature(frame(s2, d2, c2, b2))}")


print("stale-state hypotheses (what the previous case, seed 601 case 48, or the context defaults would have left behind):")
hyp("scene_epsilon of case 48 / the default (1e-4)", scene_epsilon=1e-4)
hyp("marching_step_eps of case 48", marching_step_eps=prev.marching_step_eps)
hyp("no colour map (case 48 had none)", col=None)
hyp("no environment map (case 48 had none)", bg=None)
hyp("light of case 48", light_pos=prev.light_pos)
hyp("RNG seed of case 48", seed=prev.seed)
hyp("Sun disk of case 48", sun_pos=prev.sun_pos, sun_radius=prev.sun_radius)

print("one colour texel different: which texels does pixel (13, 56) read, and which byte value gives the record's 22.310598?")
o = orc.Oracle(s, dem, col, bg)


def px(x, y):
    o.reset(); o.render(1, (x, y, x + 1, y + 1))
    return o.linear()[y, x].copy()


y0, x0, _ = RECORD["first"]
base = px(x0, y0)
foot = []
for r in range(col.shape[0]):
    for c in range(col.shape[1]):
        old = o.color[r, c, 0]
        o.color[r, c, 0] = old ^ 0x80
        if px(x0, y0)[0] != base[0]:
            foot.append((r, c))
        o.color[r, c, 0] = old
print("  footprint:", foot)
found = []
for (r, c) in foot:
    old = o.color[r, c, 0]
    for b in range(256):
        o.color[r, c, 0] = b
        if px(x0, y0)[0] == RECORD["hip"]:
            found.append((r, c, int(old), b))
    o.color[r, c, 0] = old
print("  (row, col, red as generated, red that gives the record's value):", found)
assert found == [(2, 0, 0, 255)]
r, c = 2, 0
word = int(col[r, c].view(np.uint32)[0])
print(f"  texel ({r}, {c}) = bytes {tuple(int(t) for t in col[r, c])} = little-endian word 0x{word:08X} at byte offset {(r * col.shape[1] + c) * 4} "
      f"of the {col.nbytes}-byte map; word - 1 = 0x{word - 1:08X}")
c2 = col.copy()
c2[r, c] = np.array([word - 1], np.uint32).view(np.uint8)
print(f"  bytes after the decrement: {tuple(int(t) for t in c2[r, c])}")
hyp("texel (2, 0) = its 32-bit word minus one", col=c2)
c3 = col.copy(); c3[r, c, 0] = 255
hyp("texel (2, 0): red 255 only (one byte stale)", col=c3)
