"""Randomised parity: a fixed-seed slice of tests/fuzz_cases.py through the HIP path and the oracle, bit for bit.
Includes seed 1 case 116, which exposed a shadow ray whose origin (hit + scene_epsilon * normal on a D = 1 texel) sits
just outside the bounding sphere and heads inward: the march ends at step 1 by the spec and must not resume where the
skipped steps dip back inside -- and seed 430 case 11: a camera inside the shell of overlay tubes, one tube behind the eye on
the line of sight of another: only intersections in front of the eye count (the round-1 spec picked the nearest along the whole
line and then dropped it, hiding the tube in view; the kernel's per-tile bins never held the one behind and showed it)."""
import itertools

import pytest

import fuzz_cases

pytestmark = pytest.mark.gpu


def test_random_cases_match_the_oracle(native_lib):
    total = 0
    for k, c in enumerate(itertools.islice(fuzz_cases.cases(1), 117)):
        if k < 40 or k == 116:
            total += fuzz_cases.check_case(c)["primary_hits"]
    for c in itertools.islice(fuzz_cases.cases(7), 25):
        total += fuzz_cases.check_case(c)["primary_hits"]
    total += fuzz_cases.check_case(next(itertools.islice(fuzz_cases.cases(430), 11, 12)))["primary_hits"]
    assert total > 100000


def test_exotic_cases_match_the_oracle(native_lib):
    """The rare combinations of the round-2 campaign (FUZZ_EXOTIC: camera inside the shell of overlay tubes in half of the
    cases, environment map and tubes in 70 %, paths in 85 %): the first 30 cases of exotic seed 13, in the driver-run suite."""
    total = 0
    for c in itertools.islice(fuzz_cases.cases(13, exotic=True), 30):
        total += fuzz_cases.check_case(c)["primary_rays"]
    assert total > 100000


def test_dense_dem_cases_match_the_oracle(native_lib):
    """Round 4's medium-mip scans decide per STEP which cell bounds it; the ordinary fuzz DEMs (<= 360 rows) keep a ray inside one
    cell for dozens of steps.  FUZZ_DENSE cases put 720 ... 2 880-row terrains of terraces, spikes, pits and one-texel walls under
    march steps of 1 ... 4.6 texels, with the generator's random cameras (limb-grazing, close-ups, inside the shell): the first 16
    cases of dense seed 21 in the driver-run suite, campaigns in profiles/r04_fuzz_campaign.log."""
    total = 0
    for c in itertools.islice(fuzz_cases.cases(21, dense=True), 16):
        total += fuzz_cases.check_case(c)["primary_rays"]
    assert total > 50000


def test_heavily_spilled_instantiation_matches_the_oracle():
    """Round 1 reported one miscomputed sample from render_kernel<64, STATS, !WIDE, in-wave paths, OVERLAY> under a 72-VGPR cap
    (~270 spilled VGPRs); the failing case was not kept and the march state has been restructured since (every field
    initialised before use), so this is a standing check, not a reproduction: the same configuration, rebuilt as
    moonrtx_amd/libmoonrt_spilltest.so (make spilltest), 40 fuzz scenes through exactly that kernel in a process of its own
    -- radiance, hits and counters equal the oracle's.  (500 cases passed in round 2, tools/spill_repro.py.)"""
    import os, subprocess, sys
    from moonrtx_amd import build
    lib = build.build_spilltest()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MOONRT_LIB=lib)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "spill_repro.py"), "40", "1"], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0 and "40 cases, 0 mismatches" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
