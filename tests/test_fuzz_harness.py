"""The fuzz harness tells "an input array changed between the two renders" from a parity failure (CPU only).

Regression for the one non-repeating failure of round 3 (seed 601 case 49), whose record is reproduced exactly by the oracle when
one 32-bit word of the case's colour map is decremented by one between the oracle's render and the upload
(tools/replay_seed601_case49.py)."""
import itertools

import numpy as np
import pytest

import common
import fuzz_cases


def _case(seed, k):
    return next(itertools.islice(fuzz_cases.cases(seed), k, None))


def test_seed601_case49_record_is_a_decremented_colour_word():
    """The failure record, rebuilt on the CPU: word 38 of the colour map minus one gives the 708 values / 22.310598 / 36.66."""
    desc, dem, col, bg, s, flags, tile, blocks, extra = _case(601, 49)
    assert col.shape == (12, 19, 4) and tuple(col[2, 0]) == (0, 105, 194, 165)
    ref = common.render_oracle(s, dem, col, bg, blocks=blocks)[0]
    assert ref[13, 56, 0] == np.float32(22.296503)
    c2 = col.copy()
    c2.reshape(-1).view(np.uint32)[38] -= 1
    assert tuple(c2[2, 0]) == (255, 104, 194, 165)
    lin = common.render_oracle(s, dem, c2, bg, blocks=blocks)[0]
    d = lin.view(np.uint32) != ref.view(np.uint32)
    assert int(d.sum()) == 708
    assert tuple(int(t) for t in np.argwhere(d)[0]) == (13, 56, 0) and lin[13, 56, 0] == np.float32(22.310598)
    assert abs(float(np.abs(lin.astype(np.float64) - ref).max()) - 36.66) < 0.005


def test_harness_reports_a_changed_input_not_a_parity_failure(monkeypatch):
    """A render_hip stand-in that decrements one colour word before it 'uploads' (what the failing process did to the array)."""
    c = _case(601, 49)
    desc, dem, col, bg, s, flags, tile, blocks, extra = c

    def hip_with_a_stray_decrement(scene, dem_, color=None, bg_=None, blocks=(1,), **kw):
        color.reshape(-1).view(np.uint32)[38] -= 1          # the stray write lands in the caller's array
        lin, hits, st, img = common.render_oracle(scene, dem_, color, bg_, blocks=blocks, rgba8=True)
        return lin, hits, st, img

    monkeypatch.setattr(common, "render_hip", hip_with_a_stray_decrement)
    with pytest.raises(fuzz_cases.InputChanged, match=r"colour array CHANGED UNDER THE TEST.*byte offset 152.*0069c2a5 -> ff68c2a5"):
        fuzz_cases.check_case(c)
    col.reshape(-1).view(np.uint32)[38] += 1


def test_harness_passes_when_both_sides_see_the_same_bytes(monkeypatch):
    c = _case(601, 49)

    def hip_is_the_oracle(scene, dem_, color=None, bg_=None, blocks=(1,), **kw):
        return common.render_oracle(scene, dem_, color, bg_, blocks=blocks, rgba8=True)

    monkeypatch.setattr(common, "render_hip", hip_is_the_oracle)
    st = fuzz_cases.check_case(c)
    assert st["primary_hits"] == 2576


def test_dense_flavour_leaves_the_ordinary_cases_alone():
    """FUZZ_DENSE draws its terrain LAST from the generator's second stream: case k of a seed keeps its scene, camera, flags and
    options whether or not the flavour is on (so a failing seed of an old campaign still means the same case), and with it most
    cases get a terrain of >= 720 rows under a march step of at least 1.3e-2 radii."""
    import itertools
    import numpy as np
    plain = list(itertools.islice(fuzz_cases.cases(5, dense=False), 12))
    dense = list(itertools.islice(fuzz_cases.cases(5, dense=True), 12))
    big = 0
    for a, b in zip(plain, dense):
        sa, sb = a[4], b[4]
        assert (sa.width, sa.height, sa.spp_per_launch, sa.eye, sa.target, sa.vfov_deg, sa.light_pos) == \
               (sb.width, sb.height, sb.spp_per_launch, sb.eye, sb.target, sb.vfov_deg, sb.light_pos)
        assert a[5:8] == b[5:8] and a[8]["world"] == b[8]["world"] and a[8]["inwave"] == b[8]["inwave"]
        if b[1].shape[0] >= 720:
            big += 1
            assert b[4].marching_step >= 1.3e-2 * b[4].radius / 10.0 * 0.999 and float(b[1].max()) == 1.0 and float(b[1].min()) > 0.9
        else:
            assert np.array_equal(a[1], b[1]) and sa.marching_step == sb.marching_step
    assert big >= 5
