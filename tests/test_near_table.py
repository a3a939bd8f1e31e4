"""Near table of the tracked segment (option `near_table`, segment_near_kernel) and the
pair records of the segment scans: the wavefront-per-sample search must give the bits of
the oracle's full scan (cost_evaluator.cpp:111-184: pathCostFunc / goalCostFunc) for every
table size, with the table off, and on segments that stress its construction -- odd
point counts, one chunk, odd chunk sizes made even, z != 0, a hairpin (two far-apart runs
of candidate chunks), a point the robot cannot be near, poses far from the origin
(float cell arithmetic), non-finite segment points.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import kompass_hip as kh  # noqa: E402
import synthetic as syn  # noqa: E402

from helpers import assert_cycle_equal, hip_context, hip_cycle, oracle_cycle  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert kh.device_count() >= 1, "no HIP device visible: the -m gpu tests need an MI355X"


def _base(scale=0.25, scene="open", name="cfg2", seed=5):
    inp = syn.make_controller_inputs(name, seed=seed, scale=scale, scene=scene)
    return inp


def _acc(xyz):
    d = np.diff(xyz.astype(np.float64), axis=0)
    return np.concatenate([[0.0], np.cumsum(np.hypot(d[:, 0], d[:, 1]))]).astype(np.float32)


def _segments():
    out = {}
    for S in (2, 3, 17, 33, 500, 1001):
        out[f"straight{S}"] = syn.straight_segment(S, 0.013)
    out["straight5000"] = syn.straight_segment(5000, 0.003)   # pair records beyond the table builder's LDS: rows from global memory
    xyz, acc = syn.arc_segment(1300, 6.0, 0.012)      # chunk 21 -> 22 points
    out["arc1300"] = (xyz, acc)
    xyz, acc = syn.arc_segment(777, 2.5, 0.01)        # tight arc: the robot sits near the centre of curvature
    out["arc_tight"] = (xyz, acc)
    # hairpin: out along +x, back 0.4 m to the side -- the nearest point jumps between the two legs
    s = np.arange(400) * 0.01
    leg1 = np.stack([s, np.zeros_like(s), np.zeros_like(s)], 1)
    th = np.linspace(0, np.pi, 63)[1:-1]
    turn = np.stack([s[-1] + 0.2 * np.sin(th), 0.2 - 0.2 * np.cos(th), np.zeros_like(th)], 1)
    leg2 = np.stack([s[::-1], np.full_like(s, 0.4), np.zeros_like(s)], 1)
    hp = np.concatenate([leg1, turn, leg2]).astype(np.float32)
    out["hairpin"] = (hp, _acc(hp))
    # z != 0 (the distance carries z^2 of every segment point)
    xyz, acc = syn.arc_segment(501, 10.0, 0.01)
    xyz = xyz.copy()
    xyz[:, 2] = (0.3 * np.sin(np.arange(501) * 0.05)).astype(np.float32)
    out["wavy_z"] = (xyz, acc)
    # a segment far from everything the lattice reaches
    xyz, acc = syn.straight_segment(300, 0.01)
    out["far_away"] = (xyz + np.float32([40.0, -25.0, 0.0]), acc)
    # non-finite points in the middle
    xyz, acc = syn.straight_segment(200, 0.02)
    xyz = xyz.copy()
    xyz[70] = [np.nan, 0.0, 0.0]
    xyz[120] = [np.inf, 1.0, 0.0]
    out["nonfinite"] = (xyz, acc)
    return out


_SEGS = _segments()


@pytest.mark.parametrize("seg", sorted(_SEGS))
def test_near_table_on_awkward_segments(seg):
    inp = _base()
    xyz, acc = _SEGS[seg]
    inp["seg_xyz"], inp["acc_at_seg"] = xyz, acc
    inp["ref_len"] = float(max(acc[-1], 1.0))
    o = oracle_cycle(inp)
    assert len(o["raw"]) > 100
    for opts in (dict(cost_kernel=2, fused_cycle=0), dict(fused_cycle=2), dict(cost_kernel=2, fused_cycle=0, near_table=0),
                 dict(cost_kernel=2, fused_cycle=0, near_table=16), dict(fused_cycle=2, near_table=500),
                 dict(cost_kernel=2, fused_cycle=0, force_split=1)):
        ctx = hip_context(kh, inp)
        for k, v in opts.items():
            ctx.set_option(k, v)
        assert_cycle_equal(o, hip_cycle(kh, inp, ctx=ctx))
        # a second cycle: the table is kept (same segment, same box)
        r = ctx.cycle(inp["state"], inp["P"])
        assert r.index == o["index"]
        if r.found:
            assert np.float32(r.cost) == np.float32(o["cost"])
        ctx.close()


@pytest.mark.parametrize("origin", [(0.0, 0.0), (5000.25, -3000.5), (-12345.0, 6789.0)])
def test_near_table_far_from_the_origin(origin):
    """Everything shifted: the kernels take a point's cell from float differences of large coordinates."""
    inp = _base(scene="mid")
    dx, dy = origin
    sh = np.float32([dx, dy, 0.0])
    inp["seg_xyz"] = (np.asarray(inp["seg_xyz"], np.float32) + sh).astype(np.float32)
    inp["state"] = (inp["state"][0] + dx, inp["state"][1] + dy, inp["state"][2], inp["state"][3])
    # (the sensor points are in the robot frame: they move with the state)
    o = oracle_cycle(inp)
    assert len(o["raw"]) > 50
    for opts in (dict(cost_kernel=2, fused_cycle=0), dict(fused_cycle=2), dict(cost_kernel=2, fused_cycle=0, near_table=0)):
        ctx = hip_context(kh, inp)
        for k, v in opts.items():
            ctx.set_option(k, v)
        assert_cycle_equal(o, hip_cycle(kh, inp, ctx=ctx))
        ctx.close()


def test_near_table_follows_segment_and_pose_updates():
    """The table is rebuilt when the segment changes or the reachable box leaves it, kept otherwise:
    a pose walk with a segment swap in the middle, every cycle against a context without the table."""
    inp = _base(scene="mid", scale=0.3)
    a = hip_context(kh, inp)
    b = hip_context(kh, inp)
    for c, nt in ((a, 128), (b, 0)):
        c.set_option("near_table", nt)
        c.set_option("cost_kernel", 2)
        c.set_option("fused_cycle", 0)
        c.set_weights(kh.make_weights(*inp["weights"]))
        c.set_points(inp["state"], inp["points"], inp["max_range"])
        c.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
        c.set_samples(inp["vx"], inp["vy"], inp["omega"])
    P = inp["P"]
    xyz2, acc2 = syn.arc_segment(400, 5.0, 0.015)
    for i in range(60):
        st = (0.11 * i, -0.07 * i, 0.03 * (i % 11), 0.0)
        if i == 30:
            for c in (a, b):
                c.set_tracked_segment(xyz2, acc2, inp["ref_len"])
        ra, rb = a.cycle(st, P), b.cycle(st, P)
        assert (ra.found, ra.index, ra.raw_index, ra.n_admissible) == (rb.found, rb.index, rb.raw_index, rb.n_admissible), i
        assert np.float32(ra.cost) == np.float32(rb.cost), i
        if i % 20 == 3:
            ca = a.get_samples(with_costs=True)[3]
            cb = b.get_samples(with_costs=True)[3]
            np.testing.assert_array_equal(ca.view(np.uint32), cb.view(np.uint32))
    a.close(); b.close()


def test_near_table_option_range():
    inp = _base()
    ctx = hip_context(kh, inp)
    assert ctx.get_option("near_table") == 128
    for bad in (1, 15, 513, -4):
        with pytest.raises(IndexError):   # KC_ERR_RANGE
            ctx.set_option("near_table", bad)
    ctx.set_option("near_table", 0)
    assert ctx.get_option("near_table") == 0
    ctx.close()


def test_near_table_with_the_resident_path_window():
    """The tracked segment as a window of a device-resident path (kc_dwa_set_path +
    kc_dwa_set_tracked_window: tables written by segment_window_kernel, the near table built behind it in
    stream order): a sliding window, every cycle against a context that gets the same points through
    kc_dwa_set_tracked_segment and runs without the table."""
    inp = _base(scene="open", scale=0.3)
    xyz, acc = syn.arc_segment(1500, 8.0, 0.01)
    a = hip_context(kh, dict(inp, seg_xyz=xyz[:600]))
    b = hip_context(kh, dict(inp, seg_xyz=xyz[:600]))
    for c, nt in ((a, 128), (b, 0)):
        c.set_option("near_table", nt)
        c.set_option("cost_kernel", 2)
        c.set_weights(kh.make_weights(*inp["weights"]))
        c.set_points(inp["state"], inp["points"], inp["max_range"])
        c.set_samples(inp["vx"], inp["vy"], inp["omega"])
    a.set_path(xyz, acc, float(acc[-1]))
    P = inp["P"]
    for i in range(40):
        start, size = 15 * i, 420 + (i % 5) * 37
        a.set_tracked_window(start, size)
        seg = xyz[start:start + size]
        b.set_tracked_segment(seg, acc[start:start + size], float(acc[-1]))
        st = (float(seg[0, 0]), float(seg[0, 1]), 0.02 * (i % 9), 0.0)
        ra, rb = a.cycle(st, P), b.cycle(st, P)
        assert (ra.found, ra.index, ra.raw_index, ra.n_admissible) == (rb.found, rb.index, rb.raw_index, rb.n_admissible), i
        assert np.float32(ra.cost) == np.float32(rb.cost), i
        if i % 13 == 5:
            np.testing.assert_array_equal(a.get_samples(with_costs=True)[3].view(np.uint32),
                                          b.get_samples(with_costs=True)[3].view(np.uint32))
    a.close(); b.close()
