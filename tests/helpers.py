"""Shared drivers for the parity tests: run one controller cycle through the
CPU oracle and through the HIP C ABI on identical inputs."""
from __future__ import annotations

import numpy as np

from oracle import ko


def oracle_cycle(inp, scan=None, sensor_pos=(0, 0, 0), sensor_rot=(0, 0, 0, 1)):
    """Roll-out + costs with the oracle.  Returns dict with compacted paths,
    raw indices, costs, argmin (compacted index) and min cost."""
    rb = inp["robot"]
    coll = ko.Collision(rb["shape"], rb["dims"], sensor_pos, sensor_rot, inp["octree_res"])
    st = inp["state"]
    coll.update_state(st[0], st[1], st[2])
    if scan is not None:
        coll.update_scan(scan[0], scan[1])
        ox, oy = ko.obstacles_from_scan(sensor_pos, sensor_rot, st, scan[0], scan[1])
    else:
        coll.update_points(inp["points"], True)
        ox, oy = ko.obstacles_from_points(sensor_pos, sensor_rot, st, inp["points"])
    px, py, raw, _ = ko.rollout(coll, st, inp["dt"], inp["P"], inp["vx"], inp["vy"], inp["omega"])
    w = inp["weights"]
    ci = ko.CostInputs(inp["seg_xyz"], 0, inp["acc_at_seg"], inp["ref_len"],
                       np.stack([ox, oy], axis=1), np.float32(inp["max_range"]) / np.float32(3.0),
                       inp["acc_limits"], ko.make_weights(*w))
    if len(px):
        idx, cost, costs = ko.min_trajectory_cost(ci, px, py, None)
    else:
        idx, cost, costs = -1, 0.0, np.zeros(0, np.float32)
    return dict(px=px, py=py, raw=raw, costs=costs, index=idx, cost=cost, ci=ci, coll=coll)


def oracle_cycle_mt(inp, threads=None, sensor_pos=(0, 0, 0), sensor_rot=(0, 0, 0, 1)):
    """oracle_cycle for BASELINE-size batches: every sample rolled out and
    scored independently on all host cores (oracle/ko.full_cycle: the same
    per-sample C functions, compacted in generation order, first strict minimum)."""
    rb = inp["robot"]
    coll = ko.Collision(rb["shape"], rb["dims"], sensor_pos, sensor_rot, inp["octree_res"])
    st = inp["state"]
    coll.update_state(st[0], st[1], st[2])
    coll.update_points(inp["points"], True)
    ox, oy = ko.obstacles_from_points(sensor_pos, sensor_rot, st, inp["points"])
    ci = ko.CostInputs(inp["seg_xyz"], 0, inp["acc_at_seg"], inp["ref_len"],
                       np.stack([ox, oy], axis=1), np.float32(inp["max_range"]) / np.float32(3.0),
                       inp["acc_limits"], ko.make_weights(*inp["weights"]))
    out = ko.full_cycle(coll, ci, st, inp["dt"], inp["P"], inp["vx"], inp["vy"], inp["omega"], threads)
    out.update(ci=ci, coll=coll)
    return out


def hip_context(kh, inp, sensor_pos=(0, 0, 0), sensor_rot=(0, 0, 0, 1), max_samples=None, max_points=None):
    rb = inp["robot"]
    n = len(inp["vx"])
    ctx = kh.DwaContext(rb["shape"], rb["dims"], sensor_pos, sensor_rot, inp["octree_res"], inp["dt"],
                        max_samples=max_samples or max(n, 1), max_points=max_points or inp["P"],
                        max_segment=len(inp["seg_xyz"]), max_obstacles=max(len(inp.get("points", [])), 16),
                        acc_limits=inp["acc_limits"])
    return ctx


def hip_cycle(kh, inp, scan=None, sensor_pos=(0, 0, 0), sensor_rot=(0, 0, 0, 1), ctx=None):
    own = ctx is None
    if own:
        ctx = hip_context(kh, inp, sensor_pos, sensor_rot)
    st = inp["state"]
    ctx.set_weights(kh.make_weights(*inp["weights"]))
    if scan is not None:
        ctx.set_scan(st, scan[0], scan[1], inp["max_range"])
    else:
        ctx.set_points(st, inp["points"], inp["max_range"])
    ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
    ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
    res = ctx.cycle(st, inp["P"])
    # the winner row first: after a single-launch cycle it comes from the pinned record, while
    # get_samples makes the context produce every row again through the materialising kernel
    best = ctx.get_best() if res.found else None
    px, py, raw, costs = ctx.get_samples(with_costs=True)
    out = dict(px=px.copy(), py=py.copy(), raw=raw.copy(), costs=costs.copy(), res=res.as_dict(), ctx=ctx)
    if res.found:
        out["best"] = best
        again = ctx.get_best()   # and once more from the device rows
        np.testing.assert_array_equal(again[0], best[0])
        np.testing.assert_array_equal(again[1], best[1])
    return out


def assert_cycle_equal(o, h):
    """Bit-exact: admissible set, float paths, float costs, selected index."""
    assert h["res"]["n_admissible"] == len(o["raw"])
    np.testing.assert_array_equal(h["raw"], o["raw"])
    np.testing.assert_array_equal(h["px"].view(np.uint32), o["px"].view(np.uint32))
    np.testing.assert_array_equal(h["py"].view(np.uint32), o["py"].view(np.uint32))
    np.testing.assert_array_equal(h["costs"].view(np.uint32), o["costs"].view(np.uint32))
    if o["index"] < 0:
        assert not h["res"]["found"]
    else:
        assert h["res"]["found"]
        assert h["res"]["index"] == o["index"]
        assert h["res"]["raw_index"] == int(o["raw"][o["index"]])
        assert np.float32(h["res"]["cost"]) == np.float32(o["cost"])
        bx, by, bv = h["best"]
        np.testing.assert_array_equal(bx, o["px"][o["index"]])
        np.testing.assert_array_equal(by, o["py"][o["index"]])
