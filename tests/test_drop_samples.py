"""drop_samples = false (TrajectorySampler::setSampleDroppingMode, trajectory_sampler.cpp:103-105,157-168): a
sample that collides beyond numCtrlPoints_ is frozen at its last free point with zero velocities and stays
admissible -- the only DWA configuration in which smoothness / jerk (A9) are non-zero inside a controller
cycle.  Every path of the device (single launch, three kernels with either cost kernel, split roll-out)
against the oracle's restatement: admissible set, every float of every path, every cost, winner, index."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import kompass_hip as kh  # noqa: E402
import synthetic as syn  # noqa: E402
from oracle import ko  # noqa: E402

from helpers import hip_context  # noqa: E402


def _oracle(inp, drop, num_ctrl, weights, threads=None):
    rb = inp["robot"]
    coll = ko.Collision(rb["shape"], rb["dims"], (0, 0, 0), (0, 0, 0, 1), inp["octree_res"])
    st = inp["state"]
    coll.update_state(st[0], st[1], st[2])
    coll.update_points(inp["points"], True)
    ox, oy = ko.obstacles_from_points((0, 0, 0), (0, 0, 0, 1), st, inp["points"])
    ci = ko.CostInputs(inp["seg_xyz"], 0, inp["acc_at_seg"], inp["ref_len"], np.stack([ox, oy], axis=1),
                       np.float32(inp["max_range"]) / np.float32(3.0), inp["acc_limits"], ko.make_weights(*weights))
    o = ko.full_cycle_mode(coll, ci, st, inp["dt"], inp["P"], inp["vx"], inp["vy"], inp["omega"], drop, num_ctrl, threads)
    o["keep"] = (coll, ci)
    return o


def _run(inp, weights, num_ctrl, drop, opts):
    ctx = hip_context(kh, inp)
    for k, v in opts.items():
        ctx.set_option(k, v)
    ctx.set_option("num_ctrl_points", num_ctrl)
    ctx.set_option("drop_samples", 1 if drop else 0)
    st = inp["state"]
    ctx.set_weights(kh.make_weights(*weights))
    ctx.set_points(st, inp["points"], inp["max_range"])
    ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
    ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
    res = ctx.cycle(st, inp["P"])
    best = ctx.get_best() if res.found else None
    px, py, raw, costs = ctx.get_samples(with_costs=True)
    frz = ctx.get_freeze_steps()
    out = dict(res=res.as_dict(), best=best, px=px.copy(), py=py.copy(), raw=raw.copy(), costs=costs.copy(), frz=frz.copy(),
               single=int(ctx.get_option("last_cycle_single_launch")))
    ctx.close()
    return out


def _check(o, h, inp):
    P = inp["P"]
    assert h["res"]["n_admissible"] == len(o["raw"])
    np.testing.assert_array_equal(h["raw"], o["raw"])
    np.testing.assert_array_equal(h["px"].view(np.uint32), o["px"].view(np.uint32))
    np.testing.assert_array_equal(h["py"].view(np.uint32), o["py"].view(np.uint32))
    np.testing.assert_array_equal(h["costs"].view(np.uint32), o["costs"].view(np.uint32))
    # the velocity profiles: zero from the freeze step on
    zeros = (o["vel"][0] == 0) & (o["vel"][1] == 0) & (o["vel"][2] == 0)
    want = np.where(zeros.any(axis=1), zeros.argmax(axis=1), 0).astype(np.int32) if len(zeros) else np.zeros(0, np.int32)
    np.testing.assert_array_equal(h["frz"], want)
    assert h["res"]["found"] == (o["index"] >= 0)
    if o["index"] >= 0:
        assert h["res"]["index"] == o["index"] and h["res"]["raw_index"] == int(o["raw"][o["index"]])
        assert np.float32(h["res"]["cost"]) == np.float32(o["cost"])
        bx, by, bv = h["best"]
        np.testing.assert_array_equal(bx, o["px"][o["index"]])
        np.testing.assert_array_equal(by, o["py"][o["index"]])
        for q in range(3):
            np.testing.assert_array_equal(np.asarray(bv[q], np.float32), o["vel"][q][o["index"]])


PATHS = {
    "single_launch": dict(fused_cycle=2),
    "single_launch_ticket": dict(fused_cycle=2, host_reduce=0),
    "single_launch_rows": dict(fused_cycle=2, write_paths=1),
    "three_kernels_block": dict(fused_cycle=0, cost_kernel=1),
    "three_kernels_wave": dict(fused_cycle=0, cost_kernel=2, cost_batch=0),
    "three_kernels_wave_batched": dict(fused_cycle=0, cost_kernel=2, cost_batch=2),
    "split": dict(force_split=1),
}
ALL5 = (1.0, 1.0, 1.0, 1.0, 1.0)


@pytest.mark.parametrize("path", list(PATHS))
@pytest.mark.parametrize("name,scale,seed,num_ctrl", [("cfg2", 0.25, 3, 2), ("cfg2", 0.25, 4, 12), ("cfg5", 0.08, 5, 0),
                                                      ("cfg3", 0.06, 6, 30), ("cfg1", 1.0, 7, 1)])
def test_frozen_samples_match_the_oracle(path, name, scale, seed, num_ctrl):
    inp = syn.make_controller_inputs(name, seed=seed, scale=scale)
    o = _oracle(inp, False, num_ctrl, ALL5)
    o_drop = _oracle(inp, True, num_ctrl, ALL5)
    assert len(o["raw"]) >= len(o_drop["raw"])
    h = _run(inp, ALL5, num_ctrl, False, PATHS[path])
    _check(o, h, inp)
    # and the default mode on the same context settings is what it was
    h2 = _run(inp, ALL5, num_ctrl, True, PATHS[path])
    _check(o_drop, h2, inp)
    assert not h2["frz"].any()


def test_some_samples_freeze_and_their_velocity_costs_are_not_zero():
    """The scenario really exercises A9: frozen samples exist, and with only smoothness + jerk weighted
    their cost is > 0 while the never-colliding samples cost exactly 0."""
    inp = syn.make_controller_inputs("cfg2", seed=3, scale=0.25)
    w = (0.0, 0.0, 0.0, 1.0, 1.0)
    o = _oracle(inp, False, 2, w)
    h = _run(inp, w, 2, False, dict(fused_cycle=2))
    _check(o, h, inp)
    frozen = h["frz"] > 0
    assert frozen.sum() > 50 and (~frozen).sum() > 5
    assert (h["costs"][frozen] > 0).all() and (h["costs"][~frozen] == 0).all()


@pytest.mark.parametrize("scene", ["survey", "mid"])
def test_frozen_samples_at_baseline_size(scene):
    """cfg2 at full size (8192 x 50), all five weights, freeze mode: single launch and three kernels."""
    inp = syn.make_controller_inputs("cfg2", seed=0, scene=scene)
    o = _oracle(inp, False, 2, ALL5)
    assert len(o["raw"]) > 4000
    for opts in (dict(), dict(fused_cycle=0), dict(fused_cycle=0, cost_kernel=2, cost_batch=2)):
        h = _run(inp, ALL5, 2, False, opts)
        _check(o, h, inp)
