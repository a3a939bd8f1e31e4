"""The long-list cost kernel with its per-sample part batched (sample_cost_batched_kernel, option "cost_batch"): a
wavefront per sample for the per-point work, then 64 samples at once -- a lane a sample -- for the ordered sum of
pathCostFunc (cost_evaluator.cpp:111-141), the end point's nearest segment index (goalCostFunc :157-176), the weighted
total in the reference's accumulation order (:59-100) and the key.  Same additions in the same order, same first
minimum: every cost and the selected index bit-equal to the oracle, for trajectory lengths around the tile size,
curved / non-planar segments, precomputed velocity sums, several buffers per workgroup (the double-buffer hand-off)
and the default rule (long lists only); few / clustered / dense obstacles (ring walks, the cooperative far pass, the
union rectangle), a LaserScan (the scan's near table) and list lengths that leave a short last group."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import kompass_hip as kh  # noqa: E402
import synthetic as syn  # noqa: E402
from oracle import ko  # noqa: E402

from helpers import assert_cycle_equal, hip_context, hip_cycle, oracle_cycle_mt  # noqa: E402

STATE = (0.2, -0.1, 0.3, 0.0)


def _paths(N, P, rng):
    v = rng.uniform(0.0, 1.2, (N, 1))
    w = rng.uniform(-1.2, 1.2, (N, 1))
    th = rng.uniform(-np.pi, np.pi, (N, 1)) + w * 0.1 * np.arange(P)[None]
    x = rng.uniform(-2, 2, (N, 1)) + np.cumsum(v * np.cos(th) * 0.1, 1)
    y = rng.uniform(-2, 2, (N, 1)) + np.cumsum(v * np.sin(th) * 0.1, 1)
    return x.astype(np.float32), y.astype(np.float32)


@pytest.mark.parametrize("P", [2, 37, 64, 65, 100, 130])
@pytest.mark.parametrize("flat", [True, False])
def test_batched_totals_match_the_oracle(P, flat):
    rng = np.random.default_rng(P + (7 if flat else 0))
    N, S, O = 900, 300, 600
    px, py = _paths(N, P, rng)
    seg, acc = syn.arc_segment(S, radius=5.0, spacing=0.03)
    if not flat:
        seg[:, 2] = (0.01 * np.arange(S)).astype(np.float32)
    obs = (rng.random((O, 3)) * 8 - 4).astype(np.float32)
    w = (0.7, 1.3, 2.0, 0.0, 0.0)
    ox, oy = ko.obstacles_from_points((0, 0, 0), (0, 0, 0, 1), STATE, obs)
    ci = ko.CostInputs(seg, 0, acc, 9.0, np.stack([ox, oy], 1), np.float32(10.0) / np.float32(3.0), (2.0, 0.0, 3.0),
                       ko.make_weights(*w))
    oi, oc, ocosts = ko.min_trajectory_cost(ci, px, py, None)
    for batch in (2, 0):
        ctx = kh.DwaContext(syn.CYLINDER, [0.1, 0.4], max_samples=N, max_points=P, max_segment=S, max_obstacles=O,
                            acc_limits=(2.0, 0.0, 3.0))
        ctx.set_option("cost_kernel", 2)
        ctx.set_option("cost_batch", batch)
        ctx.set_weights(kh.make_weights(*w))
        ctx.set_tracked_segment(seg, acc, 9.0)
        ctx.set_points(STATE, obs, 10.0)
        r, hcosts = ctx.cost_evaluate(px, py, None)
        np.testing.assert_array_equal(hcosts.view(np.uint32), ocosts.view(np.uint32), err_msg=f"cost_batch={batch}")
        assert r.found and r.index == oi and np.float32(r.cost) == np.float32(oc)
        ctx.close()


def test_precomputed_velocity_sums_and_deferred_sums():
    """Caller-provided velocity profiles: with the sums precomputed (velocity_group 4 / 16) or deferred to the finish
    kernel the batched kernel runs; with the sums formed by the wavefront itself (group 1) the plain one does."""
    rng = np.random.default_rng(5)
    N, P, S, O = 1500, 41, 200, 300
    px, py = _paths(N, P, rng)
    vel = [(rng.random((N, P - 1)) * 2 - 1).astype(np.float32) for _ in range(3)]
    seg, acc = syn.arc_segment(S, radius=4.0, spacing=0.02)
    obs = (rng.random((O, 3)) * 8 - 4).astype(np.float32)
    w = (0.7, 1.3, 2.0, 0.5, 0.25)
    ox, oy = ko.obstacles_from_points((0, 0, 0), (0, 0, 0, 1), STATE, obs)
    ci = ko.CostInputs(seg, 0, acc, 7.5, np.stack([ox, oy], 1), np.float32(10.0) / np.float32(3.0), (2.0, 0.0, 3.0),
                       ko.make_weights(*w))
    oi, oc, ocosts = ko.min_trajectory_cost(ci, px, py, vel)
    for group in (4, 16, 1):
        for beside in (0, 1):
            ctx = kh.DwaContext(syn.CYLINDER, [0.1, 0.4], max_samples=N, max_points=P, acc_limits=(2.0, 0.0, 3.0))
            ctx.set_option("cost_kernel", 2)
            ctx.set_option("cost_batch", 2)
            ctx.set_option("velocity_group", group)
            ctx.set_option("velocity_beside", beside)
            ctx.set_weights(kh.make_weights(*w))
            ctx.set_tracked_segment(seg, acc, 7.5)
            ctx.set_points(STATE, obs, 10.0)
            r, hcosts = ctx.cost_evaluate(px, py, vel)
            np.testing.assert_array_equal(hcosts.view(np.uint32), ocosts.view(np.uint32), err_msg=f"group={group} beside={beside}")
            assert r.found and r.index == oi
            ctx.close()


def test_many_buffers_per_workgroup():
    """40 000 samples over 256 workgroups: 157 slots each, three groups -- the third reuses the first buffer while the
    second may still be filling (the hand-off the kernel's two counters guard)."""
    rng = np.random.default_rng(9)
    N, P, S, O = 40000, 24, 120, 200
    px, py = _paths(N, P, rng)
    seg, acc = syn.straight_segment(S, 0.03)
    obs = (rng.random((O, 3)) * 8 - 4).astype(np.float32)
    w = (1.0, 1.0, 1.0, 0.0, 0.0)
    ox, oy = ko.obstacles_from_points((0, 0, 0), (0, 0, 0, 1), STATE, obs)
    ci = ko.CostInputs(seg, 0, acc, 3.6, np.stack([ox, oy], 1), np.float32(10.0) / np.float32(3.0), (2.0, 0.0, 3.0),
                       ko.make_weights(*w))
    oi, oc, ocosts = ko.min_trajectory_cost(ci, px, py, None)
    ctx = kh.DwaContext(syn.CYLINDER, [0.1, 0.4], max_samples=N, max_points=P, max_segment=S, max_obstacles=O,
                        acc_limits=(2.0, 0.0, 3.0))
    ctx.set_weights(kh.make_weights(*w))
    ctx.set_tracked_segment(seg, acc, 3.6)
    ctx.set_points(STATE, obs, 10.0)
    assert ctx.get_option("cost_batch") == 1.0          # the default rule: long lists take the batched kernel
    for _ in range(3):                                  # (and again: the counters are re-armed per launch)
        r, hcosts = ctx.cost_evaluate(px, py, None)
        np.testing.assert_array_equal(hcosts.view(np.uint32), ocosts.view(np.uint32))
        assert r.found and r.index == oi and np.float32(r.cost) == np.float32(oc)
    ctx.close()


@pytest.mark.parametrize("scene", ["mid", "open"])
def test_three_kernel_cycle_with_the_batched_kernel(scene):
    inp = syn.make_controller_inputs("cfg2", seed=2, scale=0.5, scene=scene)
    o = oracle_cycle_mt(inp)
    ctx = hip_context(kh, inp)
    for k, v in dict(fused_cycle=0, cost_kernel=2, cost_batch=2).items():
        ctx.set_option(k, v)
    h = hip_cycle(kh, inp, ctx=ctx)
    assert h["res"]["n_admissible"] == len(o["raw"])
    np.testing.assert_array_equal(h["raw"], o["raw"])
    np.testing.assert_array_equal(h["costs"].view(np.uint32), o["costs"].view(np.uint32))
    assert h["res"]["index"] == o["index"]
    ctx.close()


@pytest.mark.parametrize("P,N", [(50, 1000), (50, 64 * 7 + 1), (100, 777), (20, 2111), (7, 3000)])
@pytest.mark.parametrize("obstacles", ["few", "cluster", "dense"])
def test_batched_kernel_obstacle_paths(P, N, obstacles):
    """Every branch of the obstacle term under the batched kernel: a handful of far obstacles (skip values beyond the
    union scan: ring walks + the cooperative pass), one cluster (near for some samples, far for others), dense clutter
    (the union rectangle); list lengths with a short last group."""
    rng = np.random.default_rng(P * 1000 + N)
    S = 160
    px, py = _paths(N, P, rng)
    seg, acc = syn.arc_segment(S, radius=6.0, spacing=0.04)
    if obstacles == "few":
        obs = (rng.random((5, 3)) * 14 - 7).astype(np.float32)
    elif obstacles == "cluster":
        obs = (np.float32([1.5, -1.0, 0.0]) + 0.3 * rng.standard_normal((400, 3))).astype(np.float32)
    else:
        obs = (rng.random((3000, 3)) * 10 - 5).astype(np.float32)
    w = (0.5, 1.0, 3.0, 0.0, 0.0)
    ox, oy = ko.obstacles_from_points((0, 0, 0), (0, 0, 0, 1), STATE, obs)
    ci = ko.CostInputs(seg, 0, acc, 6.4, np.stack([ox, oy], 1), np.float32(10.0) / np.float32(3.0), (2.0, 0.0, 3.0),
                       ko.make_weights(*w))
    oi, oc, ocosts = ko.min_trajectory_cost(ci, px, py, None)
    for batch in (2, 0):
        ctx = kh.DwaContext(syn.CYLINDER, [0.1, 0.4], max_samples=N, max_points=P, max_segment=S, max_obstacles=len(obs),
                            acc_limits=(2.0, 0.0, 3.0))
        ctx.set_option("cost_kernel", 2)
        ctx.set_option("cost_batch", batch)
        ctx.set_weights(kh.make_weights(*w))
        ctx.set_tracked_segment(seg, acc, 6.4)
        ctx.set_points(STATE, obs, 10.0)
        for _ in range(2):
            r, hcosts = ctx.cost_evaluate(px, py, None)
            np.testing.assert_array_equal(hcosts.view(np.uint32), ocosts.view(np.uint32), err_msg=f"cost_batch={batch}")
            assert r.found and r.index == oi and np.float32(r.cost) == np.float32(oc)
        ctx.close()


@pytest.mark.parametrize("beams", [360, 1440])
def test_batched_kernel_with_a_laserscan(beams):
    """The scan's near table under the batched kernel (three-kernel cycle, batched kernel forced / off)."""
    from helpers import oracle_cycle
    inp = syn.make_controller_inputs("cfg2", seed=4, scale=0.3, scene="open")
    ang = np.linspace(-np.pi, np.pi, beams, endpoint=False)
    rng = 3.0 + 1.2 * np.cos(5 * ang) + 0.3 * np.sin(17 * ang)
    cur = dict(inp, state=(0.4, -0.3, 0.5, 0.0))
    o = oracle_cycle(cur, scan=(rng, ang))
    assert len(o["raw"]) > 100
    for batch in (2, 0):
        ctx = hip_context(kh, cur)
        for k, v in dict(fused_cycle=0, cost_kernel=2, cost_batch=batch).items():
            ctx.set_option(k, v)
        assert_cycle_equal(o, hip_cycle(kh, cur, scan=(rng, ang), ctx=ctx))
        ctx.close()
