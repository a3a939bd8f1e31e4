"""Seeded random scenes through the whole device path against the oracle, bit for
bit: varied obstacle geometry (clutter, rings, far clusters, nothing nearby),
robot shapes, sensor mounts, poses, tracked segments with curvature and height,
weights, horizons over one and two point tiles, point-list and laserscan
updates, several cycles per context (so both cost kernels and the lazy host
lists come into play)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import kompass_hip as kh  # noqa: E402
import synthetic as syn  # noqa: E402

from helpers import assert_cycle_equal, hip_context, hip_cycle, oracle_cycle  # noqa: E402


def _segment(rng, kind, n):
    s = np.arange(n) * rng.choice([0.005, 0.01, 0.02])
    if kind == "straight":
        xyz = np.stack([s, np.full_like(s, rng.uniform(-0.3, 0.3)), np.zeros_like(s)], 1)
    elif kind == "arc":
        R = rng.uniform(2.0, 12.0) * rng.choice([-1, 1])
        xyz = np.stack([R * np.sin(s / R), R * (1 - np.cos(s / R)), np.zeros_like(s)], 1)
    else:  # wavy, with height
        xyz = np.stack([s, 0.4 * np.sin(1.3 * s), 0.2 * np.cos(0.7 * s)], 1)
    return xyz.astype(np.float32), s.astype(np.float32)


def _obstacles(rng, kind):
    if kind == "clutter":
        n = rng.integers(300, 3000)
        p = rng.uniform(-8, 8, (n, 2))
        p = p[np.hypot(p[:, 0], p[:, 1]) > rng.uniform(0.6, 1.5)]
    elif kind == "ring":
        n = rng.integers(200, 2000)
        th = rng.uniform(0, 2 * np.pi, n)
        r = rng.uniform(1.5, 5.0) + 0.5 * np.sin(7 * th)
        p = np.stack([r * np.cos(th), r * np.sin(th)], 1)
    elif kind == "far":
        n = rng.integers(50, 600)
        c = rng.uniform(-9, 9, 2)
        c = c / max(np.hypot(*c), 1e-6) * rng.uniform(4.0, 9.0)
        p = c + rng.normal(0, 0.5, (n, 2))
    elif kind == "wall":
        n = rng.integers(100, 800)
        p = np.stack([np.full(n, rng.uniform(1.0, 3.0)), rng.uniform(-4, 4, n)], 1)
    else:  # a handful of stragglers
        p = rng.uniform(-10, 10, (rng.integers(1, 12), 2))
    z = rng.uniform(0.0, 0.6, len(p))
    return np.column_stack([p, z]).astype(np.float32)


SCENES = [(kind, seg, shape) for kind in ("clutter", "ring", "far", "wall", "few")
          for seg in ("straight", "arc", "wavy") for shape in (0, 1)]


# KC_FUZZ_SEEDS="2000,3000,...": more seed bases for a longer campaign (same process)
_SEEDS = [1000] + [int(s) for s in os.environ.get("KC_FUZZ_SEEDS", "").split(",") if s.strip()]


@pytest.mark.parametrize("seed", _SEEDS)
@pytest.mark.parametrize("case", range(len(SCENES)))
def test_random_scene(case, seed):
    kind, segk, shape = SCENES[case]
    rng = np.random.default_rng(seed + case)
    # every fifth scene has > 512 samples, so that long admissible lists (the
    # wavefront-per-sample cost kernel beyond the first cycle) occur as well
    inp = syn.make_controller_inputs("cfg1", seed=case + (seed - 1000), scale=3.0 if case % 5 == 0 else rng.choice([0.6, 1.0, 1.6]))
    inp["P"] = int(rng.choice([12, 33, 64, 65, 90]))
    inp["robot"] = (dict(shape=syn.CYLINDER, dims=[float(rng.uniform(0.08, 0.4)), 0.5]) if shape == 0
                    else dict(shape=syn.BOX, dims=[float(rng.uniform(0.2, 0.7)), float(rng.uniform(0.15, 0.5)), 0.5]))
    inp["octree_res"] = float(rng.choice([0.05, 0.1, 0.2]))
    seg, acc = _segment(rng, segk, int(rng.choice([40, 201, 700, 1500])))
    inp["seg_xyz"], inp["acc_at_seg"] = seg, acc
    inp["ref_len"] = float(acc[-1]) + float(rng.uniform(0.0, 3.0))
    inp["max_range"] = float(rng.choice([4.0, 10.0, 25.0]))
    inp["weights"] = tuple(float(w) for w in rng.choice([0.0, 0.5, 1.0, 2.0], 3)) + (0.0, 0.0)
    if sum(inp["weights"]) == 0.0:
        inp["weights"] = (1.0, 1.0, 1.0, 0.0, 0.0)
    sensor_pos = (float(rng.uniform(-0.2, 0.2)), float(rng.uniform(-0.1, 0.1)), float(rng.uniform(0.0, 0.3)))
    yaw = rng.uniform(-3.0, 3.0) if case % 3 == 0 else 0.0
    sensor_rot = (0.0, 0.0, float(np.sin(yaw / 2)), float(np.cos(yaw / 2)))
    ctx = hip_context(kh, inp, sensor_pos, sensor_rot, max_points=inp["P"])
    ctx_obst = 0
    for cycle in range(3):
        inp["state"] = (float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-3.1, 3.1)), 0.3)
        kinds = kind if cycle != 1 else str(rng.choice(["clutter", "ring", "far", "wall", "few"]))
        if (case + cycle) % 4 == 3:  # laserscan update
            n = int(rng.choice([90, 360, 1081]))
            ang = np.linspace(-np.pi, np.pi, n, endpoint=False)
            rngs = rng.uniform(0.4, 9.0, n)
            rngs[rng.integers(0, n, 3)] = np.inf
            scan = (rngs, ang)
            o = oracle_cycle(inp, scan, sensor_pos, sensor_rot)
            h = hip_cycle(kh, inp, scan, sensor_pos, sensor_rot, ctx=ctx)
        else:
            inp["points"] = _obstacles(rng, kinds)
            if len(inp["points"]) > ctx_obst:  # contexts are sized once; keep within the first size
                if ctx_obst == 0:
                    ctx.close()
                    ctx_obst = max(len(inp["points"]), 4096)
                    big = dict(inp, points=np.zeros((ctx_obst, 3), np.float32))
                    ctx = hip_context(kh, big, sensor_pos, sensor_rot, max_points=inp["P"])
                else:
                    inp["points"] = inp["points"][:ctx_obst]
            o = oracle_cycle(inp, None, sensor_pos, sensor_rot)
            h = hip_cycle(kh, inp, None, sensor_pos, sensor_rot, ctx=ctx)
        assert_cycle_equal(o, h)
    ctx.close()


@pytest.mark.parametrize("segk", ["straight", "arc", "wavy"])
def test_tracked_window_of_a_resident_path(segk):
    """kc_dwa_set_path + kc_dwa_set_tracked_window (tables built on the device)
    == the oracle on the same window, for windows of many sizes and positions,
    growing and shrinking, on one context."""
    rng = np.random.default_rng(99)
    inp = syn.make_controller_inputs("cfg1", seed=7, scale=2.0)
    inp["P"] = 40
    inp["points"] = _obstacles(rng, "clutter")
    path, acc = _segment(rng, segk, 3000)
    total = float(acc[-1]) + 0.75
    ctx = hip_context(kh, dict(inp, seg_xyz=path[:16]), max_points=inp["P"])   # small table first: it has to grow
    ctx.set_weights(kh.make_weights(*inp["weights"]))
    ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
    ctx.set_path(path, acc, total)
    windows = [(0, 40), (0, 201), (1234, 700), (1500, 1500), (2999, 1), (2990, 10), (100, 17), (0, 3000), (777, 64)]
    for k, (start, size) in enumerate(windows):
        inp["state"] = (float(path[start, 0]) + float(rng.uniform(-0.3, 0.3)),
                        float(path[start, 1]) + float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-3.1, 3.1)), 0.3)
        inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"] = path[start:start + size], acc[start:start + size], total
        o = oracle_cycle(inp)
        ctx.set_points(inp["state"], inp["points"], inp["max_range"])
        ctx.set_tracked_window(start, size)
        res = ctx.cycle(inp["state"], inp["P"])
        px, py, raw, costs = ctx.get_samples(with_costs=True)
        h = dict(px=px.copy(), py=py.copy(), raw=raw.copy(), costs=costs.copy(), res=res.as_dict())
        if res.found:
            h["best"] = ctx.get_best()
        assert_cycle_equal(o, h)
        # and the host-built tables of the same window give the same costs
        ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
        ctx.cycle(inp["state"], inp["P"])
        _, _, _, costs2 = ctx.get_samples(with_costs=True)
        np.testing.assert_array_equal(costs2.view(np.uint32), costs.view(np.uint32))
    with pytest.raises((ValueError, IndexError, RuntimeError)):
        ctx.set_tracked_window(2990, 20)
    ctx.close()
    # no resident path yet: any non-empty window is out of range; an empty path is accepted
    ctx2 = hip_context(kh, inp, max_points=inp["P"])
    with pytest.raises((ValueError, IndexError, RuntimeError)):
        ctx2.set_tracked_window(0, 1)
    ctx2.set_path(np.zeros((0, 3), np.float32), np.zeros(0, np.float32), 0.0)
    ctx2.set_tracked_window(0, 0)
    ctx2.close()
