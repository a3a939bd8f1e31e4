"""The union scan of the obstacle term (obstacle_union_scan, kc_cost_kernels.h; option "obs_union"): where a
trajectory runs through occupied bucket cells the wavefront takes a seed bound, the one rectangle of cells that holds
every obstacle within that bound of any point that can still lower it, and broadcasts those obstacles to all lanes.
A pruned search: it must return the SAME minimum as the reference's double loop (trajectory.h:218-235) -- every cost
bit-equal to the oracle, for every threshold (0 = ring walks only, small = mostly fallbacks, huge = always), over
clutter of all densities, clusters, walls, trajectories that leave the obstacles' bounding box, two-tile
trajectories (P > 64) and non-finite trajectory points."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import kompass_hip as kh  # noqa: E402
import synthetic as syn  # noqa: E402
from oracle import ko  # noqa: E402

SEEDS = [int(s) for s in __import__("os").environ.get("KC_FUZZ_SEEDS", "0").split(",")]
STATE = (0.3, -0.2, 0.4, 0.0)


def obstacles(kind, rng):
    if kind == "sparse":
        xy = rng.uniform(-6, 6, (60, 2))
    elif kind == "mid":
        xy = rng.uniform(-6, 6, (800, 2))
    elif kind == "dense":
        xy = rng.uniform(-6, 6, (6000, 2))
    elif kind == "clusters":
        c = rng.uniform(-5, 5, (12, 2))
        xy = c[rng.integers(0, 12, 1500)] + rng.normal(0, 0.15, (1500, 2))
    elif kind == "walls":      # solid lines of 5 cm cells, like a costmap border / a scanned room
        t = np.arange(-6, 6, 0.05)
        xy = np.concatenate([np.stack([t, np.full_like(t, -1.5)], 1), np.stack([t, np.full_like(t, 2.0)], 1),
                             np.stack([np.full_like(t, 4.0), t], 1), rng.uniform(-6, 6, (150, 2))])
    else:                      # "far": everything at least 2.5 m from the trajectories' region
        xy = rng.uniform(-6, 6, (3000, 2))
        xy = xy[np.hypot(xy[:, 0], xy[:, 1]) > 4.5]
    return np.concatenate([xy, np.zeros((len(xy), 1))], 1).astype(np.float32)


def trajectories(N, P, rng, spread):
    """Smooth roll-out-like curves from around the origin: speed 0-1 m/s, turn rate +-1 rad/s, 0.1 s steps."""
    v = rng.uniform(0.0, 1.0, (N, 1))
    w = rng.uniform(-1.0, 1.0, (N, 1))
    th0 = rng.uniform(-np.pi, np.pi, (N, 1))
    t = 0.1 * np.arange(P)[None]
    th = th0 + w * t
    x = rng.uniform(-spread, spread, (N, 1)) + np.cumsum(v * np.cos(th) * 0.1, 1)
    y = rng.uniform(-spread, spread, (N, 1)) + np.cumsum(v * np.sin(th) * 0.1, 1)
    return x.astype(np.float32), y.astype(np.float32)


def run(obs, px, py, w, unions, max_range=10.0):
    ox, oy = ko.obstacles_from_points((0, 0, 0), (0, 0, 0, 1), STATE, obs)
    seg, acc = syn.straight_segment(200, 0.02)
    ci = ko.CostInputs(seg, 0, acc, 4.0, np.stack([ox, oy], 1), np.float32(max_range) / np.float32(3.0),
                       (2.0, 0.0, 3.0), ko.make_weights(*w))
    oi, oc, ocosts = ko.min_trajectory_cost(ci, px, py, None)
    N, P = px.shape
    for k, u in enumerate(unions):
        ctx = kh.DwaContext(syn.CYLINDER, [0.1, 0.4], max_samples=N, max_points=P, max_obstacles=len(obs),
                            acc_limits=(2.0, 0.0, 3.0))
        ctx.set_option("cost_kernel", 2)   # the wavefront-per-sample kernel for every list length
        ctx.set_option("cost_batch", 2 * (k & 1))   # ... with and without the batched per-sample part
        ctx.set_option("obs_union", u)
        ctx.set_weights(kh.make_weights(*w))
        ctx.set_tracked_segment(seg, acc, 4.0)
        ctx.set_points(STATE, obs, max_range)
        r, hcosts = ctx.cost_evaluate(px, py, None)
        np.testing.assert_array_equal(hcosts.view(np.uint32), ocosts.view(np.uint32), err_msg=f"obs_union={u}")
        assert r.index == oi
        ctx.close()


@pytest.mark.parametrize("seed", SEEDS)
@pytest.mark.parametrize("kind", ["sparse", "mid", "dense", "clusters", "walls", "far"])
@pytest.mark.parametrize("P", [20, 50, 100])
def test_union_scan_matches_the_oracle(kind, P, seed):
    rng = np.random.default_rng(100 * seed + P + len(kind))
    obs = obstacles(kind, rng)
    px, py = trajectories(160, P, rng, spread=3.0)
    run(obs, px, py, (1.0, 1.0, 1.0, 0.0, 0.0), (0, 4, 24, 96, 4096))


@pytest.mark.parametrize("seed", SEEDS)
def test_trajectories_outside_the_bucket_grid(seed):
    """Points beyond the obstacles' bounding box (clamped cells, the distance of the point to the grid shrinks every
    guarantee) and trajectories that cross it."""
    rng = np.random.default_rng(7 + seed)
    obs = obstacles("mid", rng) * np.float32([0.3, 0.3, 1.0])   # clutter within +-1.8 m
    px, py = trajectories(200, 50, rng, spread=4.0)              # starts up to 4 m out
    run(obs, px, py, (1.0, 1.0, 1.0, 0.0, 0.0), (0, 24, 4096))
    run(obs, px + np.float32(30.0), py, (0.0, 0.0, 1.0, 0.0, 0.0), (0, 24, 4096))   # nothing within max_obstacles_dist
    run(obs, px, py, (0.0, 0.0, 1.0, 0.0, 0.0), (0, 24, 4096), max_range=1.0)       # a cap of a third of a metre


@pytest.mark.parametrize("seed", SEEDS)
def test_non_finite_points_never_win(seed):
    """`dist < minDist` is false for NaN and for +inf against DBL_MAX: such points leave the minimum to the others
    (obstacle weight only: the other terms of such a trajectory are NaN in the reference as well)."""
    rng = np.random.default_rng(11 + seed)
    obs = obstacles("mid", rng)
    px, py = trajectories(150, 50, rng, spread=3.0)
    bad = rng.integers(0, 50, 150)
    for i in range(0, 150, 3):
        px[i, bad[i]] = [np.nan, np.inf, -np.inf][(i // 3) % 3]
    for i in range(1, 150, 7):
        py[i, :] = np.nan                                        # a whole trajectory without a finite point
    run(obs, px, py, (0.0, 0.0, 1.0, 0.0, 0.0), (0, 24, 4096))
