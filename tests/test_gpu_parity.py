"""GPU parity tests: the HIP path through the C ABI vs the CPU oracle on the
same seeded inputs.  Bit-exact on admissible set, float paths, float costs and
the selected index (tolerance 0; north_star allows 1e-5 on costs).

Run with `pytest -m gpu` on an MI355X.
"""
import json
import math
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import kompass_hip as kh  # noqa: E402
import synthetic as syn  # noqa: E402
from oracle import ko  # noqa: E402

from helpers import assert_cycle_equal, hip_context, hip_cycle, oracle_cycle  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert kh.device_count() >= 1, "no HIP device visible: the -m gpu tests need an MI355X"


# ---------------------------------------------------------------------------
# controller: full cycle on scaled-down BASELINE configs
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("name,scale", [("cfg1", 1.0), ("cfg2", 0.25), ("cfg3", 0.08), ("cfg5", 0.08)])
def test_cycle_parity_configs(name, scale):
    inp = syn.make_controller_inputs(name, seed=1, scale=scale)
    o = oracle_cycle(inp)
    h = hip_cycle(kh, inp)
    assert len(o["raw"]) > 0, "scenario must leave admissible samples"
    assert len(o["raw"]) < len(inp["vx"]), "scenario must drop colliding samples"
    assert_cycle_equal(o, h)


@pytest.mark.parametrize("shape,dims", [(syn.CYLINDER, [0.2, 0.5]), (syn.BOX, [0.5, 0.3, 0.4]),
                                        (syn.SPHERE, [0.25])])
@pytest.mark.parametrize("yaw", [0.0, 0.7, -2.3])
def test_cycle_parity_shapes_points(shape, dims, yaw):
    inp = syn.make_controller_inputs("cfg1", seed=3)
    inp["robot"] = dict(shape=shape, dims=dims)
    inp["state"] = (0.3, -0.2, yaw, 0.0)
    # points at several heights: exercises the z-interval / sphere z-gap logic
    rng = np.random.default_rng(7)
    pts = inp["points"].copy()
    pts[:, 2] = rng.choice([-0.3, -0.05, 0.0, 0.12, 0.31, 0.6], size=len(pts)).astype(np.float32)
    inp["points"] = pts
    o = oracle_cycle(inp)
    h = hip_cycle(kh, inp)
    assert_cycle_equal(o, h)


@pytest.mark.parametrize("shape,dims", [(syn.CYLINDER, [0.15, 0.4]), (syn.BOX, [0.4, 0.4, 1.0])])
def test_cycle_parity_laserscan_sensor_frame(shape, dims):
    """LaserScan input: octree lives in the (rotated, offset) sensor frame."""
    inp = syn.make_controller_inputs("cfg1", seed=5)
    inp["robot"] = dict(shape=shape, dims=dims)
    inp["state"] = (1.0, 2.0, 0.4, 0.0)
    inp["seg_xyz"] = inp["seg_xyz"] + np.float32([1.0, 2.0, 0.0])
    ang = np.linspace(0, 2 * math.pi, 360, endpoint=False)
    rng = np.random.default_rng(11)
    ranges = 1.2 + 1.5 * rng.random(360)
    ranges[5] = np.inf  # filtered by the collision path only (Q7 excluded: finite elsewhere)
    ranges[5] = 3.0
    spos = (0.1, -0.05, 0.3)
    half = 0.35  # sensor yawed by 0.7 rad
    srot = (0.0, 0.0, math.sin(half), math.cos(half))
    o = oracle_cycle(inp, scan=(ranges, ang), sensor_pos=spos, sensor_rot=srot)
    h = hip_cycle(kh, inp, scan=(ranges, ang), sensor_pos=spos, sensor_rot=srot)
    assert 0 < len(o["raw"]) < len(inp["vx"])
    assert_cycle_equal(o, h)


def test_cycle_no_obstacles_and_all_blocked():
    inp = syn.make_controller_inputs("cfg1", seed=2)
    inp["points"] = np.zeros((0, 3), np.float32)
    o = oracle_cycle(inp)
    h = hip_cycle(kh, inp)
    assert len(o["raw"]) == len(inp["vx"])
    assert_cycle_equal(o, h)
    # a wall of points right in front of the robot: nothing admissible
    ys = np.arange(-3, 3, 0.02)
    wall = np.stack([np.full_like(ys, 0.12), ys, np.zeros_like(ys)], axis=1).astype(np.float32)
    inp["points"] = wall
    o = oracle_cycle(inp)
    h = hip_cycle(kh, inp)
    assert len(o["raw"]) == 0 and not h["res"]["found"] and h["res"]["index"] == -1
    assert_cycle_equal(o, h)


def test_sample_window_matches_oracle():
    """A1: host lattice (kc_dwa_sample_window) == oracle list, all robot types."""
    lim_o = ko.make_limits(**syn.LIMITS)
    lim_h = kh.make_limits(**syn.LIMITS)
    for ctr in (syn.ACKERMANN, syn.DIFFERENTIAL_DRIVE, syn.OMNI):
        for cur in [(0.5, 0.0, 0.0), (0.0, 0.0, 0.0), (-0.3, 0.2, 1.9), (1.0, -1.0, -2.0)]:
            for L, A in [(4, 4), (11, 11), (20, 7)]:
                ovx, ovy, oom = ko.sample_velocities(ctr, lim_o, cur, 0.1, L, A)
                ctx = kh.DwaContext(syn.CYLINDER, [0.1, 0.4], max_samples=4096, max_points=8)
                hvx, hvy, hom = ctx.sample_window(ctr, lim_h, cur, L, A)
                np.testing.assert_array_equal(hvx, ovx)
                np.testing.assert_array_equal(hvy, ovy)
                np.testing.assert_array_equal(hom, oom)
                ctx.close()


# ---------------------------------------------------------------------------
# cost evaluator: the reference's closed-form cases through kc_cost_evaluate
# ---------------------------------------------------------------------------
GOLD = json.loads((Path(__file__).parent / "golden" / "cost_kat.json").read_text())


def _kat_eval(weights, path_pts, vels=None, obstacles=None):
    ref = ko.Path([[0, 0, 0], [10.0, 0, 0]])
    ref.interpolate(1.0)
    ref.segment(5.0, 10000)
    s0, s1 = ref.segment_range(0)
    seg = np.stack([ref.x[s0:s1 + 1], ref.y[s0:s1 + 1], ref.z[s0:s1 + 1]], axis=1)
    acc = ref.acc[s0:s1 + 1]
    ctx = kh.DwaContext(syn.CYLINDER, [0.1, 0.4], max_samples=4, max_points=8, acc_limits=(1, 1, 1))
    ctx.set_weights(kh.make_weights(**weights))
    ctx.set_tracked_segment(seg, acc, ref.total_length)
    if obstacles:
        ctx.set_points((0, 0, 0, 0), np.float32(obstacles), 30.0)
    pts = np.float32(path_pts).reshape(-1, 3)
    n = len(pts)
    v = np.float32(vels if vels is not None else [(0, 0, 0)] * (n - 1)).reshape(n - 1, 3)
    r, costs = ctx.cost_evaluate(pts[:, 0][None], pts[:, 1][None], [v[:, 0][None], v[:, 1][None], v[:, 2][None]])
    assert r.found and r.index == 0
    ctx.close()
    return float(costs[0])


def _solo(name):
    w = dict(path=0.0, goal=0.0, obstacles=0.0, smoothness=0.0, jerk=0.0)
    w[name] = 1.0
    return w


def test_cost_known_answers_on_gpu():
    z5 = [(0.0, 0.0, 0.0)] * 5
    got = {
        "goal_cost_on_straight_path": [_kat_eval(_solo("goal"), [(4.0, 0, 0)] * 5)],
        "goal_cost_tie_breaker": [_kat_eval(_solo("goal"), [(4.0, 0.1, 0)] * 5),
                                  _kat_eval(_solo("goal"), [(4.0, 0.5, 0)] * 5)],
        "path_cost_centered_sample": [_kat_eval(_solo("path"), [(float(i), 0, 0) for i in range(5)])],
        "path_cost_constant_lateral_offset": [_kat_eval(_solo("path"), [(float(i), 0.5, 0) for i in range(5)])],
        "smoothness_cost_constant_velocity": [_kat_eval(_solo("smoothness"), z5, [(1.0, 0, 0)] * 4)],
        "smoothness_cost_single_step_change": [
            _kat_eval(_solo("smoothness"), z5, [(0, 0, 0), (1, 0, 0), (1, 0, 0), (1, 0, 0)])],
        "jerk_cost_known_second_diff": [_kat_eval(_solo("jerk"), z5, [(0, 0, 0), (1, 0, 0), (3, 0, 0), (6, 0, 0)])],
        "obstacles_cost_at_max_range": [_kat_eval(_solo("obstacles"), z5, obstacles=[(20.0, 0, 0)])],
        "obstacles_cost_at_zero_distance": [_kat_eval(_solo("obstacles"), z5, obstacles=[(0.0, 0, 0)])],
        "obstacles_cost_at_half_range": [_kat_eval(_solo("obstacles"), z5, obstacles=[(5.0, 0, 0)])],
    }
    for name, vals in got.items():
        g = GOLD["cost"][name]
        for v, e in zip(vals, g["expected"]):
            if e == 0.0:
                assert abs(v) <= 1e-12, (name, v)
            else:
                assert abs(v - e) <= g["tol"] * min(abs(v), abs(e)), (name, v, e)


def test_cost_evaluate_random_with_velocities():
    """Generic evaluator (arbitrary paths + velocity profiles, all 5 weights)."""
    rng = np.random.default_rng(21)
    N, P, S, O = 300, 37, 260, 500
    px = (rng.random((N, P)) * 6 - 1).astype(np.float32)
    py = (rng.random((N, P)) * 4 - 2).astype(np.float32)
    vel = [(rng.random((N, P - 1)) * 2 - 1).astype(np.float32) for _ in range(3)]
    seg, acc = syn.arc_segment(S, radius=4.0, spacing=0.02)
    seg[:, 2] = (0.01 * np.arange(S)).astype(np.float32)  # non-zero z path
    obs = (rng.random((O, 3)) * 8 - 3).astype(np.float32)
    state = (0.2, -0.1, 0.3, 0.0)
    w = (0.7, 1.3, 2.0, 0.5, 0.25)
    ox, oy = ko.obstacles_from_points((0, 0, 0), (0, 0, 0, 1), state, obs)
    ci = ko.CostInputs(seg, 40, np.concatenate([np.zeros(40, np.float32), acc]), 7.5,
                       np.stack([ox, oy], 1), np.float32(10.0) / np.float32(3.0), (2.0, 0.0, 3.0),
                       ko.make_weights(*w))
    oi, oc, ocosts = ko.min_trajectory_cost(ci, px, py, vel)
    ctx = kh.DwaContext(syn.CYLINDER, [0.1, 0.4], max_samples=N, max_points=P, acc_limits=(2.0, 0.0, 3.0))
    ctx.set_weights(kh.make_weights(*w))
    ctx.set_tracked_segment(seg, acc, 7.5)
    ctx.set_points(state, obs, 10.0)
    r, hcosts = ctx.cost_evaluate(px, py, vel)
    np.testing.assert_array_equal(hcosts.view(np.uint32), ocosts.view(np.uint32))
    assert r.found and r.index == oi and np.float32(r.cost) == np.float32(oc)
    # the velocity sums by the pass of their own (N and P - 1 multiples of neither 4 nor 16, an axis without a limit)
    for group in (4, 16, 1):
        ctx.set_option("velocity_group", group)
        r, hcosts = ctx.cost_evaluate(px, py, vel)
        np.testing.assert_array_equal(hcosts.view(np.uint32), ocosts.view(np.uint32))
        assert r.found and r.index == oi
    # lowest-index tie-break: duplicate the winner in front of itself
    px2 = np.concatenate([px[oi:oi + 1], px]); py2 = np.concatenate([py[oi:oi + 1], py])
    vel2 = [np.concatenate([v[oi:oi + 1], v]) for v in vel]
    ctx2 = kh.DwaContext(syn.CYLINDER, [0.1, 0.4], max_samples=N + 1, max_points=P, acc_limits=(2.0, 0.0, 3.0))
    ctx2.set_weights(kh.make_weights(*w))
    ctx2.set_tracked_segment(seg, acc, 7.5)
    ctx2.set_points(state, obs, 10.0)
    r2, _ = ctx2.cost_evaluate(px2, py2, vel2)
    assert r2.index == 0


# ---------------------------------------------------------------------------
# mapper
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("H,W,res,pos,orient,n,scale", [
    (400, 400, 0.05, (0, 0, 0), 0.0, 3600, 1.0),
    (200, 300, 0.1, (0.35, -0.2, 0.1), 0.6, 777, 1.0),
    (101, 77, 0.07, (-0.5, 0.4, 0), -2.0, 360, 0.5),
    (1000, 1000, 0.05, (0, 0, 0), 0.0, 4096, 4.0),
    (40, 50, 0.1, (0.3, 0.2, 0), 0.3, 64, 0.3),              # smaller than one tile row
    (640, 160, 0.05, (-14.0, 3.0, 0), 1.0, 1500, 2.5),       # sensor near a corner, tile-aligned grid
    (333, 517, 0.02, (1.0, -2.0, 0), 2.2, 2048, 1.0),        # long lines, ragged tiles
])
@pytest.mark.parametrize("tiles", ["2", "1"])
def test_mapper_parity(H, W, res, pos, orient, n, scale, tiles, monkeypatch):
    monkeypatch.setenv("KC_MAPPER_TILES", tiles)   # 2: tiled scan, 1: three beam-parallel passes (default for plain scans)
    ang, rng = syn.dense_scan(n, scale)
    r = np.random.default_rng(5)
    rng = rng * (0.6 + 0.8 * r.random(n))
    rng[::17] = 0.0          # zero-length rays
    rng[::29] *= 10.0        # rays leaving the grid
    want = ko.scan_to_grid(H, W, res, pos, orient, ang, rng)
    m = kh.MapperContext(H, W, res, pos, orient, n)
    got = m.scan_to_grid(ang, rng)
    assert got.shape == (H, W) and got.dtype == np.int32
    assert set(np.unique(got)) <= {-1, 0, 100}
    np.testing.assert_array_equal(got, want)
    # second call with new ranges, same angles (trig table reuse path)
    rng2 = rng[::-1].copy()
    np.testing.assert_array_equal(m.scan_to_grid(ang, rng2), ko.scan_to_grid(H, W, res, pos, orient, ang, rng2))
    m.close()


@pytest.mark.parametrize("staged", ["0", "1"])
def test_mapper_fixture_scan(staged, monkeypatch):
    """The reference's own 360-beam fixture (tests/resources/mapping/laserscan_data.json); staged = the copies a
    device without a large BAR gets (KC_MAPPER_STAGED, read when the context is made)."""
    monkeypatch.setenv("KC_MAPPER_STAGED", staged)
    data = json.loads((Path(__file__).parent / "golden" / "laserscan_data.json").read_text())
    rng = np.array(data["ranges"], dtype=np.float64)
    ang = data["angle_min"] + np.arange(len(rng)) * data["angle_increment"]
    rng = np.clip(np.nan_to_num(rng, posinf=20.0), 0, 20.0)
    want = ko.scan_to_grid(200, 200, 0.1, (0, 0, 0), 0.0, ang, rng)
    m = kh.MapperContext(200, 200, 0.1, (0, 0, 0), 0.0, len(rng))
    np.testing.assert_array_equal(m.scan_to_grid(ang, rng), want)
    assert (want == 100).sum() > 0


@pytest.mark.parametrize("tiles", ["1", "2"])
def test_mapper_random_scenes(tiles, monkeypatch):
    """Random geometry, sensor pose and ranges (some zero, some far beyond the grid), one beam to thousands;
    beam-parallel passes (default for plain scans) and the tiled scan."""
    monkeypatch.setenv("KC_MAPPER_TILES", tiles)
    r = np.random.default_rng(77 + int(tiles))
    for case in range(40):
        H, W = int(r.integers(3, 700)), int(r.integers(3, 700))
        res = float(r.choice([0.02, 0.05, 0.1, 0.25]))
        ext = min(H, W) * res
        pos = (float(r.uniform(-0.6, 0.6) * ext), float(r.uniform(-0.6, 0.6) * ext), 0.0)   # may lie outside the grid
        orient = float(r.uniform(-3.2, 3.2))
        n = int(r.choice([1, 2, 17, 360, 1000, 3000]))
        ang = np.sort(r.uniform(-np.pi, np.pi, n)) if case % 3 else r.uniform(-7, 7, n)     # unsorted angles too
        rng = r.uniform(0, 1.2 * ext, n) * r.choice([1.0, 1.0, 1.0, 0.0, 12.0], n)
        want = ko.scan_to_grid(H, W, res, pos, orient, ang, rng)
        m = kh.MapperContext(H, W, res, pos, orient, n)
        got = m.scan_to_grid(ang, rng)
        assert np.array_equal(got, want), (case, H, W, res, pos, orient, n, int((got != want).sum()))
        m.close()


def test_mapper_empty_scan():
    m = kh.MapperContext(50, 60, 0.1)
    g = m.scan_to_grid(np.zeros(0), np.zeros(0))
    assert (g == -1).all()


# ---------------------------------------------------------------------------
# device arithmetic self-check: correctly rounded f32 div/sqrt
# ---------------------------------------------------------------------------
def test_full_size_cfg2_properties():
    """cfg2 at BASELINE size: properties that need no oracle run -- winner is
    admissible, its cost is the minimum of the per-sample costs, index is the
    first minimum, shard union == unsharded."""
    inp = syn.make_controller_inputs("cfg2", seed=0)
    h = hip_cycle(kh, inp)
    res = h["res"]
    assert res["n_samples"] == 8192 and 0 < res["n_admissible"] < 8192
    costs = h["costs"]
    assert res["found"] and res["index"] == int(np.argmin(costs))
    assert np.float32(res["cost"]) == costs.min()
    assert (np.diff(h["raw"]) > 0).all()
    # two shards: min over shard keys == unsharded key
    ctx = h["ctx"]
    keys, counts = [], []
    for first, count in [(0, 5000), (5000, 3192)]:
        ctx.set_shard(first, count)
        r = ctx.cycle(inp["state"], inp["P"])
        keys.append((np.float32(r.cost), r.raw_index) if r.found else (np.float32(np.inf), 1 << 40))
        counts.append(r.n_admissible)
    assert sum(counts) == res["n_admissible"]
    best = min(keys)
    assert best[1] == res["raw_index"] and best[0] == np.float32(res["cost"])
    # admissible samples never touch an occupied voxel: re-check 64 rows on CPU
    o = oracle_cycle(dict(inp, vx=inp["vx"][h["raw"][:64]], vy=inp["vy"][h["raw"][:64]],
                          omega=inp["omega"][h["raw"][:64]]))
    assert len(o["raw"]) == 64
    np.testing.assert_array_equal(o["px"], h["px"][:64])
    np.testing.assert_array_equal(o["costs"].view(np.uint32), costs[:64].view(np.uint32))


_PATH_SCENARIOS = [
    # (config, lattice scale, robot shape, dims, variant)
    ("cfg1", 1.0, syn.CYLINDER, [0.1, 0.4], None),
    ("cfg2", 0.25, syn.BOX, [0.3, 0.2, 0.4], None),
    ("cfg3", 0.04, syn.CYLINDER, [0.2, 0.4], None),       # P = 100: two point tiles per wavefront
    ("cfg2", 0.25, syn.CYLINDER, [0.1, 0.4], "open"),     # every sample admissible
    ("cfg2", 0.1, syn.CYLINDER, [0.1, 0.4], "longseg"),   # 1600-point segment: chunks of 25
    ("cfg2", 0.25, syn.CYLINDER, [0.1, 0.4], "ring"),     # obstacles on a closed curve 3-7 m away: far searches
    ("cfg3", 0.04, syn.CYLINDER, [0.1, 0.4], "ring"),     # ... with two point tiles per wavefront
]


def _path_scenario(name, scale, shape, dims, variant, seed=4):
    inp = syn.make_controller_inputs(name, seed=seed, scale=scale)
    inp["robot"] = dict(shape=shape, dims=dims)
    if variant == "open":
        pts = np.asarray(inp["points"], dtype=np.float32).reshape(-1, 3)
        inp["points"] = pts[np.hypot(pts[:, 0], pts[:, 1]) > 4.0]
    if variant == "ring":
        ang, rng = syn.dense_scan(1500, 1.0)
        ring = np.zeros((len(ang), 3), np.float32)
        ring[:, 0] = (rng * np.cos(ang)).astype(np.float32)
        ring[:, 1] = (rng * np.sin(ang)).astype(np.float32)
        inp["points"] = ring
    if variant == "longseg":
        seg, acc = syn.straight_segment(1600, 0.005)
        inp["seg_xyz"], inp["acc_at_seg"] = seg, acc
    return inp


@pytest.mark.parametrize("opts", [
    dict(force_split=1),           # roll-out + host window bits + pose-parallel collision + compaction
    dict(cost_kernel=2),           # wavefront-per-sample cost kernel for every list
    dict(cost_kernel=1),           # workgroup-per-sample cost kernel for every list
    dict(device_trig=0),           # FALLBACK: the host's libm trig table, complete before the launch
    dict(device_trig=0, force_split=1),
    dict(sensor_on_host=1),        # sensor update (voxel bitmap, buckets) built on the host, dilate_kernel behind it
    dict(sensor_two_launch=1),     # the sensor build of clouds beyond 32 k points, at any size
    dict(fused_cycle=0),           # three kernels
    dict(fused_cycle=2, host_reduce=0),   # single launch with the ticket epilogue (what a sharded cycle runs)
], ids=lambda e: ",".join(f"{k}={v}" for k, v in e.items()))
def test_alternate_paths_equal_default_path(opts):
    """Every alternative device path (kc_dwa_set_option, per context) must agree bit for bit with the default
    path and with the oracle: paths, admissible set, per-sample costs, winner."""
    for k, sc in enumerate(_PATH_SCENARIOS):
        inp = _path_scenario(*sc)
        o = oracle_cycle(inp)
        h = hip_cycle(kh, inp)
        assert_cycle_equal(o, h)
        if sc[4] == "open":
            assert len(h["raw"]) == len(inp["vx"])  # nothing dropped
        for rep in range(2):  # (a second context under the same options: same answer again)
            ctx = hip_context(kh, inp)
            for name, v in opts.items():
                ctx.set_option(name, v)
            a = hip_cycle(kh, inp, ctx=ctx)
            assert_cycle_equal(o, a)
            np.testing.assert_array_equal(a["costs"].view(np.uint32), h["costs"].view(np.uint32))
            ctx.close()


@pytest.mark.parametrize("where", ["body", "tail", "none"])
def test_point_list_with_non_finite_entries(where):
    """NaN / inf coordinates in a point list: dropped by the octree (add_voxel) and never the nearest
    obstacle, on the device builder as in the oracle -- also when they sit in the last points of the
    list (the vectorised bounds pass of the host treats the tail separately)."""
    inp = syn.make_controller_inputs("cfg2", seed=6, scale=0.25)
    pts = np.asarray(inp["points"], np.float32).reshape(-1, 3).copy()
    pts = pts[: len(pts) - (len(pts) % 4) + 2]          # a tail of two points behind the groups of four
    if where == "body":
        pts[10, 0] = np.nan
        pts[501, 1] = np.inf
        pts[999, 2] = -np.inf
    elif where == "tail":
        pts[-1, 1] = np.nan
    inp["points"] = pts
    o = oracle_cycle(inp)
    h = hip_cycle(kh, inp)
    assert len(o["raw"]) > 0
    assert_cycle_equal(o, h)


def test_contexts_side_by_side_and_from_two_threads():
    """Two controller contexts, a mapper and a cloud context in one process:
    interleaved from one thread, then driven from two threads at once (the
    worker pool of the host trig table is shared and serialises its jobs).
    Every cycle must equal the oracle's, whatever ran in between."""
    import threading
    inps = [_path_scenario(*_PATH_SCENARIOS[1]), _path_scenario(*_PATH_SCENARIOS[5])]
    want = [oracle_cycle(i) for i in inps]
    ctxs = [hip_context(kh, i) for i in inps]
    m = kh.MapperContext(200, 200, 0.05, (0, 0, 0), 0.0, 512)
    ang, rng = syn.dense_scan(512, 0.5)
    grid_want = ko.scan_to_grid(200, 200, 0.05, (0, 0, 0), 0.0, ang, rng)
    for rep in range(3):
        for k in (0, 1, 1, 0):
            m.scan_to_grid_device(ang, rng)
            assert_cycle_equal(want[k], hip_cycle(kh, inps[k], ctx=ctxs[k]))
        np.testing.assert_array_equal(m.scan_to_grid(ang, rng), grid_want)
    errors = []

    def drive(k):
        try:
            for rep in range(20):
                assert_cycle_equal(want[k], hip_cycle(kh, inps[k], ctx=ctxs[k]))
        except Exception as e:  # noqa: BLE001 -- reported by the main thread
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=drive, args=(k,)) for k in (0, 1)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for c in ctxs:
        c.close()
    m.close()


def test_publish_result_hands_over_the_device_record():
    """Multi-GPU hand-off: rollout + evaluate, (all-reduce by the caller on the
    device record), kc_dwa_publish_result, kc_dwa_fetch_result -- without an
    exchange the record is this context's own result; after the caller lowers
    the key in place, the lowered key comes back."""
    import ctypes as C

    inp = syn.make_controller_inputs("cfg1", seed=2)
    ref = hip_cycle(kh, inp)
    ctx = hip_context(kh, inp)
    st = inp["state"]
    ctx.set_weights(kh.make_weights(*inp["weights"]))
    ctx.set_points(st, inp["points"], inp["max_range"])
    ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
    ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
    ctx.cycle(st, inp["P"])
    with pytest.raises(kh.KompassHipError):   # a host-reduced cycle leaves no device record to hand over
        ctx.publish_result()
    for _ in range(3):
        ctx.rollout(st, inp["P"])
        ctx.evaluate()
        ctx.publish_result()
        r = ctx.fetch_result()
        assert r.found and r.raw_index == ref["res"]["raw_index"]
        assert np.float32(r.cost) == np.float32(ref["res"]["cost"])
        assert r.n_admissible == ref["res"]["n_admissible"] and r.index == ref["res"]["index"]
    # stand-in for the all-reduce: another rank's better key written into the record
    hip = C.CDLL("libamdhip64.so")
    better = np.array([kh.lib().kc_key_pack(C.c_float(0.125), C.c_int64(123456))], np.int64)
    ctx.rollout(st, inp["P"])
    ctx.evaluate()
    ctx.fetch_result()  # the stream is idle now
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(C.c_void_p(ctx.result_device_ptr()), better.ctypes.data_as(C.c_void_p), 8, 1) == 0
    ctx.publish_result()
    r = ctx.fetch_result()
    assert r.found and r.raw_index == 123456 and np.float32(r.cost) == np.float32(0.125)


@pytest.mark.parametrize("H,W", [(200, 200), (123, 77), (401, 399)])
def test_mapper_scan_sequences_on_one_context(H, W):
    """Plain scans alternate between two device grids (the endpoint kernel of a scan clears the other
    grid for the next; odd cell counts fall back to the memset): many different scans on ONE context,
    an empty scan and a device-resident scan in between, every grid equal to the oracle's."""
    r = np.random.default_rng(5 + H)
    res = 0.05
    m = kh.MapperContext(H, W, res, (0.1, -0.2, 0.0), 0.3, 720)
    ext = min(H, W) * res
    for case in range(12):
        n = int(r.choice([1, 90, 720]))
        ang = np.sort(r.uniform(-np.pi, np.pi, n))
        rng = r.uniform(0.05, 0.9 * ext, n)
        if case == 5:
            g = m.scan_to_grid(np.zeros(0), np.zeros(0))
            assert (g == -1).all()
            continue
        want = ko.scan_to_grid(H, W, res, (0.1, -0.2, 0.0), 0.3, ang, rng)
        if case % 4 == 3:   # device-resident scan, then the same scan copied out: two scans, same grid
            m.scan_to_grid_device(ang, rng)
            m.sync()
        got = m.scan_to_grid(ang, rng)
        assert np.array_equal(got, want), (case, n, int((got != want).sum()))
    m.close()


@pytest.mark.parametrize("shape,dims", [(syn.CYLINDER, [0.1, 0.4]), (syn.BOX, [0.5, 0.3, 0.6])], ids=["cylinder", "box"])
@pytest.mark.parametrize("opts", [dict(), dict(sensor_on_host=1), dict(sensor_two_launch=1), dict(force_split=1)],
                         ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()) or "default")
def test_sensor_frame_point_list(shape, dims, opts):
    """updateSensorData(cloud, global_frame = false), collision_check.h:119-131 -- the list is in
    the SENSOR frame, the octree frame is body->tf * sensor_tf_body (a mount rotated about z and shifted); the
    obstacle list of the cost term is the same as for a world-frame list.  Every device path against the oracle's
    restatement (ko_coll_update_points(..., 0)); a mount that is not a rotation about z is refused."""
    import math

    inp = syn.make_controller_inputs("cfg2", seed=9, scale=0.25)
    inp["robot"] = dict(shape=shape, dims=dims)
    spos, srot = (0.12, -0.05, 0.15), (0.0, 0.0, math.sin(0.2), math.cos(0.2))
    st = (0.3, -0.2, 0.25, 0.0)
    pts = np.ascontiguousarray(inp["points"], np.float32)
    rb = inp["robot"]
    coll = ko.Collision(rb["shape"], rb["dims"], spos, srot, inp["octree_res"])
    coll.update_state(st[0], st[1], st[2])
    coll.update_points(pts, False)
    ox, oy = ko.obstacles_from_points(spos, srot, st, pts)
    px, py, raw, _ = ko.rollout(coll, st, inp["dt"], inp["P"], inp["vx"], inp["vy"], inp["omega"])
    ci = ko.CostInputs(inp["seg_xyz"], 0, inp["acc_at_seg"], inp["ref_len"], np.stack([ox, oy], axis=1),
                       np.float32(inp["max_range"]) / np.float32(3.0), inp["acc_limits"], ko.make_weights(*inp["weights"]))
    assert 0 < len(px) < len(inp["vx"])
    idx, cost, costs = ko.min_trajectory_cost(ci, px, py, None)
    # the world-frame reading of the same numbers is a different scene: the test would not notice a frame mix-up otherwise
    coll_w = ko.Collision(rb["shape"], rb["dims"], spos, srot, inp["octree_res"])
    coll_w.update_state(st[0], st[1], st[2])
    coll_w.update_points(pts, True)
    assert not np.array_equal(ko.rollout(coll_w, st, inp["dt"], inp["P"], inp["vx"], inp["vy"], inp["omega"])[2], raw)
    ctx = hip_context(kh, inp, spos, srot)
    for k, v in opts.items():
        ctx.set_option(k, v)
    ctx.set_weights(kh.make_weights(*inp["weights"]))
    ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
    ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
    for rep in range(2):
        ctx.set_points(st, pts, inp["max_range"], global_frame=False)
        r = ctx.cycle(st, inp["P"])
        hx, hy, hraw, hcosts = ctx.get_samples(with_costs=True)
        np.testing.assert_array_equal(hraw, raw)
        np.testing.assert_array_equal(hx.view(np.uint32), px.view(np.uint32))
        np.testing.assert_array_equal(hcosts.view(np.uint32), costs.view(np.uint32))
        assert r.found and r.index == idx and np.float32(r.cost) == np.float32(cost)
    ctx.close()
    tilted = hip_context(kh, inp, spos, (0.0, math.sin(0.2), 0.0, math.cos(0.2)))
    with pytest.raises(Exception):
        tilted.set_points(st, pts, inp["max_range"], global_frame=False)
    tilted.close()
