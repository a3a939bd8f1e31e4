"""The single-launch controller cycle (kc_dwa_cycle with option "fused_cycle",
default on: roll-out + collision gate + costs + argmin + record in ONE kernel,
SURVEY 7 step 5 / dwa.h:215-229) against the three-kernel cycle and the oracle;
the per-context options; the call sequences the synchronisation shortcuts of
the context have to survive."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import kompass_hip as kh  # noqa: E402
import synthetic as syn  # noqa: E402

from helpers import assert_cycle_equal, hip_context, hip_cycle, oracle_cycle  # noqa: E402
from test_gpu_parity import _PATH_SCENARIOS, _path_scenario  # noqa: E402


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert kh.device_count() >= 1, "no HIP device visible: the -m gpu tests need an MI355X"


def _prepared(inp, **options):
    ctx = hip_context(kh, inp)
    for k, v in options.items():
        ctx.set_option(k, v)
    st = inp["state"]
    ctx.set_weights(kh.make_weights(*inp["weights"]))
    ctx.set_points(st, inp["points"], inp["max_range"])
    ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
    ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
    return ctx


@pytest.mark.parametrize("k", range(len(_PATH_SCENARIOS)))
def test_single_launch_cycle_equals_three_kernel_cycle_and_oracle(k):
    inp = _path_scenario(*_PATH_SCENARIOS[k])
    o = oracle_cycle(inp)
    one = _prepared(inp, fused_cycle=2)
    three = _prepared(inp, fused_cycle=0)
    h1 = hip_cycle(kh, inp, ctx=one)
    h3 = hip_cycle(kh, inp, ctx=three)
    # the single launch with its device-side epilogue (arrival ticket; what a sharded cycle runs)
    tick = _prepared(inp, fused_cycle=2, host_reduce=0)
    assert_cycle_equal(o, hip_cycle(kh, inp, ctx=tick))
    tick.close()
    assert one.get_option("fused_cycle") == 2 and three.get_option("fused_cycle") == 0
    assert three.get_option("last_cycle_single_launch") == 0
    assert one.get_option("last_cycle_single_launch") == 1, "cost tables must fit beside the roll-out tile here"
    assert_cycle_equal(o, h1)
    assert_cycle_equal(o, h3)
    # the rows stored by the cycle kernel itself (write_paths) are the same rows
    wp = _prepared(inp, write_paths=1, fused_cycle=2)
    r = wp.cycle(inp["state"], inp["P"])
    px, py, raw, costs = wp.get_samples(with_costs=True)
    np.testing.assert_array_equal(px.view(np.uint32), o["px"].view(np.uint32))
    np.testing.assert_array_equal(py.view(np.uint32), o["py"].view(np.uint32))
    np.testing.assert_array_equal(costs.view(np.uint32), o["costs"].view(np.uint32))
    assert r.index == o["index"]
    for c in (one, three, wp):
        c.close()


def test_many_cycles_moving_pose_both_paths_agree():
    """200 cycles with a pose that moves every cycle (nothing reusable), survivors
    from none to all: the two paths must agree every time; a few against the oracle."""
    inp = syn.make_controller_inputs("cfg2", seed=3, scale=0.3, scene="mid")
    # (fused_cycle = 2: by itself kc_dwa_cycle would hand this small, survivor-rich shard to the cost kernels)
    one, three, tick = _prepared(inp, fused_cycle=2), _prepared(inp, fused_cycle=0), _prepared(inp, fused_cycle=2, host_reduce=0)
    P = inp["P"]
    for i in range(200):
        st = (0.02 * (i % 50) - 0.5, 0.013 * (i % 37) - 0.2, 0.05 * (i % 9) - 0.2, 0.0)
        a, b = one.cycle(st, P), three.cycle(st, P)
        t = tick.cycle(st, P)
        assert (t.found, t.index, t.raw_index, t.n_admissible) == (a.found, a.index, a.raw_index, a.n_admissible), i
        assert (a.found, a.index, a.raw_index, a.n_admissible) == (b.found, b.index, b.raw_index, b.n_admissible), i
        assert np.float32(a.cost) == np.float32(b.cost)
        if a.found:
            np.testing.assert_array_equal(one.get_best()[0], three.get_best()[0])
        if i % 50 == 7:
            o = oracle_cycle(dict(inp, state=st))
            assert a.n_admissible == len(o["raw"]) and a.index == o["index"]
    one.close(); three.close(); tick.close()


@pytest.mark.parametrize("opts", [dict(fused_cycle=2), dict(fused_cycle=2, cycle_samples=16), dict(fused_cycle=2, cycle_samples=32),
                                  dict(fused_cycle=2, cycle_samples=16, host_reduce=0), dict(near_table=0, fused_cycle=2), dict(host_reduce=0), dict(cost_kernel=1), dict(cost_kernel=2), dict(force_split=1), dict(device_trig=0),
                                  dict(device_trig=0, force_split=1), dict(device_trig=0, fused_cycle=2, cycle_samples=16), dict(sensor_on_host=1), dict(sensor_two_launch=1),
                                  dict(fused_cycle=0, cost_kernel=2)],
                         ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()))
def test_options_per_context_give_identical_results(opts):
    """Every switch of kc_dwa_set_option, set on ONE context of a process that
    also runs a default context: same bits as the oracle."""
    for sc in (_PATH_SCENARIOS[1], _PATH_SCENARIOS[5]):
        inp = _path_scenario(*sc)
        o = oracle_cycle(inp)
        ctx = hip_context(kh, inp)
        for k, v in opts.items():
            ctx.set_option(k, v)
            assert ctx.get_option(k) == v
        assert_cycle_equal(o, hip_cycle(kh, inp, ctx=ctx))
        assert_cycle_equal(o, hip_cycle(kh, inp))
        ctx.close()
    with pytest.raises(ValueError):
        hip_context(kh, inp).set_option("no_such_option", 1)


def test_evaluate_without_fetch_then_table_updates():
    """after a fetched cycle, kc_dwa_evaluate again WITHOUT a fetch, then
    a new tracked segment (host stores into the table the queued kernel reads),
    then a cycle; and kc_dwa_set_tracked_window followed by kc_dwa_set_tracked_segment."""
    inp = _path_scenario(*_PATH_SCENARIOS[3])       # open space: the cost kernel has work to do
    seg2 = inp["seg_xyz"] + np.float32([0.0, 0.35, 0.0])
    o1 = oracle_cycle(inp)
    o2 = oracle_cycle(dict(inp, seg_xyz=seg2))
    for fused in (2, 1, 0):
        ctx = _prepared(inp, fused_cycle=fused)
        st, P = inp["state"], inp["P"]
        for rep in range(5):
            r = ctx.cycle(st, P)
            assert r.index == o1["index"] and np.float32(r.cost) == np.float32(o1["cost"])
            ctx.evaluate()                                   # queued, never fetched
            ctx.set_tracked_segment(seg2, inp["acc_at_seg"], inp["ref_len"])
            r = ctx.cycle(st, P)
            assert r.index == o2["index"] and np.float32(r.cost) == np.float32(o2["cost"]), (fused, rep)
            _, _, _, costs = ctx.get_samples(with_costs=True)
            np.testing.assert_array_equal(costs.view(np.uint32), o2["costs"].view(np.uint32))
            # resident window, then host-built tables over it
            ctx.set_path(seg2, inp["acc_at_seg"], inp["ref_len"])
            ctx.set_tracked_window(0, len(seg2))
            ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
            r = ctx.cycle(st, P)
            assert r.index == o1["index"] and np.float32(r.cost) == np.float32(o1["cost"]), (fused, rep)
        ctx.close()


def test_rollout_evaluate_after_a_single_launch_cycle():
    """The split entry points keep working on a context whose last cycle was a
    single launch (rows are re-materialised, the list is rebuilt from the flags)."""
    inp = _path_scenario(*_PATH_SCENARIOS[0])
    o = oracle_cycle(inp)
    ctx = _prepared(inp)
    st, P = inp["state"], inp["P"]
    r0 = ctx.cycle(st, P)
    ctx.evaluate()                      # second evaluate of the same roll-out
    r1 = ctx.fetch_result()
    ctx.rollout(st, P); ctx.evaluate()
    r2 = ctx.fetch_result()
    for r in (r0, r1, r2):
        assert r.index == o["index"] and np.float32(r.cost) == np.float32(o["cost"])
        assert r.n_admissible == len(o["raw"])
    assert ctx.count_admissible_before(int(o["raw"][o["index"]])) == o["index"]
    ctx.close()


def test_two_threads_with_differently_configured_contexts():
    """contexts that take different paths through the shared host pool and the BAR hand-offs,
    driven from two threads at once -- a single-launch early-launched cycle beside a split / timed /
    copy-path one -- 60 cycles each with fresh sensor data and lattice every few cycles."""
    import threading

    inps = [_path_scenario(*_PATH_SCENARIOS[1]), _path_scenario(*_PATH_SCENARIOS[5])]
    want = [oracle_cycle(i) for i in inps]
    configs = [(dict(fused_cycle=2), dict(force_split=1)), (dict(fused_cycle=2, host_reduce=0), dict(device_trig=0)),
               (dict(), dict(fused_cycle=0, device_trig=0)), (dict(device_trig=0), dict())]
    for opts_a, opts_b in configs:
        ctxs = [_prepared(inps[0], **opts_a), _prepared(inps[1], **opts_b)]
        ctxs[1].timing_enable(True)
        errors = []

        def drive(k):
            try:
                inp, ctx = inps[k], ctxs[k]
                for rep in range(60):
                    if rep % 4 == 0:
                        ctx.set_points(inp["state"], inp["points"], inp["max_range"])
                        ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
                    if rep % 7 == 0:
                        ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
                    r = ctx.cycle(inp["state"], inp["P"])
                    assert r.index == want[k]["index"] and np.float32(r.cost) == np.float32(want[k]["cost"]), rep
                    assert r.n_admissible == len(want[k]["raw"])
                    if r.found and rep % 5 == 0:
                        np.testing.assert_array_equal(ctx.get_best()[0], want[k]["px"][want[k]["index"]])
            except Exception as e:  # noqa: BLE001 -- reported by the main thread
                errors.append((k, repr(e)))

        threads = [threading.Thread(target=drive, args=(k,)) for k in (0, 1)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, (opts_a, opts_b, errors)
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("zoff", [0.0, 0.12, -0.2])
def test_sphere_on_the_single_launch_path(zoff):
    """Spheres: the z gap of a voxel column is a byte code into a table of the few
    voxel layers within the sphere's height, read by the exact tests of the fused kernel.  Points at
    several heights, the sensor above / below the sphere's centre: single launch, three kernels and the
    split path (window built on the host) against the oracle."""
    inp = syn.make_controller_inputs("cfg2", seed=6, scale=0.25)
    inp["robot"] = dict(shape=syn.SPHERE, dims=[0.22])
    rng = np.random.default_rng(9)
    pts = np.asarray(inp["points"], np.float32).copy()
    pts[:, 2] = rng.choice([-0.31, -0.2, -0.05, 0.0, 0.07, 0.12, 0.21, 0.4], size=len(pts)).astype(np.float32)
    inp["points"] = pts
    spos = (0.05, -0.02, zoff)
    o = oracle_cycle(inp, sensor_pos=spos)
    assert 0 < len(o["raw"]) < len(inp["vx"])
    for opts, single in ((dict(fused_cycle=2), 1), (dict(fused_cycle=0), 0), (dict(force_split=1), 0),
                         (dict(fused_cycle=2, host_reduce=0), 1), (dict(fused_cycle=2, cycle_samples=16), 1)):
        ctx = hip_context(kh, inp, sensor_pos=spos)
        for k, v in opts.items():
            ctx.set_option(k, v)
        assert_cycle_equal(o, hip_cycle(kh, inp, sensor_pos=spos, ctx=ctx))
        assert ctx.get_option("last_cycle_single_launch") == single
        # and again on the same context (tables resident)
        r = ctx.cycle(inp["state"], inp["P"])
        assert r.index == o["index"] and r.n_admissible == len(o["raw"])
        ctx.close()


@pytest.mark.parametrize("beams", [360, 1440, 4096])
def test_room_like_scans_far_walls_dense_rows(beams):
    """Laser scans of a room: every trajectory point is metres from the nearest wall, and a dense scan puts
    hundreds of points into one bucket row -- the cooperative far search of the wavefront-per-sample cost stage
    walks such rows with all 64 lanes (kLongRun), short runs with one lane.  Single launch, three kernels and
    the workgroup-per-sample kernel against the oracle, two poses (one near a wall: mixed near / far points)."""
    inp = syn.make_controller_inputs("cfg2", seed=4, scale=0.22, scene="open")
    ang = np.linspace(-np.pi, np.pi, beams, endpoint=False)
    rng = 3.0 + 1.2 * np.cos(5 * ang) + 0.3 * np.sin(17 * ang)
    for st in ((0.0, 0.0, 0.3, 0.0), (1.1, -0.6, -1.0, 0.0)):
        cur = dict(inp, state=st)
        o = oracle_cycle(cur, scan=(rng, ang))
        assert len(o["raw"]) > 40
        for opts in (dict(fused_cycle=2), dict(fused_cycle=0, cost_kernel=2), dict(fused_cycle=0, cost_kernel=1)):
            ctx = hip_context(kh, cur)
            for k, v in opts.items():
                ctx.set_option(k, v)
            assert_cycle_equal(o, hip_cycle(kh, cur, scan=(rng, ang), ctx=ctx))
            ctx.close()


def test_find_best_path_is_the_four_entries_in_one_call():
    """kc_dwa_find_best_path (DWA::findBestPath, dwa.h:183-230) = kc_dwa_sample_window + kc_dwa_set_points |
    kc_dwa_set_scan + kc_dwa_set_tracked_segment + kc_dwa_cycle: same record, same winner row, same as the
    oracle; parts left out keep the context's state."""
    from oracle import ko

    inp = syn.make_controller_inputs("cfg2", seed=3, scale=0.3)
    lim = kh.make_limits(syn.LIMITS["vx"], syn.LIMITS["vy"], syn.LIMITS["omega"])
    pts = np.ascontiguousarray(inp["points"], np.float32)
    seg = np.ascontiguousarray(inp["seg_xyz"], np.float32)
    acc = np.ascontiguousarray(inp["acc_at_seg"], np.float32)
    one, four = hip_context(kh, inp), hip_context(kh, inp)
    for c in (one, four):
        c.set_weights(kh.make_weights(*inp["weights"]))
    P = inp["P"]
    for i in range(6):
        st = (0.02 * i, -0.01 * i, 0.03 * i, 0.0)
        cur = (0.4 + 0.01 * i, 0.0, 0.02 * (i - 2))
        a = one.find_best_path(st, P, window=(syn.DIFFERENTIAL_DRIVE, lim, cur, 21, 15), points=pts,
                               max_sensor_range=inp["max_range"], segment=(seg, acc, inp["ref_len"]))
        wvx, wvy, wom = four.sample_window(syn.DIFFERENTIAL_DRIVE, lim, cur, 21, 15)
        four.set_points(st, pts, inp["max_range"])
        four.set_tracked_segment(seg, acc, inp["ref_len"])
        b = four.cycle(st, P)
        assert (a.found, a.index, a.raw_index, a.n_admissible, a.n_samples) == (b.found, b.index, b.raw_index, b.n_admissible, b.n_samples)
        assert np.float32(a.cost) == np.float32(b.cost)
        if a.found:
            np.testing.assert_array_equal(one.get_best()[0], four.get_best()[0])
        o = oracle_cycle(dict(inp, vx=wvx, vy=wvy, omega=wom, state=st))
        assert a.n_admissible == len(o["raw"]) and a.index == o["index"]
        if a.found:
            assert np.float32(a.cost) == np.float32(o["cost"])
    # parts left out: the context keeps its window / sensor data / segment
    c2 = one.find_best_path(st, P)
    assert (c2.found, c2.index, c2.raw_index, c2.n_admissible) == (a.found, a.index, a.raw_index, a.n_admissible)
    # a scan instead of points
    ang, rng = syn.dense_scan(360)
    s1 = one.find_best_path(st, P, scan=(rng, ang), max_sensor_range=inp["max_range"])
    four.set_scan(st, rng, ang, inp["max_range"])
    s2 = four.cycle(st, P)
    assert (s1.found, s1.index, s1.raw_index, s1.n_admissible) == (s2.found, s2.index, s2.raw_index, s2.n_admissible)
    one.close(); four.close()


@pytest.mark.gpu
def test_a_moving_window_finds_its_patterns_on_the_device():
    """A robot whose velocity wanders changes the PATTERN of its window lattice (an axis gains or loses a value, a value
    crosses |v| = kMinVel) between a handful of patterns: every cycle equals a fresh context's, and a pattern seen
    before comes back by a swap of device tables instead of a rebuild (counters pattern_hits / pattern_builds)."""
    import synthetic as syn

    inp = syn.make_controller_inputs("cfg2", seed=2, scale=0.25, scene="mid")
    lim = kh.make_limits(syn.LIMITS["vx"], syn.LIMITS["vy"], syn.LIMITS["omega"])
    ctr = syn.CONFIGS["cfg2"]["ctr"]
    P = inp["P"]
    seg = np.asarray(inp["seg_xyz"], np.float32)
    sacc = np.ascontiguousarray(inp["acc_at_seg"], np.float32)
    pts = np.ascontiguousarray(inp["points"], np.float32)

    def make():
        rb = inp["robot"]
        c = kh.DwaContext(rb["shape"], rb["dims"], (0, 0, 0), (0, 0, 0, 1), inp["octree_res"], inp["dt"], max_samples=4096,
                          max_points=P, max_segment=len(seg), max_obstacles=len(pts), acc_limits=inp["acc_limits"])
        c.set_weights(kh.make_weights(*inp["weights"]))
        return c

    ctx = make()
    rng = np.random.default_rng(5)
    vels = [(float(v), 0.0, float(o)) for v, o in zip(rng.uniform(-0.1, 1.0, 6), rng.uniform(-0.8, 0.8, 6))]
    walk = [vels[int(k)] for k in rng.integers(0, len(vels), 60)]
    counts = set()
    for i, cur in enumerate(walk):
        st = (0.0, 0.0, 0.01 * (i % 5), 0.0)
        r = ctx.find_best_path(st, P, window=(ctr, lim, cur, 31, 31), points=pts, max_sensor_range=inp["max_range"],
                               segment=(seg, sacc, inp["ref_len"]))
        counts.add(int(r.n_samples))
        ref = make()   # the same cycle on a context that has never seen another window
        q = ref.find_best_path(st, P, window=(ctr, lim, cur, 31, 31), points=pts, max_sensor_range=inp["max_range"],
                               segment=(seg, sacc, inp["ref_len"]))
        assert (r.found, r.index, r.raw_index, r.n_admissible, r.n_samples) == (q.found, q.index, q.raw_index, q.n_admissible, q.n_samples)
        assert np.float32(r.cost) == np.float32(q.cost)
        if r.found:
            np.testing.assert_array_equal(np.array(ctx.get_best()[0]), np.array(ref.get_best()[0]))
        ref.close()
    assert len(counts) > 1, "the walk was meant to change the pattern"
    builds, hits = ctx.get_option("pattern_builds"), ctx.get_option("pattern_hits")
    assert builds <= len(vels) + 1 and hits >= 1, (builds, hits, counts)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("P", [44, 38, 31])
def test_shorter_horizons_through_the_teams(P):
    """cfg2 on the clutter scene with a shortened horizon: 700 .. 2000 samples survive, three or four a workgroup -- the
    workgroups that cost their survivors by QUARTERS, whose obstacle term the team's last wavefront now forms
    (wave_obstacle_term; round 4: the block walk a point apiece made those workgroups 16 us late).  Full size, every
    cost against the threaded oracle."""
    import synthetic as syn
    from helpers import oracle_cycle_mt

    inp = syn.make_controller_inputs("cfg2", seed=0, scene="survey")
    inp = dict(inp, P=P)
    o = oracle_cycle_mt(inp)
    ctx = hip_context(kh, inp)
    h = hip_cycle(kh, inp, ctx=ctx)
    assert h["res"]["n_admissible"] == len(o["raw"]) and 500 < len(o["raw"]) < 3000
    np.testing.assert_array_equal(h["raw"], o["raw"])
    np.testing.assert_array_equal(h["costs"].view(np.uint32), o["costs"].view(np.uint32))
    assert h["res"]["found"] and h["res"]["index"] == o["index"]
    for tm in (2, 0):   # halves only / a wavefront a sample: the same record
        ctx.set_option("team_max", tm)
        r = ctx.cycle(inp["state"], P)
        assert ctx.get_option("last_cycle_single_launch") == 1.0
        assert (r.found, r.index, r.raw_index, r.n_admissible) == (True, h["res"]["index"], h["res"]["raw_index"], h["res"]["n_admissible"])
        assert np.float32(r.cost) == np.float32(h["res"]["cost"])
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("p_occ,scale", [(0.0003, 1.0), (0.001, 0.5), (0.0001, 0.5)])
def test_sparse_scenes_every_cost_against_the_oracle(p_occ, scale):
    """A few obstacles metres away from every trajectory: no point of a sample has an occupied cell within two cells, the
    case the union-rectangle scan used to hand to the ring walks (round 4: 73 us a cycle; its seed block now grows to the
    smallest skip value of the sample).  Every admissible sample's cost, the winner, bit for bit."""
    import synthetic as syn
    from helpers import oracle_cycle_mt

    inp = syn.make_controller_inputs("cfg2", seed=0, scale=scale, scene="survey")
    inp = dict(inp, points=syn.costmap_points(syn.CONFIGS["cfg2"]["map_side"], 0.05, 3, p_occ=p_occ, free_radius=1.0))
    o = oracle_cycle_mt(inp)
    ctx = hip_context(kh, inp)
    h = hip_cycle(kh, inp, ctx=ctx)
    assert h["res"]["n_admissible"] == len(o["raw"]) and len(o["raw"]) > 0.8 * len(inp["vx"])
    np.testing.assert_array_equal(h["raw"], o["raw"])
    np.testing.assert_array_equal(h["costs"].view(np.uint32), o["costs"].view(np.uint32))
    assert h["res"]["found"] and h["res"]["index"] == o["index"]
    for opt in (dict(fused_cycle=0), dict(obs_union=0)):   # the three-kernel cycle; the walks alone
        for k, v in opt.items():
            ctx.set_option(k, v)
        r = ctx.cycle(inp["state"], inp["P"])
        assert (r.found, r.index, r.raw_index, r.n_admissible) == (True, h["res"]["index"], h["res"]["raw_index"], h["res"]["n_admissible"])
        assert np.float32(r.cost) == np.float32(h["res"]["cost"])
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("sensor_z", [0.0, 0.18, -0.12])
def test_sphere_sensor_update_on_the_device(sensor_z):
    """Round 4: a sphere's sensor update runs in the one-launch device build (voxel acceptance by layer, the smallest
    layer code of every column, the gap LUT of the layers the cloud's z range can hold).  Clouds over several heights and
    sizes, the sensor above / below the sphere's centre: the oracle's admissible set, costs and winner, and the
    host-built update's record."""
    inp = syn.make_controller_inputs("cfg2", seed=5, scale=0.25, scene="mid")
    inp["robot"] = dict(shape=syn.SPHERE, dims=[0.22])
    rng = np.random.default_rng(int(100 * abs(sensor_z)) + 3)
    dev = hip_context(kh, inp, sensor_pos=(0, 0, sensor_z))
    host = hip_context(kh, inp, sensor_pos=(0, 0, sensor_z))
    host.set_option("sensor_on_host", 1)
    for n in (7, 900, 5000):
        pts = np.asarray(inp["points"], np.float32)[rng.choice(len(inp["points"]), n, replace=False)].copy()
        pts[:, 2] = rng.choice([-0.35, -0.1, 0.0, 0.07, 0.2, 0.45], n)
        cur = dict(inp, points=pts)
        o = oracle_cycle(cur, sensor_pos=(0, 0, sensor_z))
        h = hip_cycle(kh, cur, sensor_pos=(0, 0, sensor_z), ctx=dev)
        assert_cycle_equal(o, h)
        g = hip_cycle(kh, cur, sensor_pos=(0, 0, sensor_z), ctx=host)
        assert (h["res"]["found"], h["res"]["index"], h["res"]["n_admissible"]) == (g["res"]["found"], g["res"]["index"], g["res"]["n_admissible"])
    # a 3-D cloud two and a half metres tall (fifty voxel layers: the layer table holds the ones that can touch the sphere)
    pts = np.asarray(inp["points"], np.float32)[rng.choice(len(inp["points"]), 3000, replace=False)].copy()
    pts[:, 2] = rng.uniform(-0.6, 1.9, len(pts)).astype(np.float32)
    cur = dict(inp, points=pts)
    o = oracle_cycle(cur, sensor_pos=(0, 0, sensor_z))
    h = hip_cycle(kh, cur, sensor_pos=(0, 0, sensor_z), ctx=dev)
    assert_cycle_equal(o, h)
    g = hip_cycle(kh, cur, sensor_pos=(0, 0, sensor_z), ctx=host)
    assert (h["res"]["found"], h["res"]["index"], h["res"]["n_admissible"]) == (g["res"]["found"], g["res"]["index"], g["res"]["n_admissible"])
    dev.close()
    host.close()


@pytest.mark.gpu
@pytest.mark.parametrize("dims", [[1.2, 0.3, 0.4], [0.2, 0.9, 0.3], [1.5, 0.2, 0.5], [0.8, 0.4, 0.4], [2.4, 0.2, 0.3]])
@pytest.mark.parametrize("mode", ["cloud", "scan", "freeze"])
def test_long_boxes_look_up_circles_along_their_axis(dims, mode):
    """A box at least twice as long as wide: the pose gate looks `cover` circles up in the dilated masks (outer radius
    hypot(A / cover, B), inner radius B, CollDev::cover) instead of one circle around the centre -- long axis along x or
    along y, 2 .. 8 circles, point cloud / LaserScan / the freezing sampler; against the oracle, and against the single
    look-up (option box_cover = 0), several yaws."""
    inp = syn.make_controller_inputs("cfg2", seed=6, scale=0.25, scene="mid")
    inp["robot"] = dict(shape=syn.BOX, dims=dims)
    scan = None
    if mode == "scan":
        ang = np.linspace(-np.pi, np.pi, 720, endpoint=False)
        scan = (2.2 + 0.9 * np.cos(3 * ang) + 0.2 * np.sin(11 * ang), ang)
    for yaw in (0.0, 0.7, -2.1):
        cur = dict(inp, state=(0.1, -0.2, yaw, 0.0))
        o = oracle_cycle(cur, scan=scan) if mode != "freeze" else None
        res = []
        for cover in (1, 0):
            ctx = hip_context(kh, cur)
            ctx.set_option("box_cover", cover)
            if mode == "freeze":
                ctx.set_option("drop_samples", 0)
            h = hip_cycle(kh, cur, scan=scan, ctx=ctx)
            assert ctx.get_option("last_cycle_single_launch") == 1
            if o is not None:
                assert_cycle_equal(o, h)
            res.append(h)
            ctx.close()
        assert res[0]["res"] == res[1]["res"]
        np.testing.assert_array_equal(res[0]["raw"], res[1]["raw"])
        np.testing.assert_array_equal(res[0]["costs"].view(np.uint32), res[1]["costs"].view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("beams", [360, 1440, 4096])
def test_scans_with_beams_without_a_return_keep_their_near_table(beams):
    """Real scanners report inf (or NaN) beyond their range.  Such a beam's obstacle never wins a minimum (cost side:
    `dist < minDist` is false for it; collision side: add_voxel drops it), so it stays in the scan's list and
    out of the chunk boxes -- the scan keeps its near table (until round 4 one such beam sent the whole scan to the bucket
    search).  Runs of missing beams, isolated ones, a whole chunk without returns; against the oracle, near table on / off."""
    inp = syn.make_controller_inputs("cfg2", seed=4, scale=0.3, scene="open")
    ang = np.linspace(-np.pi, np.pi, beams, endpoint=False)
    rng = 3.0 + 1.2 * np.cos(5 * ang) + 0.3 * np.sin(17 * ang)
    r = np.random.default_rng(beams)
    rng[r.random(beams) < 0.05] = np.inf
    rng[r.integers(0, beams, 5)] = np.nan
    a0 = beams // 3
    rng[a0:a0 + beams // 20] = np.inf            # a doorway: a run longer than a chunk of the scan
    cur = dict(inp, state=(0.3, -0.2, 0.4, 0.0))
    o = oracle_cycle(cur, scan=(rng, ang))
    assert len(o["raw"]) > 100
    for opts in (dict(), dict(fused_cycle=0, cost_kernel=2, cost_batch=2), dict(obs_near=0)):
        ctx = hip_context(kh, cur)
        for k, v in opts.items():
            ctx.set_option(k, v)
        assert_cycle_equal(o, hip_cycle(kh, cur, scan=(rng, ang), ctx=ctx))
        if "obs_near" not in opts:
            assert ctx.get_option("obs_near_rides") + ctx.get_option("obs_near_builds") >= 1
        ctx.close()


@pytest.mark.gpu
def test_clouds_of_several_hundred_thousand_points_on_the_device():
    """A raw depth-camera cloud (640 x 480 = 307 200 points) or a 128-beam lidar sweep: the device-side sensor update
    takes up to 2^20 points (262 144 until round 4; beyond it the host build took 6 ms at 500 k points).  Same admissible
    set, costs and winner as the host build, NaN points included."""
    inp = syn.make_controller_inputs("cfg2", seed=0, scale=0.35)
    rng = np.random.default_rng(3)
    n = 330_000
    th, rad = rng.uniform(-np.pi, np.pi, n), rng.uniform(2.2, 9.0, n) ** 1.0
    keep = rng.random(n) < np.clip((rad - 2.0) / 6.0, 0.02, 1.0)      # sparse near the robot, dense far out
    pts = np.stack([rad * np.cos(th), rad * np.sin(th), rng.uniform(-0.2, 1.5, n)], 1).astype(np.float32)
    pts[~keep, 2] = 5.0                                                # (above the robot: dropped by the voxel rule, still obstacles of the cost)
    pts[::997, 0] = np.nan
    out = []
    for host in (0, 1):
        ctx = hip_context(kh, dict(inp, points=pts), sensor_pos=(0, 0, 0.3))
        ctx.set_option("sensor_on_host", host)
        h = hip_cycle(kh, dict(inp, points=pts), sensor_pos=(0, 0, 0.3), ctx=ctx)
        out.append(h)
        ctx.close()
    assert out[0]["res"] == out[1]["res"] and out[0]["res"]["n_admissible"] > 50
    np.testing.assert_array_equal(out[0]["raw"], out[1]["raw"])
    np.testing.assert_array_equal(out[0]["costs"].view(np.uint32), out[1]["costs"].view(np.uint32))
