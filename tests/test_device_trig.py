"""Device trig (round 3, csrc/kc_trig_exact.h): the roll-out kernels evaluate glibc's `sincos` algorithm
themselves instead of reading a table the host's libm produced.  Pinned three ways: the table the algorithm
reads is regenerated from exact rational arithmetic and equals the committed header; the host restatement
equals the installed `sincos` on the library's argument set (what the library itself checks when it is loaded);
and -- on the GPU -- the kernel's table equals `sincos` of the host libm bit for bit on the yaw chains of every
BASELINE lattice and on awkward ones, and cycles with the switch on and off are bit-equal to the oracle
(the host-trig path is what rounds 1-3 shipped)."""
import ctypes
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kompass-core_amd"), os.path.join(ROOT, "tools")]

import kompass_hip as kh  # noqa: E402


def test_table_header_is_what_the_generator_produces():
    import gen_sincostab as g

    tab = g.table()
    assert len(tab) == 440 and tab[:4] == [0.0, 0.0, 1.0, 0.0]
    text = open(os.path.join(ROOT, "kompass-core_amd", "csrc", "kc_sincostab.h")).read()
    body = text.split("KC_SINCOSTAB_VALUES", 1)[1].replace("\\", " ")
    vals = [float.fromhex(t) for t in body.replace(",", " ").split()]
    assert vals == tab
    # high parts are the correctly rounded sine / cosine of k / 128 (what math.sin returns up to its own error)
    for k in (1, 17, 64, 109):
        assert abs(tab[4 * k] - np.sin(k / 128.0)) <= np.spacing(tab[4 * k])
        assert abs(tab[4 * k + 2] - np.cos(k / 128.0)) <= np.spacing(tab[4 * k + 2])


def test_restated_sincos_equals_the_installed_libm():
    assert kh.trig_selfcheck() > 50000


def _libm_table(yaw0, omega, P, dt):
    libm = ctypes.CDLL("libm.so.6")
    libm.sincos.argtypes = [ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    libm.sincos.restype = None
    out = np.empty((P, len(omega), 2))
    s, c = ctypes.c_double(), ctypes.c_double()
    for r, om in enumerate(omega):
        yaw = float(yaw0)
        w = float(om) * float(dt)
        for k in range(P):
            libm.sincos(yaw, ctypes.byref(s), ctypes.byref(c))
            out[k, r] = (c.value, s.value)
            yaw += w
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("yaw0,om_max,A,P,dt", [
    (0.3, 1.5, 128, 50, 0.05),            # cfg2: diff drive
    (-2.9, 0.9, 257, 100, 0.05),          # cfg3: Ackermann, long horizon
    (3.141592653589793, 2.0, 41, 50, 0.1),  # cfg5-like, start at pi
    (0.0, 1e-9, 9, 20, 0.05),             # |yaw| below 2^-27: sin = x, cos = 1
    (1.5707963267948966, 0.5, 33, 64, float(np.float32(0.02))),   # around pi / 2: the pi/2 - |x| branch
    (50.0, 3.0, 17, 200, 0.05),           # Cody-Waite reduction, every quadrant
    (-1.0e6, 40.0, 9, 300, 0.25),
    (99999000.0, 0.0, 3, 4, 0.05),        # just inside the range
    (0.2, 1.0, 5, 1500, 0.01),            # a very long horizon (no LDS staging: any number of steps)
])
def test_device_table_equals_host_sincos(yaw0, om_max, A, P, dt):
    omega = np.linspace(-om_max, om_max, A)
    got = kh.trig_table(yaw0, omega, P, dt)
    want = _libm_table(yaw0, omega, P, dt)
    np.testing.assert_array_equal(got.view(np.uint64), want.view(np.uint64))


@pytest.mark.gpu
def test_device_table_random_rows():
    rng = np.random.default_rng(7)
    for _ in range(6):
        omega = rng.uniform(-4, 4, size=rng.integers(1, 70))
        yaw0 = rng.uniform(-7, 7)
        P = int(rng.integers(2, 130))
        got = kh.trig_table(yaw0, omega, P, 0.05)
        np.testing.assert_array_equal(got.view(np.uint64), _libm_table(yaw0, omega, P, 0.05).view(np.uint64))


@pytest.mark.gpu
def test_device_table_quarter_million_entries():
    """250 000 entries in one table (2 500 rows x 100 steps, yaw chains from -40 to 40 rad: every branch of the
    algorithm, every quadrant of the reduction) against the host's sincos."""
    rng = np.random.default_rng(11)
    A, P, dt = 2500, 100, 0.05
    omega = rng.uniform(-8.0, 8.0, size=A)
    got = kh.trig_table(-0.37, omega, P, dt)
    libm = ctypes.CDLL("libm.so.6")
    libm.sincos.argtypes = [ctypes.c_double, ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    libm.sincos.restype = None
    yaw = np.full(A, -0.37)
    w = omega * dt                      # (elementwise IEEE products and sums: the kernel's chain)
    want = np.empty((P, A, 2))
    s, c = ctypes.c_double(), ctypes.c_double()
    for k in range(P):
        for r in range(A):
            libm.sincos(float(yaw[r]), ctypes.byref(s), ctypes.byref(c))
            want[k, r] = (c.value, s.value)
        yaw = yaw + w
    np.testing.assert_array_equal(got.view(np.uint64), want.view(np.uint64))


@pytest.mark.gpu
def test_table_outside_the_range_is_refused():
    with pytest.raises(Exception):
        kh.trig_table(2.0e8, [0.0], 4, 0.05)


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,scale,scene", [("cfg2", 0.5, "mid"), ("cfg5", 0.12, "survey"), ("cfg1", 1.0, "survey")])
@pytest.mark.parametrize("shape", ["cylinder", "box"])
def test_cycles_with_and_without_device_trig_equal_the_oracle(cfg, scale, scene, shape):
    import synthetic as syn

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import assert_cycle_equal, hip_context, hip_cycle, oracle_cycle

    inp = syn.make_controller_inputs(cfg, seed=5, scale=scale, scene=scene)
    if shape == "box":
        inp = dict(inp, robot=dict(shape=syn.BOX, dims=[0.5, 0.34, 0.3]))
    for yaw in (inp["state"][2], 2.6, -3.1):
        cur = dict(inp, state=(inp["state"][0], inp["state"][1], yaw, inp["state"][3]))
        o = oracle_cycle(cur)
        for opts in (dict(), dict(device_trig=0), dict(fused_cycle=2), dict(force_split=1), dict(fused_cycle=0)):
            ctx = hip_context(kh, cur)
            for k, v in opts.items():
                ctx.set_option(k, v)
            assert ctx.get_option("device_trig") == float(opts.get("device_trig", 1))
            assert_cycle_equal(o, hip_cycle(kh, cur, ctx=ctx))
            ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,scale,scan", [("cfg2", 0.5, False), ("cfg2", 1.0, False), ("cfg1", 1.0, False), ("cfg2", 0.5, True)])
def test_table_riding_in_the_sensor_launch(cfg, scale, scan):
    """A cycle that follows a sensor update finds its trig table formed inside that update's launch (same yaw,
    lattice, horizon); a cycle at another yaw, or behind a new lattice, forms its own.  Every cycle equals the
    oracle; `trig_rides` counts the launches that carried a table."""
    import synthetic as syn

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import assert_cycle_equal, hip_context, hip_cycle, oracle_cycle

    inp = syn.make_controller_inputs(cfg, seed=2, scale=scale, scene="mid")
    sc = None
    if scan:
        ang = np.linspace(-np.pi, np.pi, 720, endpoint=False)
        sc = (ang, 2.5 + 0.8 * np.sin(3 * ang))
    ctx = hip_context(kh, inp)
    rides = []
    for i, yaw in enumerate((0.0, 0.4, 0.4, -2.2, 3.0)):
        cur = dict(inp, state=(0.02 * i, -0.01 * i, yaw, 0.0))
        o = oracle_cycle(cur, scan=sc) if scan else oracle_cycle(cur)
        assert_cycle_equal(o, hip_cycle(kh, cur, scan=sc, ctx=ctx))     # set_points / set_scan, segment, samples, cycle
        rides.append(ctx.get_option("trig_rides"))
    assert rides[0] == 0 and rides[-1] >= 3, rides                       # (the first update knows no horizon yet)
    # the update's yaw is not the cycle's: the kernel forms its own rows
    st_a, st_b = (0.0, 0.0, 0.3, 0.0), (0.0, 0.0, -0.9, 0.0)
    ctx.set_weights(kh.make_weights(*inp["weights"]))
    if scan:
        ctx.set_scan(st_a, sc[0], sc[1], inp["max_range"])
    else:
        ctx.set_points(st_a, inp["points"], inp["max_range"])
    r = ctx.cycle(st_b, inp["P"])
    ctx2 = hip_context(kh, inp)
    ctx2.set_option("device_trig", 0)
    ctx2.set_weights(kh.make_weights(*inp["weights"]))
    if scan:
        ctx2.set_scan(st_a, sc[0], sc[1], inp["max_range"])
    else:
        ctx2.set_points(st_a, inp["points"], inp["max_range"])
    ctx2.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
    ctx2.set_samples(inp["vx"], inp["vy"], inp["omega"])
    r2 = ctx2.cycle(st_b, inp["P"])
    assert (r.found, r.index, r.raw_index, r.n_admissible) == (r2.found, r2.index, r2.raw_index, r2.n_admissible)
    assert np.float32(r.cost) == np.float32(r2.cost)
    # a new lattice between the update and the cycle
    ctx.set_points(st_a, inp["points"], inp["max_range"]) if not scan else ctx.set_scan(st_a, sc[0], sc[1], inp["max_range"])
    ctx.set_samples(inp["vx"] * 0.9, inp["vy"] * 0.9, inp["omega"] * 0.8)
    ctx2.set_samples(inp["vx"] * 0.9, inp["vy"] * 0.9, inp["omega"] * 0.8)
    ctx2.set_points(st_a, inp["points"], inp["max_range"]) if not scan else ctx2.set_scan(st_a, sc[0], sc[1], inp["max_range"])
    r, r2 = ctx.cycle(st_a, inp["P"]), ctx2.cycle(st_a, inp["P"])
    assert (r.found, r.index, r.raw_index, r.n_admissible) == (r2.found, r2.index, r2.raw_index, r2.n_admissible)
    assert np.float32(r.cost) == np.float32(r2.cost)
    ctx.close(); ctx2.close()


@pytest.mark.gpu
@pytest.mark.parametrize("opts", [dict(), dict(fused_cycle=2), dict(fused_cycle=0), dict(force_split=1)],
                         ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()) or "default")
def test_more_trig_rows_than_the_lds_table_holds(opts):
    """450 distinct omegas: rows beyond the 384 the kernels keep in LDS are read from global memory; the riding
    table of such a lattice (450 x 50 entries over 22 workgroups) as well."""
    import synthetic as syn

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import assert_cycle_equal, hip_context, hip_cycle, oracle_cycle

    inp = syn.make_controller_inputs("cfg2", seed=4, scale=0.5, scene="mid")
    vx, vy, om = syn.lattice_nonholonomic(8, 450)
    inp = dict(inp, vx=vx, vy=vy, omega=om)
    ctx = hip_context(kh, inp)
    for k, v in opts.items():
        ctx.set_option(k, v)
    assert ctx.get_option("device_trig") == 1.0
    for yaw in (0.1, -2.0, 0.1):
        cur = dict(inp, state=(0.0, 0.0, yaw, 0.0))
        assert_cycle_equal(oracle_cycle(cur), hip_cycle(kh, cur, ctx=ctx))
    assert ctx.get_option("trig_rows") == 450
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("opts", [dict(), dict(fused_cycle=0)],
                         ids=lambda d: ",".join(f"{k}={v}" for k, v in d.items()) or "default")
def test_more_axis_values_than_the_lds_tables_hold(opts):
    """An omni lattice with 150 vx and 70 vy values: the cycle kernel keeps the first 128 / 64 values of the axes in LDS
    (round 4) and reads the rest from global memory."""
    import synthetic as syn

    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import assert_cycle_equal, hip_context, hip_cycle, oracle_cycle

    inp = syn.make_controller_inputs("cfg5", seed=9, scale=0.1, scene="mid")
    vx, vy, om = syn.lattice_omni(150, 70, 5)
    inp = dict(inp, vx=vx, vy=vy, omega=om)
    ctx = hip_context(kh, inp)
    for k, v in opts.items():
        ctx.set_option(k, v)
    for yaw in (0.3, -1.1):
        cur = dict(inp, state=(0.0, 0.0, yaw, 0.0))
        assert_cycle_equal(oracle_cycle(cur), hip_cycle(kh, cur, ctx=ctx))
    ctx.close()
