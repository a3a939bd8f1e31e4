"""Bayesian mapper update + previous-grid warp.

The reference's own tests only print these grids (mapper_test.cpp:136-220), so
parity is UNPINNED by the reference: the CPU tests pin the oracle to closed
forms of the inverse sensor model and to the properties the source implies, the
GPU tests compare the HIP path with the oracle bit for bit.
"""
import json
from pathlib import Path

import numpy as np
import pytest

import synthetic as syn
from oracle import ko

# LocalMapper Bayesian ctor arguments of the reference benchmark
# (benchmark_runner.cpp:207-212): p_prior 0.6, p_occupied 0.9, p_empty 0.1,
# range_sure 0.1, range_max 20, wall_size 0.2
BENCH = dict(p_prior=0.6, p_occupied=0.9, p_empty=0.1, range_sure=0.1, range_max=20.0, wall_size=0.2)
# the first ctor's defaults (local_mapper.h:22-24)
DEFAULT = dict(p_prior=0.5, p_occupied=0.6, p_empty=0.4, range_sure=1.0, range_max=20.0, wall_size=0.2)


def _scan(n, scale, seed=5):
    ang, rng = syn.dense_scan(n, scale)
    r = np.random.default_rng(seed)
    rng = rng * (0.6 + 0.8 * r.random(n))
    rng[::17] = 0.0
    rng[::29] *= 10.0
    return ang, rng


# ---------------------------------------------------------------------------
# oracle (CPU)
# ---------------------------------------------------------------------------
def test_oracle_occupancy_is_the_plain_scan_and_untouched_cells_hold_the_prior():
    ang, rng = _scan(720, 1.0)
    m = ko.BayesMapper(200, 160, 0.05, (0.1, -0.05, 0), 0.3, **BENCH)
    g, p = m.scan_to_grid_baysian(ang, rng)
    np.testing.assert_array_equal(g, ko.scan_to_grid(200, 160, 0.05, (0.1, -0.05, 0), 0.3, ang, rng))
    assert (p[g == -1] == np.float32(0.6)).all()
    assert (p[g != -1] != np.float32(0.6)).any()
    assert np.isfinite(p).all() and p.min() >= 0.0 and p.max() <= 1.0


def test_oracle_sensor_model_closed_form():
    """prior 0.5 and previous 0.5 make both odds factors 1, so the cell value
    is the sensor probability itself: p_empty before (range - wall), p_occupied
    behind it, both pulled towards the prior beyond range_sure
    (local_mapper.cpp:106-125)."""
    res, rng_m = 0.1, 3.0
    m = ko.BayesMapper(100, 100, res, (0, 0, 0), 0.0, **DEFAULT)
    g, p = m.scan_to_grid_baysian([0.0], [rng_m])
    s = 49  # start cell = central cell = round(100 / 2) - 1
    for k in range(0, 31):
        d = np.float32(k) * np.float32(res)          # integer cell distance * resolution
        pf = 0.4 if d < np.float32(rng_m) - np.float32(0.2) else 0.6
        delta = 0.0 if d < 1.0 else 1.0
        want = pf + delta * ((float(d) - 1.0) / 20.0) * (0.5 - pf)
        assert abs(float(p[s + k, s]) - want) < 2e-6, (k, p[s + k, s], want)
    assert g[s + 30, s] == 100 and (g[s:s + 30, s] == 0).all()
    assert (p[:, s + 1] == np.float32(0.5)).all()


def test_oracle_last_beam_decides_a_shared_cell():
    """Two beams over the same cells: the second one's range is the one that
    counts (gridDataProb(pt) = newValue, local_mapper.cpp:199)."""
    m = ko.BayesMapper(100, 100, 0.1, (0, 0, 0), 0.0, **DEFAULT)
    _, p_ab = m.scan_to_grid_baysian([0.0, 0.0], [3.0, 1.5])
    _, p_b = m.scan_to_grid_baysian([0.0], [1.5])
    _, p_ba = m.scan_to_grid_baysian([0.0, 0.0], [1.5, 3.0])
    _, p_a = m.scan_to_grid_baysian([0.0], [3.0])
    s = 49
    np.testing.assert_array_equal(p_ab[s:s + 16, s], p_b[s:s + 16, s])
    np.testing.assert_array_equal(p_ab[s + 16:, s], p_a[s + 16:, s])   # only the long beam got there
    np.testing.assert_array_equal(p_ba[:, s], p_a[:, s])


def test_oracle_warp_matrix_inverts_the_reference_transform():
    m = ko.BayesMapper(120, 90, 0.05, (0, 0, 0), 0.0, **BENCH)
    pos, th = (0.4, -0.25), 0.35
    inv = m.warp_matrix(pos, th).astype(np.float64)
    c0, c1 = 120 // 2 - 1, 90 // 2 - 1
    cc0 = c0 + int(np.float32(pos[0]) / np.float32(0.05))
    cc1 = c1 + int(np.float32(pos[1]) / np.float32(0.05))
    c, s = np.cos(-th), np.sin(-th)
    fwd = np.array([[c, -s, 0.5 * 120 - cc1 + (cc0 * s - cc1 * c)],
                    [s, c, 0.5 * 90 - cc0 - (cc0 * c + cc1 * s)],
                    [0, 0, 1]])
    np.testing.assert_allclose(inv @ fwd, np.eye(3), atol=2e-5)


def test_oracle_warp_keeps_a_constant_grid_and_fills_with_the_prior():
    m = ko.BayesMapper(80, 80, 0.1, (0, 0, 0), 0.0, **BENCH)
    w = m.get_previous_grid_in_current_pose((0.3, 0.1), 0.2)
    np.testing.assert_allclose(w, 0.6, atol=1e-6)
    r = np.random.default_rng(2)
    prev = r.uniform(0.05, 0.95, (80, 80)).astype(np.float32)
    m.set_previous(prev)
    np.testing.assert_array_equal(m.previous(), prev)
    w = m.get_previous_grid_in_current_pose((0.3, 0.1), 0.2)
    inside = w != np.float32(0.6)
    assert inside.any() and (~inside).any()
    # bilinear values stay inside the range of the source grid
    assert w[inside].min() >= prev.min() - 1e-6 and w[inside].max() <= prev.max() + 1e-6


# ---------------------------------------------------------------------------
# HIP path vs oracle (GPU)
# ---------------------------------------------------------------------------
_CASES = [
    (640, 160, 0.05, (-14.0, 3.0, 0), 1.0, 1500, 2.5, BENCH),
    (333, 517, 0.02, (1.0, -2.0, 0), 2.2, 2048, 1.0, DEFAULT),
    (400, 400, 0.05, (0, 0, 0), 0.0, 3600, 1.0, BENCH),
    (200, 300, 0.1, (0.35, -0.2, 0.1), 0.6, 777, 1.0, DEFAULT),
    (101, 77, 0.07, (-0.5, 0.4, 0), -2.0, 360, 0.5, BENCH),
    (1000, 1000, 0.05, (0, 0, 0), 0.0, 4096, 4.0, BENCH),
]


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.gpu
@pytest.mark.parametrize("tiles", ["3", "1", "0"])  # hybrid (default) / all tiles / all beam-parallel
@pytest.mark.parametrize("H,W,res,pos,orient,n,scale,params", _CASES)
def test_bayes_scan_parity(H, W, res, pos, orient, n, scale, params, tiles, monkeypatch):
    import kompass_hip as kh
    monkeypatch.setenv("KC_MAPPER_TILES", tiles)
    ang, rng = _scan(n, scale)
    o = ko.BayesMapper(H, W, res, pos, orient, **params)
    m = kh.MapperContext(H, W, res, pos, orient, n)
    m.enable_bayes(**params)
    r = np.random.default_rng(11)
    for step in range(3):
        want_g, want_p = o.scan_to_grid_baysian(ang, rng)
        got_g, got_p = m.scan_to_grid_baysian(ang, rng)
        np.testing.assert_array_equal(got_g, want_g)
        np.testing.assert_array_equal(_bits(got_p), _bits(want_p))
        # next round: a non-constant previous grid and other ranges
        prev = r.uniform(0.02, 0.98, (H, W)).astype(np.float32)
        o.set_previous(prev)
        m.set_previous_prob(prev)
        rng = rng[::-1].copy()
    # the plain scan on the same context is unaffected by the Bayesian state
    np.testing.assert_array_equal(m.scan_to_grid(ang, rng), ko.scan_to_grid(H, W, res, pos, orient, ang, rng))
    m.close()


@pytest.mark.gpu
def test_bayes_fixture_scan_and_empty_scan():
    import kompass_hip as kh
    data = json.loads((Path(__file__).parent / "golden" / "laserscan_data.json").read_text())
    rng = np.array(data["ranges"], dtype=np.float64)
    ang = data["angle_min"] + np.arange(len(rng)) * data["angle_increment"]
    rng = np.clip(np.nan_to_num(rng, posinf=20.0), 0, 20.0)
    o = ko.BayesMapper(200, 200, 0.1, (0, 0, 0), 0.0, **DEFAULT)
    m = kh.MapperContext(200, 200, 0.1, (0, 0, 0), 0.0, len(rng))
    with pytest.raises(ValueError):
        m.scan_to_grid_baysian(ang, rng)       # not enabled yet
    m.enable_bayes(**DEFAULT)
    want_g, want_p = o.scan_to_grid_baysian(ang, rng)
    got_g, got_p = m.scan_to_grid_baysian(ang, rng)
    np.testing.assert_array_equal(got_g, want_g)
    np.testing.assert_array_equal(_bits(got_p), _bits(want_p))
    g, p = m.scan_to_grid_baysian(np.zeros(0), np.zeros(0))
    assert (g == -1).all() and (p == np.float32(0.5)).all()
    # and a scan after the empty one still matches
    got_g, got_p = m.scan_to_grid_baysian(ang, rng)
    np.testing.assert_array_equal(_bits(got_p), _bits(want_p))


@pytest.mark.gpu
@pytest.mark.parametrize("H,W,res", [(200, 200, 0.05), (151, 97, 0.1), (1000, 1000, 0.05)])
def test_warp_parity(H, W, res):
    import kompass_hip as kh
    r = np.random.default_rng(H + W)
    o = ko.BayesMapper(H, W, res, (0, 0, 0), 0.0, **BENCH)
    m = kh.MapperContext(H, W, res, (0, 0, 0), 0.0, 16)
    m.enable_bayes(**BENCH)
    np.testing.assert_array_equal(_bits(m.previous_prob()), _bits(o.previous()))
    ext = 0.5 * min(H, W) * res
    for pos, th in [((0.0, 0.0), 0.0), ((0.3 * ext, -0.2 * ext), 0.4), ((-0.9 * ext, 0.7 * ext), -2.5),
                    ((0.05, 0.02), 3.0), ((5 * ext, 5 * ext), 0.1)]:
        prev = r.uniform(0.02, 0.98, (H, W)).astype(np.float32)
        o.set_previous(prev)
        m.set_previous_prob(prev)
        want = o.get_previous_grid_in_current_pose(pos, th)
        m.get_previous_grid_in_current_pose(pos, th)
        np.testing.assert_array_equal(_bits(m.previous_prob()), _bits(want))
        # a second warp of the warped grid (the reference warps in place, call after call)
        want = o.get_previous_grid_in_current_pose(pos, -th)
        m.get_previous_grid_in_current_pose(pos, -th)
        np.testing.assert_array_equal(_bits(m.previous_prob()), _bits(want))


@pytest.mark.gpu
def test_mapping_loop_with_feedback():
    """warp -> scan -> feed the probabilities back, five steps of a moving
    robot; the device-side feedback copy equals uploading the same grid."""
    import kompass_hip as kh
    H = W = 300
    o = ko.BayesMapper(H, W, 0.05, (0.1, 0, 0), 0.2, **BENCH)
    m = kh.MapperContext(H, W, 0.05, (0.1, 0, 0), 0.2, 1024)
    m.enable_bayes(**BENCH)
    for step in range(5):
        ang, rng = _scan(1024, 1.0, seed=step)
        if step:
            o.get_previous_grid_in_current_pose((0.02 * step, -0.01 * step), 0.03 * step)
            m.get_previous_grid_in_current_pose((0.02 * step, -0.01 * step), 0.03 * step)
        want_g, want_p = o.scan_to_grid_baysian(ang, rng)
        got_g, got_p = m.scan_to_grid_baysian(ang, rng)
        np.testing.assert_array_equal(got_g, want_g)
        np.testing.assert_array_equal(_bits(got_p), _bits(want_p))
        o.set_previous(want_p)
        m.set_previous_prob(None)
    np.testing.assert_array_equal(_bits(m.previous_prob()), _bits(o.previous()))


@pytest.mark.gpu
def test_bayes_random_scenes():
    """Random grids, sensor poses (also outside the grid), beam sets (ordered and not) and ranges through
    the tiled Bayesian scan, against the oracle."""
    import kompass_hip as kh
    r = np.random.default_rng(4242)
    for case in range(30):
        H, W = int(r.integers(3, 500)), int(r.integers(3, 500))
        res = float(r.choice([0.02, 0.05, 0.1, 0.25]))
        ext = min(H, W) * res
        pos = (float(r.uniform(-0.6, 0.6) * ext), float(r.uniform(-0.6, 0.6) * ext), 0.0)
        orient = float(r.uniform(-3.2, 3.2))
        n = int(r.choice([1, 3, 64, 360, 1500, 4000]))
        ang = np.sort(r.uniform(-np.pi, np.pi, n)) if case % 3 else r.uniform(-7, 7, n)
        rng = r.uniform(0, 1.2 * ext, n) * r.choice([1.0, 1.0, 1.0, 0.0, 12.0], n)
        params = BENCH if case % 2 else DEFAULT
        o = ko.BayesMapper(H, W, res, pos, orient, **params)
        m = kh.MapperContext(H, W, res, pos, orient, n)
        m.enable_bayes(**params)
        prev = r.uniform(0.02, 0.98, (H, W)).astype(np.float32)
        o.set_previous(prev)
        m.set_previous_prob(prev)
        want_g, want_p = o.scan_to_grid_baysian(ang, rng)
        got_g, got_p = m.scan_to_grid_baysian(ang, rng)
        assert np.array_equal(got_g, want_g), (case, H, W, res, pos, orient, n, int((got_g != want_g).sum()))
        assert np.array_equal(_bits(got_p), _bits(want_p)), (case, H, W, res, pos, orient, n)
        m.close()
