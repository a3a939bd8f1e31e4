"""CPU: the threaded oracle helpers the full-size GPU parity tests use
(ko.full_cycle, ko.costs_mt) must reproduce the serial oracle -- the
restatement of trajectory_sampler.cpp:118-179 + cost_evaluator.cpp:49-109 that
the reference's known-answer tests pin (tests/test_oracle_kat.py) -- for any
thread count: same admissible set, floats, per-sample costs and first minimum."""
import numpy as np
import pytest

import synthetic as syn
from oracle import ko

from helpers import oracle_cycle, oracle_cycle_mt


@pytest.mark.parametrize("name,scale,scene", [("cfg1", 1.0, "survey"), ("cfg2", 0.25, "survey"),
                                              ("cfg2", 0.2, "mid"), ("cfg3", 0.06, "mid"),
                                              ("cfg5", 0.08, "mid"), ("cfg2", 0.1, "open")])
@pytest.mark.parametrize("threads", [1, 3, 8])
def test_full_cycle_equals_serial_oracle(name, scale, scene, threads):
    inp = syn.make_controller_inputs(name, seed=2, scale=scale, scene=scene)
    a = oracle_cycle(inp)
    b = oracle_cycle_mt(inp, threads=threads)
    np.testing.assert_array_equal(a["raw"], b["raw"])
    np.testing.assert_array_equal(a["px"].view(np.uint32), b["px"].view(np.uint32))
    np.testing.assert_array_equal(a["py"].view(np.uint32), b["py"].view(np.uint32))
    np.testing.assert_array_equal(a["costs"].view(np.uint32), b["costs"].view(np.uint32))
    assert a["index"] == b["index"]
    if a["index"] >= 0:
        assert np.float32(a["cost"]) == np.float32(b["cost"])
    if scene == "open":
        assert len(a["raw"]) == len(inp["vx"])


def test_full_cycle_ties_go_to_the_lowest_index():
    inp = syn.make_controller_inputs("cfg1", seed=2)
    # the same velocity three times: three equal costs
    k = 37
    for key in ("vx", "vy", "omega"):
        inp[key] = np.concatenate([inp[key], inp[key][k:k + 1], inp[key][k:k + 1]])
    a = oracle_cycle(inp)
    b = oracle_cycle_mt(inp, threads=4)
    assert a["index"] == b["index"]
    np.testing.assert_array_equal(a["costs"].view(np.uint32), b["costs"].view(np.uint32))


def test_costs_mt_equals_min_trajectory_cost():
    rng = np.random.default_rng(3)
    N, P, S, O = 200, 23, 150, 300
    px = (rng.random((N, P)) * 6 - 1).astype(np.float32)
    py = (rng.random((N, P)) * 4 - 2).astype(np.float32)
    vel = [(rng.random((N, P - 1)) * 2 - 1).astype(np.float32) for _ in range(3)]
    seg, acc = syn.arc_segment(S, radius=4.0, spacing=0.02)
    obs = (rng.random((O, 2)) * 8 - 3).astype(np.float32)
    ci = ko.CostInputs(seg, 0, acc, 7.5, obs, np.float32(10.0) / np.float32(3.0), (2.0, 0.0, 3.0),
                       ko.make_weights(0.7, 1.3, 2.0, 0.5, 0.25))
    for v in (None, vel):
        i0, c0, k0 = ko.min_trajectory_cost(ci, px, py, v)
        i1, c1, k1 = ko.costs_mt(ci, px, py, v, threads=5)
        assert i0 == i1 and np.float32(c0) == np.float32(c1)
        np.testing.assert_array_equal(k0.view(np.uint32), k1.view(np.uint32))
