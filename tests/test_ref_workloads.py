"""The reference's two published benchmark workloads at their real size
(src/kompass_cpp/benchmarks/benchmark_runner.cpp:152-217; the only points with a
published number): every per-trajectory cost of
CostEvaluator_5k_Trajs (5001 x 1000 points, P = 1000 is ten times what the
controller configs use) and every cell of Mapper_Dense_400x400, against the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import kompass_hip as kh  # noqa: E402
import synthetic as syn  # noqa: E402
from oracle import ko  # noqa: E402


@pytest.fixture(scope="module")
def cost5k():
    import bench

    w = bench.ref_cost5k_inputs()
    ci = ko.CostInputs(w["seg"], w["s0"], w["acc"], w["total"], None, np.float32(10.0) / np.float32(3.0),
                       w["acc_limits"], ko.make_weights(*w["weights"]))
    oi, oc, ocosts = ko.costs_mt(ci, w["px"], w["py"], w["vel"])
    return w, oi, oc, ocosts


@pytest.mark.parametrize("kernel", [0, 1, 2], ids=["auto", "workgroup-per-sample", "wavefront-per-sample"])
def test_cost5k_every_cost_equals_oracle(cost5k, kernel):
    w, oi, oc, ocosts = cost5k
    N, P = w["px"].shape
    assert (N, P, len(w["seg"])) == (5001, 1000, 1000)
    ctx = kh.DwaContext(syn.CYLINDER, [0.1, 0.4], max_samples=N, max_points=P, max_segment=len(w["seg"]),
                        acc_limits=w["acc_limits"])
    ctx.set_option("cost_kernel", kernel)
    ctx.set_weights(kh.make_weights(*w["weights"]))
    ctx.set_tracked_segment(w["seg"], w["acc"][w["s0"]:w["s0"] + len(w["seg"])], w["total"])
    ctx.cost_upload(w["px"], w["py"], w["vel"])
    for _ in range(2):   # resident: the second call sees the same samples
        r, costs = ctx.cost_evaluate_resident()
        np.testing.assert_array_equal(costs.view(np.uint32), ocosts.view(np.uint32))
        assert r.found and r.index == oi and np.float32(r.cost) == np.float32(oc)
        assert r.n_admissible == N
    r2, costs2 = ctx.cost_evaluate(w["px"], w["py"], w["vel"])   # the one-call form
    np.testing.assert_array_equal(costs2.view(np.uint32), ocosts.view(np.uint32))
    ctx.close()


@pytest.mark.parametrize("group", [1, 4, 16])
@pytest.mark.parametrize("weights", [(1, 1, 0, 1, 1), (0, 0, 0, 1, 0), (0, 0, 0, 0, 1)], ids=["all", "smoothness", "jerk"])
def test_cost5k_velocity_sums_by_group(cost5k, group, weights):
    """Option velocity_group: the ordered smoothness / jerk sums inside the cost kernel (1 sample per wavefront)
    or by velocity_sums_kernel (4 / 16 samples per wavefront, DPP rotation inside 16- / 4-lane groups) --
    the same bits; also with only one of the two costs asked for (the pass then runs one kind)."""
    w, oi, oc, ocosts = cost5k
    N, P = w["px"].shape
    if weights != tuple(w["weights"]):
        ci = ko.CostInputs(w["seg"], w["s0"], w["acc"], w["total"], None, np.float32(10.0) / np.float32(3.0),
                           w["acc_limits"], ko.make_weights(*weights))
        oi, oc, ocosts = ko.costs_mt(ci, w["px"], w["py"], w["vel"])
    ctx = kh.DwaContext(syn.CYLINDER, [0.1, 0.4], max_samples=N, max_points=P, max_segment=len(w["seg"]),
                        acc_limits=w["acc_limits"])
    assert ctx.get_option("velocity_group") == 0
    ctx.set_option("velocity_group", group)
    ctx.set_weights(kh.make_weights(*weights))
    ctx.set_tracked_segment(w["seg"], w["acc"][w["s0"]:w["s0"] + len(w["seg"])], w["total"])
    ctx.cost_upload(w["px"], w["py"], w["vel"])
    for beside in (1, 0):   # the pass on a second stream beside the cost kernel (+ velocity_finish_kernel), or in front of it
        ctx.set_option("velocity_beside", beside)
        for _ in range(2):
            r, costs = ctx.cost_evaluate_resident()
            np.testing.assert_array_equal(costs.view(np.uint32), ocosts.view(np.uint32))
            assert r.found and r.index == oi and np.float32(r.cost) == np.float32(oc)
            assert r.n_admissible == N
    with pytest.raises(IndexError):   # KC_ERR_RANGE
        ctx.set_option("velocity_group", 8)
    ctx.close()


def test_mapper400_grid_equals_oracle():
    g = syn.REF_MAPPER400
    ang, rng = syn.dense_scan(g["beams"], 1.0)
    want = ko.scan_to_grid(g["height"], g["width"], g["res"], (0, 0, 0), 0.0, ang, rng)
    m = kh.MapperContext(g["height"], g["width"], g["res"], (0, 0, 0), 0.0, g["beams"])
    np.testing.assert_array_equal(m.scan_to_grid(ang, rng), want)
    assert (want == 100).sum() > 2500 and (want == 0).sum() > 20000   # every beam ends inside the grid
    m.close()
