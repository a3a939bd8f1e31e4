"""Full-size parity: every BASELINE.json controller config at its real size
(cfg2 8192 x 50, cfg3 32768 x 100, cfg5 65536 x 50 with all five weights)
against the CPU oracle -- admissible set, every float of every admissible
path, every per-sample cost, winner index and cost, bit for bit -- and the
multi-GPU decomposition replayed as 8 sequential shards on one context.

The oracle scores the samples independently on all host cores
(oracle/ko.full_cycle, pinned to the serial oracle by
tests/test_oracle_full_cycle.py); the reference's own loop is
trajectory_sampler.cpp:118-179 + cost_evaluator.cpp:49-109.

Scenes: SURVEY 8(d)'s clutter ("survey") kills 95 % of cfg2 and all of cfg3,
so the configs also run on a thinned scene ("mid", about half admissible)
and cfg2 in open space (every sample admissible).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import kompass_hip as kh  # noqa: E402
import synthetic as syn  # noqa: E402

from helpers import assert_cycle_equal, hip_context, hip_cycle, oracle_cycle_mt  # noqa: E402

FULL = [("cfg2", "survey", 8192), ("cfg2", "mid", 8192), ("cfg2", "open", 8192),
        ("cfg3", "survey", 32768), ("cfg3", "mid", 32768),
        ("cfg5", "survey", 65536), ("cfg5", "mid", 65536)]

_oracle_cache = {}


def _oracle(name, scene):
    key = (name, scene)
    if key not in _oracle_cache:
        inp = syn.make_controller_inputs(name, seed=0, scene=scene)
        _oracle_cache[key] = (inp, oracle_cycle_mt(inp))
    return _oracle_cache[key]


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    assert kh.device_count() >= 1, "no HIP device visible: the -m gpu tests need an MI355X"


@pytest.mark.parametrize("mode", [1, 2], ids=["auto", "single-launch-forced"])
@pytest.mark.parametrize("name,scene,n", FULL, ids=[f"{a}-{b}" for a, b, _ in FULL])
def test_full_size_cycle_equals_oracle(name, scene, n, mode):
    """mode 1: what kc_dwa_cycle picks by itself (one launch up to 8192 samples per GPU, three
    kernels beyond); mode 2: the single launch at every size."""
    inp, o = _oracle(name, scene)
    assert len(inp["vx"]) == n
    ctx0 = hip_context(kh, inp)
    ctx0.set_option("fused_cycle", mode)
    h = hip_cycle(kh, inp, ctx=ctx0)
    assert ctx0.get_option("last_cycle_single_launch") == (1 if (mode == 2 or n <= 8192) else 0)   # first cycle: no
    # admissible count of a previous cycle to go by
    assert h["res"]["n_samples"] == n
    if scene == "open":
        assert len(o["raw"]) == n
    if (name, scene) == ("cfg3", "survey"):
        assert len(o["raw"]) == 0          # the survey scene leaves nothing at a 10 s horizon
    elif scene != "open":
        assert 0 < len(o["raw"]) < n
    assert_cycle_equal(o, h)
    # a second cycle on the same context (dilated masks now built by dilate_kernel,
    # cost kernel chosen from the admissible count of the first)
    ctx = h["ctx"]
    r = ctx.cycle(inp["state"], inp["P"])
    px, py, raw, costs = ctx.get_samples(with_costs=True)
    h2 = dict(px=px, py=py, raw=raw, costs=costs, res=r.as_dict())
    if r.found:
        h2["best"] = ctx.get_best()
    assert_cycle_equal(o, h2)
    ctx.close()


@pytest.mark.parametrize("name,scene", [("cfg3", "mid"), ("cfg5", "mid"), ("cfg2", "survey")])
def test_eight_sequential_shards_equal_unsharded_and_oracle(name, scene):
    """BASELINE cfg3 / cfg5 are one fixed batch split over 8 GPUs:
    replay the 8 shards one after the other on one context.  Union of the
    admissible sets == oracle's, per-sample costs equal, min over the packed
    shard keys == oracle's winner, compacted index rebuilt from the per-shard
    counts (kc_dwa_count_admissible_before) == oracle's index."""
    import sharding

    inp, o = _oracle(name, scene)
    n = len(inp["vx"])
    ctx = hip_context(kh, inp)   # shards of n / 8 <= 8192 samples: the single-launch cycle
    st = inp["state"]
    ctx.set_weights(kh.make_weights(*inp["weights"]))
    ctx.set_points(st, inp["points"], inp["max_range"])
    ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
    ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
    world = 8
    keys, raws, costs, pxs = [], [], [], []
    for g in range(world):
        first, count = sharding.shard_range(n, g, world)
        ctx.set_shard(first, count)
        r = ctx.cycle(st, inp["P"])
        assert r.n_samples == count
        keys.append(sharding.key_pack(r.cost, r.raw_index) if r.found else sharding.KEY_NONE)
        px, py, raw, c = ctx.get_samples(with_costs=True)
        assert len(raw) == r.n_admissible
        assert ((raw >= first) & (raw < first + count)).all()
        raws.append(raw.copy()); costs.append(c.copy()); pxs.append(px.copy())
    raw_all = np.concatenate(raws)
    np.testing.assert_array_equal(raw_all, o["raw"])
    np.testing.assert_array_equal(np.concatenate(costs).view(np.uint32), o["costs"].view(np.uint32))
    np.testing.assert_array_equal(np.concatenate(pxs).view(np.uint32), o["px"].view(np.uint32))
    best = min(keys)                       # the all-reduce(min) of the 8-byte keys
    found, cost, raw_win = sharding.key_unpack(best)
    assert found == (o["index"] >= 0)
    if found:
        assert raw_win == int(o["raw"][o["index"]])
        assert np.float32(cost) == np.float32(o["cost"])
        # the reference-numbered index: admissible samples in front of the winner, summed over shards
        total = 0
        for g in range(world):
            first, count = sharding.shard_range(n, g, world)
            ctx.set_shard(first, count)
            ctx.rollout(st, inp["P"])
            total += ctx.count_admissible_before(raw_win)
        assert total == o["index"]
    # and the unsharded context state comes back
    ctx.set_shard(0, n)
    r = ctx.cycle(st, inp["P"])
    assert r.n_admissible == len(o["raw"]) and r.index == o["index"]
    ctx.close()
