"""Sharding by rule (kc_dwa_set_shard_rule: contiguous blocks, or dealt by trig row) and the single
collective of a sharded cycle (kc_dwa_cycle_sharded: ONE all-reduce of [key, error word, every rank's
admissible bitmap]).  One process, one GPU: the W shares run one after the other on W contexts'
worth of state; the multi-process run of the same protocol is tests/test_shm_ranks.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import kompass_hip as kh  # noqa: E402
import sharding  # noqa: E402
import synthetic as syn  # noqa: E402

from helpers import hip_context, oracle_cycle, oracle_cycle_mt  # noqa: E402


def _prepare(ctx, inp):
    st = inp["state"]
    ctx.set_weights(kh.make_weights(*inp["weights"]))
    ctx.set_points(st, inp["points"], inp["max_range"])
    ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])


def _rows(inp):
    _, rows = np.unique(np.asarray(inp["omega"], np.float64) + 0.0, return_inverse=True)
    return rows


@pytest.mark.parametrize("name,scale,seed", [("cfg2", 0.25, 3), ("cfg5", 0.08, 4), ("cfg1", 1.0, 5)])
@pytest.mark.parametrize("world", [2, 3, 8])
@pytest.mark.parametrize("mode", [kh.SHARD_BLOCKS, kh.SHARD_ROWS])
def test_shares_by_rule_cover_the_list_and_merge_to_the_oracle(name, scale, seed, world, mode):
    inp = syn.make_controller_inputs(name, seed=seed, scale=scale)
    o = oracle_cycle(inp)
    n = len(inp["vx"])
    owner = kh.shard_plan(_rows(inp), world, mode)
    ctx = hip_context(kh, inp)
    _prepare(ctx, inp)
    st = inp["state"]
    raws, costs, keys, rows_needed = [], [], [], []
    for r in range(world):
        ctx.set_shard_rule(r, world, mode)
        ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])   # the FULL list on every rank
        mine = np.nonzero(owner == r)[0]
        assert int(ctx.get_option("shard_samples")) == len(mine)
        rows_needed.append(int(ctx.get_option("trig_rows")))
        res = ctx.cycle(st, inp["P"])
        assert res.n_samples == len(mine)
        px, py, raw, c = ctx.get_samples(with_costs=True)
        assert len(raw) == res.n_admissible
        assert np.isin(raw, mine).all()            # global ids, all of them this rank's
        raws.append(raw.copy()); costs.append(c.copy())
        if res.found:
            assert ctx.owns_sample(res.raw_index)
            keys.append(sharding.key_pack(res.cost, res.raw_index))
            bx, by, bv = ctx.get_best()
            k = int(np.nonzero(o["raw"] == res.raw_index)[0][0])
            np.testing.assert_array_equal(bx, o["px"][k])
            np.testing.assert_array_equal(by, o["py"][k])
            vx, vy, om = ctx.get_sample_velocity(res.raw_index)
            assert (vx, vy, om) == (inp["vx"][res.raw_index], inp["vy"][res.raw_index], inp["omega"][res.raw_index])
    raw_all = np.concatenate(raws)
    order = np.argsort(raw_all, kind="stable")
    np.testing.assert_array_equal(raw_all[order], o["raw"])
    np.testing.assert_array_equal(np.concatenate(costs)[order].view(np.uint32), o["costs"].view(np.uint32))
    found, cost, raw_win = sharding.key_unpack(min(keys)) if keys else (False, 0.0, -1)
    assert found == (o["index"] >= 0)
    if found:
        assert raw_win == int(o["raw"][o["index"]]) and np.float32(cost) == np.float32(o["cost"])
        total = 0
        for r in range(world):
            ctx.set_shard_rule(r, world, mode)
            ctx.rollout(st, inp["P"])
            total += ctx.count_admissible_before(raw_win)
        assert total == o["index"]
    if mode == kh.SHARD_ROWS and name != "cfg1":
        # what the rule is for: a rank evaluates about 1 / W of the host's trig table
        full_rows = len(set(_rows(inp)))
        assert max(rows_needed) <= -(-full_rows // world) + 1
    # the rule off again: the context owns the whole list
    ctx.set_shard_rule(0, 1, -1)
    res = ctx.cycle(st, inp["P"])
    assert res.n_admissible == len(o["raw"]) and res.index == o["index"]
    ctx.close()


@pytest.mark.parametrize("name,scene", [("cfg3", "mid"), ("cfg5", "mid")])
def test_eight_row_dealt_shares_at_baseline_size(name, scene):
    """BASELINE cfg3 / cfg5 split over 8 GPUs by trig row: 8 sequential shares on one context ==
    unsharded == oracle (admissible set, every cost, winner, reference-numbered index)."""
    inp = syn.make_controller_inputs(name, seed=0, scene=scene)
    o = oracle_cycle_mt(inp)
    world = 8
    owner = kh.shard_plan(_rows(inp), world, kh.SHARD_ROWS)
    ctx = hip_context(kh, inp)
    _prepare(ctx, inp)
    st = inp["state"]
    raws, costs, keys, trig_rows = [], [], [], []
    for r in range(world):
        ctx.set_shard_rule(r, world, kh.SHARD_ROWS)
        ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
        trig_rows.append(int(ctx.get_option("trig_rows")))
        res = ctx.cycle(st, inp["P"])
        assert int(ctx.get_option("last_cycle_single_launch")) == 1   # n / 8 <= 8192: one launch
        _, _, raw, c = ctx.get_samples(with_costs=True, with_paths=False)
        assert (owner[raw] == r).all()
        raws.append(raw.copy()); costs.append(c.copy())
        if res.found:
            keys.append(sharding.key_pack(res.cost, res.raw_index))
    raw_all = np.concatenate(raws)
    order = np.argsort(raw_all, kind="stable")
    np.testing.assert_array_equal(raw_all[order], o["raw"])
    np.testing.assert_array_equal(np.concatenate(costs)[order].view(np.uint32), o["costs"].view(np.uint32))
    found, cost, raw_win = sharding.key_unpack(min(keys))
    assert found and raw_win == int(o["raw"][o["index"]]) and np.float32(cost) == np.float32(o["cost"])
    full_rows = len(set(_rows(inp)))
    assert max(trig_rows) <= -(-full_rows // world) + 1, (trig_rows, full_rows)
    ctx.close()


def test_world_of_one_through_rccl_with_a_rule():
    """The single-collective protocol through the real ncclAllReduce (a world of one): index and
    n_admissible come out of the exchanged bitmaps, no second collective."""
    uid = kh.comm_unique_id()
    comm = kh.Comm(0, 1, uid, device=0)
    assert comm.transport == "rccl"
    for name, scale, seed in [("cfg2", 0.25, 1), ("cfg5", 0.08, 2)]:
        inp = syn.make_controller_inputs(name, seed=seed, scale=scale)
        o = oracle_cycle(inp)
        for mode in (kh.SHARD_BLOCKS, kh.SHARD_ROWS, None):
            for fused in (1, 0):
                ctx = hip_context(kh, inp)
                ctx.set_option("fused_cycle", fused)
                _prepare(ctx, inp)
                if mode is not None:
                    ctx.set_shard_rule(0, 1, mode)
                ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
                for rep in range(3):
                    r = ctx.cycle_sharded(comm, inp["state"], inp["P"])
                    assert r.found and r.raw_index == int(o["raw"][o["index"]])
                    assert np.float32(r.cost) == np.float32(o["cost"])
                    assert r.n_admissible == len(o["raw"]) and r.index == o["index"]
                    bx, by, _ = ctx.get_best()
                    np.testing.assert_array_equal(bx, o["px"][o["index"]])
                ctx.close()
    comm.close()


def test_a_rank_without_samples_still_serves_the_full_list():
    """world greater than the number of dealt rows -- some rank's KC_SHARD_ROWS share is EMPTY.
    That rank still reports the FULL window from kc_dwa_sample_window, answers velocity look-ups for any
    global id, owns nothing, and the rule can be taken off again."""
    inp = syn.make_controller_inputs("cfg1", seed=7)
    vx = np.array([0.1, 0.2, 0.3, 0.1, 0.2], np.float64)
    vy = np.zeros(5)
    om = np.array([-0.2, 0.0, 0.2, -0.2, 0.0], np.float64)
    world = 8
    _, rows = np.unique(om + 0.0, return_inverse=True)
    owner = kh.shard_plan(rows, world, kh.SHARD_ROWS)
    empty = [r for r in range(world) if not (owner == r).any()]
    assert empty
    ctx = hip_context(kh, inp)
    _prepare(ctx, inp)
    for r in range(world):
        ctx.set_shard_rule(r, world, kh.SHARD_ROWS)
        ctx.set_samples(vx, vy, om)
        mine = np.nonzero(owner == r)[0]
        assert int(ctx.get_option("shard_samples")) == len(mine)
        for g in range(5):
            assert ctx.get_sample_velocity(g) == (vx[g], vy[g], om[g])
            assert ctx.owns_sample(g) == (g in mine)
        res = ctx.cycle(inp["state"], inp["P"])
        assert res.n_samples == len(mine)
        if r in empty:
            assert not res.found and res.n_admissible == 0
    # a window under the rule on a rank with nothing: the full count comes back
    r = empty[0]
    ctx.set_shard_rule(r, world, kh.SHARD_ROWS)
    ctx.set_samples(vx, vy, om)
    wvx, wvy, wom = ctx.sample_window(kh.DIFFERENTIAL_DRIVE, kh.make_limits(), (0.2, 0.0, 0.0), 2, 3)
    n_full = len(wvx)
    assert n_full > 0 and ctx.sample_window(kh.DIFFERENTIAL_DRIVE, kh.make_limits(), (0.2, 0.0, 0.0), 2, 3,
                                            want_list=False) == n_full
    _, wrows = np.unique(np.asarray(wom) + 0.0, return_inverse=True)
    wowner = kh.shard_plan(wrows, world, kh.SHARD_ROWS)
    assert int(ctx.get_option("shard_samples")) == int((wowner == r).sum())
    assert ctx.get_sample_velocity(n_full - 1) == (wvx[-1], wvy[-1], wom[-1])
    ctx.set_shard_rule(0, 1, -1)
    ctx.set_samples(vx, vy, om)
    assert int(ctx.get_option("shard_samples")) == 5
    ctx.close()
