"""N > 1 path on CPU: world_size-2 gloo run of the sharded cycle (contiguous
sample blocks, ONE all-reduce(min) of the packed key, compacted index rebuilt
with one all-reduce(sum)) must reproduce the unsharded oracle result."""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import sharding
import synthetic as syn
from helpers import oracle_cycle

ROOT = Path(__file__).resolve().parent.parent


def test_shard_range_partitions():
    for n in [0, 1, 7, 8192, 65250]:
        for w in [1, 2, 3, 8]:
            parts = [sharding.shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and sum(c for _, c in parts) == n
            for (f0, c0), (f1, _) in zip(parts, parts[1:]):
                assert f0 + c0 == f1
            assert max(c for _, c in parts) - min(c for _, c in parts) <= 1


def test_key_order_is_lowest_cost_then_lowest_index():
    rng = np.random.default_rng(0)
    costs = np.float32(rng.random(200) * 3 - 0.5)
    costs[17] = costs[5]  # tie
    costs[33] = np.float32(0.0); costs[34] = np.float32(-0.0)
    keys = [sharding.key_pack(c, i) for i, c in enumerate(costs)]
    best = min(range(200), key=lambda i: keys[i])
    want = min(range(200), key=lambda i: (float(costs[i]), i))
    assert best == want
    found, cost, idx = sharding.key_unpack(keys[best])
    assert found and idx == best and np.float32(cost) == costs[best]
    assert sharding.key_pack(np.inf, 3) == sharding.KEY_NONE
    assert sharding.key_unpack(sharding.KEY_NONE) == (False, 0.0, -1)


@pytest.mark.timeout(300)
def test_two_rank_gloo_matches_unsharded(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = tmp_path / "res.json"
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           str(ROOT / "tests" / "_gloo_worker.py"), str(out)]
    subprocess.run(cmd, check=True, env=env, timeout=280, capture_output=True)
    got = json.loads(out.read_text())
    cases = [("cfg1", 1.0, 1), ("cfg2", 0.25, 2), ("cfg5", 0.08, 3)]
    assert len(got) == 4 * len(cases)
    for i, (name, scale, seed) in enumerate(cases):
        o = oracle_cycle(syn.make_controller_inputs(name, seed=seed, scale=scale))
        # [0] round-1 protocol (key all-reduce + count all-reduce); [1], [2] the product's single
        # all-reduce of the exchange record with block / trig-row shares, merged by kc_shard_merge
        for g in got[4 * i: 4 * i + 3]:
            assert g["found"] == (o["index"] >= 0)
            assert g["index"] == o["index"], (name, g)
            assert g["raw"] == int(o["raw"][o["index"]])
            assert np.float32(g["cost"]) == np.float32(o["cost"])
            if "n_admissible" in g:
                assert g["n_admissible"] == len(o["raw"])
        assert got[4 * i + 3]["failed"] is True  # one rank's error word fails the cycle on every rank


def test_shard_plan_rules():
    """kc_shard_plan (pure host): blocks are shard_range; rows deal whole trig rows round-robin, a heavy
    row (omni: omega = 0) sample by sample; every sample has exactly one owner."""
    import kompass_hip as kh

    for name, scale in [("cfg2", 0.25), ("cfg3", 0.05), ("cfg5", 0.08)]:
        inp = syn.make_controller_inputs(name, seed=0, scale=scale)
        n = len(inp["vx"])
        _, rows = np.unique(np.asarray(inp["omega"], np.float64) + 0.0, return_inverse=True)
        for w in (1, 2, 3, 8):
            ob = kh.shard_plan(rows, w, kh.SHARD_BLOCKS)
            for r in range(w):
                f, c = sharding.shard_range(n, r, w)
                assert (ob[f:f + c] == r).all()
            orow = kh.shard_plan(rows, w, kh.SHARD_ROWS)
            assert orow.min() >= 0 and orow.max() < w
            counts = np.bincount(orow, minlength=w)
            per_row = np.bincount(rows)
            heavy = per_row > 2 * -(-n // len(per_row))
            # light rows are never split; shares are balanced to within one light row
            for a in np.nonzero(~heavy)[0]:
                assert len(set(orow[rows == a])) == 1
            assert counts.max() - counts.min() <= per_row[~heavy].max() + 1
            # trig rows a rank needs: about 1 / w of them (+ the heavy ones)
            need = [len(set(rows[orow == r])) for r in range(w)]
            assert max(need) <= -(-int((~heavy).sum()) // w) + int(heavy.sum())


def test_plan_and_merge_with_more_ranks_than_rows():
    """with more ranks than dealt trig rows some ranks own NOTHING; the plan still gives every sample
    one owner and the merge of an exchange record with empty shares names the winner in the FULL numbering."""
    import kompass_hip as kh

    rows = np.array([0, 1, 2, 0, 1], np.int32)       # 5 samples, 3 trig rows
    world = 8
    owner = kh.shard_plan(rows, world, kh.SHARD_ROWS)
    counts = np.bincount(owner, minlength=world)
    assert counts.sum() == len(rows) and (counts == 0).sum() >= world - len(rows)
    rw = 1                                            # 64 share-local ids per rank
    adm = np.array([1, 0, 1, 1, 1], bool)             # sample 1 hit something
    costs = np.float32([0.7, 0.1, 0.5, 0.9, 0.5])     # winner: sample 2 (tie with 4: lowest index)
    rec = np.full(2 + world * rw, np.iinfo(np.int64).max, np.int64)
    rec[1] = 0
    for r in range(world):
        mine = np.nonzero(owner == r)[0]
        bits = 0
        for local, g in enumerate(mine):
            if adm[g]:
                bits |= 1 << local
                rec[0] = min(rec[0], sharding.key_pack(costs[g], int(g)))
        rec[2 + r * rw] = bits                         # an empty share contributes 0: nothing admissible
    res = kh.shard_merge(rec, rw, world, kh.SHARD_ROWS, owner, len(rows))
    assert res.found and res.raw_index == 2 and np.float32(res.cost) == np.float32(0.5)
    assert res.n_admissible == int(adm.sum()) and res.index == 1   # admissible samples in front of 2: {0}
    assert res.n_samples == len(rows)
