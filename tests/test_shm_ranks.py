"""kc_dwa_cycle_sharded with MORE THAN ONE RANK on a one-GPU box: W processes, each with its own
context on device 0, exchange through the library's shared-memory transport (kc_comm_create_shm;
RCCL refuses two ranks on one device).  Every rank must return the unsharded oracle's result -- found,
cost bits, raw index, the reference-numbered index and the global admissible count out of the ONE
exchange -- and an error on one rank must fail THAT cycle on every rank and leave the next cycle paired
up (a failure on one rank has to be every rank's failure of the SAME cycle)."""
import json
import os
import subprocess
import sys
import uuid
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import kompass_hip as kh  # noqa: E402
import synthetic as syn  # noqa: E402

from helpers import oracle_cycle  # noqa: E402

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "tests"))
from _shm_worker import custom_cost, poses  # noqa: E402


def _run(tmp_path, world, scenario, cfg, scale, seed, mode):
    name = uuid.uuid4().hex[:16]
    procs = []
    for r in range(world):
        env = dict(os.environ, KC_SHM_TIMEOUT_MS="60000", OMP_NUM_THREADS="1", OPENBLAS_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(ROOT / "tests" / "_shm_worker.py"), str(r), str(world), name,
                                       str(tmp_path), scenario, cfg, str(scale), str(seed), str(mode)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    import time

    t_end = time.time() + 240
    while time.time() < t_end and any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            break   # a rank died: the others would only wait for it
        time.sleep(0.2)
    for p in procs:
        if p.poll() is None:
            p.kill()
    errs = [p.communicate()[1] for p in procs]
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} rc {p.returncode}: {errs[r][-1500:]}"
    return [json.loads((tmp_path / f"rank{r}.json").read_text()) for r in range(world)]


def _oracles(cfg, scale, seed):
    inp = syn.make_controller_inputs(cfg, seed=seed, scale=scale)
    res = []
    for k in range(6):
        i2 = dict(inp, state=poses(inp, k))
        if k >= 3:
            i2 = dict(i2, vx=inp["vx"] * 0.97, vy=inp["vy"] * 0.97)
        res.append(oracle_cycle(i2))
    return res


def _check_cycle(rec, o, n_total):
    assert rec["ok"], rec
    assert rec["found"] == (o["index"] >= 0)
    assert rec["n_admissible"] == len(o["raw"]) and rec["n_samples"] == n_total
    if rec["found"]:
        assert rec["raw"] == int(o["raw"][o["index"]]) and rec["index"] == o["index"]
        assert np.float32(rec["cost"]) == np.float32(o["cost"])


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,cfg,scale,seed,mode", [
    (2, "cfg2", 0.25, 11, kh.SHARD_BLOCKS), (2, "cfg2", 0.25, 12, kh.SHARD_ROWS),
    (4, "cfg5", 0.08, 13, kh.SHARD_ROWS), (3, "cfg1", 1.0, 14, kh.SHARD_BLOCKS)])
def test_ranks_agree_with_the_unsharded_oracle(tmp_path, world, cfg, scale, seed, mode):
    got = _run(tmp_path, world, "plain", cfg, scale, seed, mode)
    ora = _oracles(cfg, scale, seed)
    n_total = len(syn.make_controller_inputs(cfg, seed=seed, scale=scale)["vx"])
    for k, o in enumerate(ora):
        owners = 0
        for r in range(world):
            _check_cycle(got[r][k], o, n_total)
            if got[r][k].get("owns"):
                owners += 1
                np.testing.assert_array_equal(np.float32(got[r][k]["best_x"]), o["px"][o["index"]])
        assert owners == (1 if o["index"] >= 0 else 0)   # exactly one rank holds the winner's row


@pytest.mark.timeout(600)
@pytest.mark.parametrize("world,cfg,scale,seed,mode", [(2, "cfg2", 0.25, 41, kh.SHARD_ROWS), (3, "cfg1", 1.0, 42, kh.SHARD_BLOCKS)])
def test_custom_costs_with_a_sharded_controller(tmp_path, world, cfg, scale, seed, mode):
    """VERDICT r3 item 6 / SURVEY 8e row 2: custom cost callbacks are host-side; with a sharded DWA every rank adds
    them to the device totals of its own admissible rows and the ranks exchange their bests
    (kc_dwa_exchange_best).  Every rank must return what ONE process gets that adds the callback to the oracle's
    per-sample totals (cost_evaluator.cpp:96-100: float = (double) total + weight * (double) cost, first strict
    minimum in generation order)."""
    got = _run(tmp_path, world, "custom", cfg, scale, seed, mode)
    ora = _oracles(cfg, scale, seed)
    n_total = len(syn.make_controller_inputs(cfg, seed=seed, scale=scale)["vx"])
    for k, o in enumerate(ora):
        best, arg = np.float32(np.finfo(np.float32).max), -1
        for i, (g, c) in enumerate(zip(o["raw"], o["costs"])):
            t = np.float32(np.float64(c) + 2.5 * np.float64(custom_cost(int(g))))
            if t < best:
                best, arg = t, i
        for r in range(world):
            rec = got[r][k]
            assert rec["ok"], rec
            assert rec["n_admissible"] == len(o["raw"]) and rec["n_samples"] == n_total
            assert rec["found"] == (arg >= 0)
            if arg >= 0:
                assert rec["raw"] == int(o["raw"][arg]) and rec["index"] == arg
                assert np.float32(rec["cost"]) == best


@pytest.mark.timeout(600)
def test_a_failure_before_the_exchange_is_collective(tmp_path):
    world, cfg, scale, seed = 3, "cfg2", 0.25, 31
    got = _run(tmp_path, world, "prefail", cfg, scale, seed, kh.SHARD_ROWS)
    ora = _oracles(cfg, scale, seed)
    n_total = len(syn.make_controller_inputs(cfg, seed=seed, scale=scale)["vx"])
    for r in range(world):
        assert not got[r][1]["ok"]
        if r == world - 1:
            assert "num_points" in got[r][1]["error"]        # its own, specific error
        else:
            assert "failed before the exchange" in got[r][1]["error"]
        for k in (0, 2, 3, 4, 5):
            _check_cycle(got[r][k], ora[k], n_total)
