"""One rank of tests/test_shm_ranks.py: a process with its own context on the (shared) GPU, the ranks
meeting in the library's shared-memory transport (kc_comm_create_shm).  Everything in
kc_dwa_cycle_sharded except the ncclAllReduce call itself runs exactly as on an 8-GPU node."""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kompass-core_amd"), str(ROOT / "tests")]

import kompass_hip as kh  # noqa: E402
import synthetic as syn  # noqa: E402


def custom_cost(raw):
    """The host callback of the `custom` scenario: a float32 cost per GLOBAL sample id."""
    return np.float32(((raw * 2654435761) % 1000) / 1000.0)


def poses(inp, k):
    x, y, yaw, sp = inp["state"]
    return (x + 0.01 * k, y - 0.005 * k, yaw + 0.02 * ((k % 5) - 2), sp)


def main():
    rank, world, name, out_dir, scenario, cfg, scale, seed, mode = sys.argv[1:10]
    rank, world, scale, seed, mode = int(rank), int(world), float(scale), int(seed), int(mode)
    inp = syn.make_controller_inputs(cfg, seed=seed, scale=scale)
    rb = inp["robot"]
    n, P = len(inp["vx"]), inp["P"]
    ctx = kh.DwaContext(rb["shape"], rb["dims"], (0, 0, 0), (0, 0, 0, 1), inp["octree_res"], inp["dt"],
                        max_samples=n, max_points=P, max_segment=len(inp["seg_xyz"]),
                        max_obstacles=max(len(inp["points"]), 16), acc_limits=inp["acc_limits"])
    comm = kh.Comm(rank, world, device=0, shm_name=name)
    assert comm.transport == "shm"
    ctx.set_weights(kh.make_weights(*inp["weights"]))
    ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
    ctx.set_shard_rule(rank, world, mode)
    ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
    out = []
    for k in range(6):
        st = poses(inp, k)
        ctx.set_points(st, inp["points"], inp["max_range"])
        if k == 3:  # a new list with the same pattern of trig rows (what a controller's next window is)
            ctx.set_samples(inp["vx"] * 0.97, inp["vy"] * 0.97, inp["omega"])
        p_call = P
        if scenario == "prefail" and k == 1 and rank == world - 1:
            p_call = P + 1  # beyond max_points: this rank fails BEFORE the exchange
        try:
            if scenario == "custom":
                # custom cost callbacks of a sharded DWA (cost_evaluator.cpp:96-100): the cycle of this rank's
                # share, the callback added on the host to the device totals of its own admissible rows in the
                # reference's rounding (float = (double) total + w * (double) c), its own first minimum into the
                # exchange (kc_dwa_exchange_best)
                status = 0
                found, best, braw = False, np.float32(np.finfo(np.float32).max), -1
                try:
                    ctx.cycle(st, p_call)
                    _, _, raw, costs = ctx.get_samples(with_costs=True, with_paths=False)
                    for g, c in zip(raw, costs):
                        t = np.float32(np.float64(c) + 2.5 * np.float64(custom_cost(int(g))))
                        if t < best:
                            found, best, braw = True, t, int(g)
                except (RuntimeError, IndexError, ValueError):
                    status = 1
                r = ctx.exchange_best(comm, found, best, braw, status=status)
            else:
                r = ctx.cycle_sharded(comm, st, p_call)
            rec = dict(ok=True, found=bool(r.found), cost=float(r.cost), raw=int(r.raw_index), index=int(r.index),
                       n_admissible=int(r.n_admissible), n_samples=int(r.n_samples),
                       owns=bool(r.found and ctx.owns_sample(r.raw_index)))
            if rec["owns"]:
                bx, by, _ = ctx.get_best()
                rec["best_x"] = [float(v) for v in bx]
        except (RuntimeError, IndexError, ValueError) as e:  # (the binding maps kc_status to these)
            rec = dict(ok=False, error=str(e))
        out.append(rec)
    (Path(out_dir) / f"rank{rank}.json").write_text(json.dumps(out))
    comm.close()
    ctx.close()


if __name__ == "__main__":
    main()
