"""Worker of tests/test_sharding_gloo.py: one rank of the sharded controller
cycle on CPU (gloo).  The per-shard compute is the oracle; what is under test is
the sharding / key / all-reduce logic of kompass-core_amd/sharding.py."""
import json
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path[:0] = [str(ROOT), str(ROOT / "kompass-core_amd"), str(ROOT / "tests")]

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import sharding  # noqa: E402
import synthetic as syn  # noqa: E402
from helpers import oracle_cycle  # noqa: E402


def main():
    out_path = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    results = []
    for name, scale, seed in [("cfg1", 1.0, 1), ("cfg2", 0.25, 2), ("cfg5", 0.08, 3)]:
        inp = syn.make_controller_inputs(name, seed=seed, scale=scale)
        n = len(inp["vx"])
        first, count = sharding.shard_range(n, rank, world)
        sl = slice(first, first + count)
        shard = dict(inp, vx=inp["vx"][sl], vy=inp["vy"][sl], omega=inp["omega"][sl])
        o = oracle_cycle(shard)
        # device record of one rank: key with the GLOBAL raw index
        if o["index"] >= 0:
            key = sharding.key_pack(o["cost"], first + int(o["raw"][o["index"]]))
        else:
            key = sharding.KEY_NONE
        kt = torch.tensor([key], dtype=torch.int64)
        sharding.allreduce_best(kt)
        found, cost, raw = sharding.key_unpack(int(kt.item()))
        local_before = int((o["raw"] + first < raw).sum()) if found else 0
        idx = sharding.global_compact_index(local_before) if found else -1
        results.append(dict(name=name, found=found, cost=cost, raw=raw, index=idx, rank=rank))
        # round 3: the product's single-collective protocol -- exchange record (key, error word, every
        # rank's admissible bitmap) through ONE all-reduce(min), merged by the LIBRARY's host function
        # (kc_shard_merge), shares dealt by the library's rule (kc_shard_plan)
        import kompass_hip as kh

        _, rows = np.unique(np.asarray(inp["omega"], np.float64) + 0.0, return_inverse=True)
        for mode in (kh.SHARD_BLOCKS, kh.SHARD_ROWS):
            owner = kh.shard_plan(rows, world, mode)
            mine = np.nonzero(owner == rank)[0]
            counts = [int((owner == q).sum()) for q in range(world)]
            rw = sharding.words_per_rank(counts)
            shard = dict(inp, vx=inp["vx"][mine], vy=inp["vy"][mine], omega=inp["omega"][mine])
            o = oracle_cycle(shard)
            key = sharding.key_pack(o["cost"], int(mine[o["raw"][o["index"]]])) if o["index"] >= 0 else sharding.KEY_NONE
            rec = torch.from_numpy(sharding.exchange_record(rank, world, rw, key, o["raw"]))
            dist.all_reduce(rec, op=dist.ReduceOp.MIN)  # the cycle's ONE collective
            r = kh.shard_merge(rec.numpy(), rw, world, mode, owner, n)
            results.append(dict(name=name, mode=int(mode), found=bool(r.found), cost=float(r.cost), raw=int(r.raw_index),
                                index=int(r.index), n_admissible=int(r.n_admissible), rank=rank))
        # a failing rank makes the cycle fail everywhere: rank 1 reports an error this time
        rec = torch.from_numpy(sharding.exchange_record(rank, world, rw, key, o["raw"], error=int(rank == world - 1)))
        dist.all_reduce(rec, op=dist.ReduceOp.MIN)
        try:
            kh.shard_merge(rec.numpy(), rw, world, kh.SHARD_ROWS, owner, n)
            results.append(dict(name=name, mode=-1, failed=False))
        except RuntimeError:
            results.append(dict(name=name, mode=-1, failed=True))
    if rank == 0:
        Path(out_path).write_text(json.dumps(results))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
