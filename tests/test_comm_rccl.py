"""The C-ABI collective (kc_comm_* / kc_dwa_cycle_sharded: RCCL inside
libkompass_hip.so) with a world of one rank on one GPU: the
ncclAllReduce runs for real (a copy onto itself), stream ordering and the
hand-off of the reduced record are the multi-GPU ones.  More ranks need more
GPUs: unmeasured on hardware until an 8-GPU node runs bench.py --gpus N; the
decomposition itself is covered by the eight sequential shards of
tests/test_full_size_parity.py and the world-size-2 gloo test."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import kompass_hip as kh  # noqa: E402
import synthetic as syn  # noqa: E402

from helpers import hip_context, oracle_cycle  # noqa: E402


def test_world_of_one_through_rccl():
    assert kh.device_count() >= 1
    uid = kh.comm_unique_id()
    assert len(uid) == kh.COMM_ID_BYTES and any(uid)
    comm = kh.Comm(0, 1, uid, device=0)
    assert kh.lib().kc_comm_world(comm.h) == 1 and kh.lib().kc_comm_rank(comm.h) == 0
    inp = syn.make_controller_inputs("cfg2", seed=1, scale=0.25)
    o = oracle_cycle(inp)
    for fused in (1, 0):
        ctx = hip_context(kh, inp)
        ctx.set_option("fused_cycle", fused)
        st = inp["state"]
        ctx.set_weights(kh.make_weights(*inp["weights"]))
        ctx.set_points(st, inp["points"], inp["max_range"])
        ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
        ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
        for rep in range(5):
            r = ctx.cycle_sharded(comm, st, inp["P"])
            assert r.found and r.raw_index == int(o["raw"][o["index"]])
            assert np.float32(r.cost) == np.float32(o["cost"]) and r.n_admissible == len(o["raw"])
            assert ctx.global_index(comm, r.raw_index) == o["index"]
            bx, by, _ = ctx.get_best()
            np.testing.assert_array_equal(bx, o["px"][o["index"]])
        # two halves of the list one after the other on this context, each through the collective:
        # the min of the two reduced keys is the unsharded winner
        n = len(inp["vx"])
        keys = []
        for first, count in ((0, n // 2), (n // 2, n - n // 2)):
            ctx.set_shard(first, count)
            r = ctx.cycle_sharded(comm, st, inp["P"])
            keys.append((np.float32(r.cost), r.raw_index) if r.found else (np.float32(np.inf), 1 << 40))
        assert min(keys) == (np.float32(o["cost"]), int(o["raw"][o["index"]]))
        ctx.close()
    with pytest.raises(IndexError):
        kh.Comm(3, 2, uid)
    comm.close()
