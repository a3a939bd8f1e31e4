"""Multi-GPU sharding of the trajectory batch.

The samples of one controller cycle are independent, so rank g rolls out and
scores the contiguous block [first, first + count) of the host-generated sample
list; inputs (pose, trig table, tracked segment, obstacles) are replicated.
The only exchange step is ONE 8-byte all-reduce(min) of the packed
(cost, global sample index) key -- lexicographic int64 order reproduces
LowestCost::combine (datatypes/trajectory.h:630-636) because the global sample
index order is the reference's generation order.  The reference-numbered
(admissible-only) index is rebuilt on demand with one all-reduce(sum).

Round 3: the product path (kc_dwa_cycle_sharded) makes ONE all-reduce(int64 x (2 +
world x words), min) of an exchange record -- [best key, error word, every rank's
admissible bitmap] -- so that the reference-numbered index and the global
admissible count need no second collective; shares are dealt by rule
(kc_dwa_set_shard_rule: contiguous blocks, or by trig row).  `exchange_record`
below builds that record from a shard's result the way xchg_pack_kernel does;
the merge is the library's own host function (kompass_hip.shard_merge).

Works on any torch.distributed backend: "nccl" (= RCCL over xGMI) on GPUs,
"gloo" on CPU for the tests.
"""
from __future__ import annotations

import numpy as np

KEY_NONE = (1 << 63) - 1


def shard_range(n_total: int, rank: int, world: int):
    """Contiguous, balanced block of rank `rank`: (first, count)."""
    first = n_total * rank // world
    last = n_total * (rank + 1) // world
    return first, last - first


INT64_MAX = (1 << 63) - 1


def words_per_rank(counts) -> int:
    """64-bit bitmap words every rank's region of the exchange record holds."""
    return max((max(counts) + 63) // 64, 1) if len(counts) else 1


def exchange_record(rank: int, world: int, rw: int, key: int, admissible_local_ids, error: int = 0):
    """This rank's contribution to the exchange record (csrc/kc_shard.h): word 0 the packed key with
    the GLOBAL raw index, word 1 0 / -1 (error), then world regions of rw bitmap words -- this rank's
    admissible samples by shard-local id in its own region, INT64_MAX everywhere else."""
    x = np.full(2 + world * rw, INT64_MAX, np.int64)
    x[0] = key if not error else KEY_NONE
    x[1] = -1 if error else 0
    bits = np.zeros(rw * 64, np.uint8)
    ids = np.asarray(admissible_local_ids, np.int64)
    if not error and len(ids):
        bits[ids] = 1
    x[2 + rank * rw: 2 + (rank + 1) * rw] = np.packbits(bits, bitorder="little").view(np.int64)
    return x


def float_sortable(cost) -> int:
    """float32 -> int32 whose signed order equals the float order."""
    f = np.float32(cost) + np.float32(0.0)
    b = int(np.array([f], np.float32).view(np.int32)[0])
    return b if b >= 0 else b ^ 0x7FFFFFFF


def key_pack(cost, index: int) -> int:
    """Same packing as kc_key_pack (include/kompass_hip.h)."""
    if not (np.float32(cost) < np.finfo(np.float32).max):
        return KEY_NONE
    hi = float_sortable(cost) & 0xFFFFFFFF
    k = (hi << 32) | (int(index) & 0xFFFFFFFF)
    return k - (1 << 64) if k >= (1 << 63) else k


def key_unpack(key: int):
    """-> (found, cost, global sample index)."""
    if key == KEY_NONE:
        return False, 0.0, -1
    u = key & 0xFFFFFFFFFFFFFFFF
    s = (u >> 32) & 0xFFFFFFFF
    if s >= 1 << 31:
        s -= 1 << 32
    b = s if s >= 0 else s ^ 0x7FFFFFFF
    cost = float(np.array([b], np.int32).view(np.float32)[0])
    return True, cost, int(u & 0xFFFFFFFF)


def allreduce_best(key_tensor, group=None):
    """In-place all-reduce(min) of the int64 key tensor (1 element, on the
    device the backend wants).  The single collective of a cycle."""
    import torch.distributed as dist

    # (also with a single rank: the collective is then a copy, but the stream
    # ordering and the RCCL set-up are exercised exactly as with several)
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(key_tensor, op=dist.ReduceOp.MIN, group=group)
    return key_tensor


def global_compact_index(local_count_before: int, device=None, group=None) -> int:
    """Sum over ranks of `admissible samples in front of the winner on my
    shard` = the winner's index in the reference's admissible-only list."""
    import torch
    import torch.distributed as dist

    t = torch.tensor([int(local_count_before)], dtype=torch.int64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t.item())


class DevicePtrTensor:
    """Expose a raw device allocation owned by libkompass_hip.so to torch
    (zero copy) through __cuda_array_interface__."""

    def __init__(self, ptr: int, n_int64: int):
        self.__cuda_array_interface__ = {
            "shape": (n_int64,), "typestr": "<i8", "data": (int(ptr), False), "version": 2,
        }


def as_torch_int64(ptr: int, n: int, device):
    import torch

    return torch.as_tensor(DevicePtrTensor(ptr, n), device=device)
