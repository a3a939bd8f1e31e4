"""ctypes binding of libkompass_hip.so (C ABI: include/kompass_hip.h).

Thin, typed access to the HIP hot path for Python callers (tests, bench.py and
the kompass_core-style wrappers).  There is no CPU fallback: loading fails
loudly when the library has not been built, and every compute call raises
`KompassHipError` when no HIP device is usable.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
# KOMPASS_HIP_LIB: another build of the same library (same-box A/B of two builds, tools/ab_libs.sh)
LIB_PATH = Path(os.environ["KOMPASS_HIP_LIB"]) if os.environ.get("KOMPASS_HIP_LIB") else _HERE / "lib" / "libkompass_hip.so"

ACKERMANN, DIFFERENTIAL_DRIVE, OMNI = 0, 1, 2
CYLINDER, BOX, SPHERE = 0, 1, 2
UNEXPLORED, EMPTY, OCCUPIED = -1, 0, 100

KC_OK = 0
_ERR_TO_EXC = {-1: ValueError, -2: IndexError, -3: RuntimeError, -4: NotImplementedError, -5: RuntimeError}


class KompassHipError(RuntimeError):
    pass


class State(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("yaw", C.c_double), ("speed", C.c_double)]


class Limits(C.Structure):
    _fields_ = [
        ("vx_max", C.c_double), ("vx_acc", C.c_double), ("vx_dec", C.c_double),
        ("vy_max", C.c_double), ("vy_acc", C.c_double), ("vy_dec", C.c_double),
        ("omega_max_angle", C.c_double), ("omega_max", C.c_double),
        ("omega_acc", C.c_double), ("omega_dec", C.c_double),
    ]


class Weights(C.Structure):
    _fields_ = [
        ("reference_path_distance_weight", C.c_double),
        ("goal_distance_weight", C.c_double),
        ("obstacles_distance_weight", C.c_double),
        ("smoothness_weight", C.c_double),
        ("jerk_weight", C.c_double),
    ]


class DwaParams(C.Structure):
    _fields_ = [
        ("shape", C.c_int), ("dims", C.c_float * 3), ("ndims", C.c_int),
        ("sensor_pos", C.c_float * 3), ("sensor_rot_xyzw", C.c_float * 4),
        ("octree_res", C.c_double), ("time_step", C.c_double),
        ("max_samples", C.c_size_t), ("max_points", C.c_size_t),
        ("max_segment", C.c_size_t), ("max_obstacles", C.c_size_t),
        ("acc_limits", C.c_float * 3), ("device", C.c_int),
    ]


class StepInputs(C.Structure):
    """kc_step_inputs: one reference controller cycle (kc_dwa_find_best_path)."""
    _fields_ = [
        ("ctr_type", C.c_int), ("limits", C.c_void_p), ("cur_vx", C.c_double), ("cur_vy", C.c_double),
        ("cur_omega", C.c_double), ("max_linear_samples", C.c_int), ("max_angular_samples", C.c_int),
        ("points_xyz", C.c_void_p), ("n_points", C.c_size_t), ("scan_ranges", C.c_void_p), ("scan_angles", C.c_void_p),
        ("n_beams", C.c_size_t), ("max_sensor_range", C.c_float),
        ("seg_xyz", C.c_void_p), ("seg_x", C.c_void_p), ("seg_y", C.c_void_p), ("seg_z", C.c_void_p),
        ("acc_at_seg", C.c_void_p), ("seg_size", C.c_size_t), ("ref_path_length", C.c_float),
        ("num_points", C.c_size_t),
    ]


class Result(C.Structure):
    _fields_ = [
        ("found", C.c_int), ("cost", C.c_float), ("index", C.c_int64),
        ("raw_index", C.c_int64), ("n_admissible", C.c_int64), ("n_samples", C.c_int64),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


_fp = C.POINTER(C.c_float)
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)
_vp = C.c_void_p
_sz = C.c_size_t

# every symbol include/kompass_hip.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "kc_last_error": (C.c_char_p, []),
    "kc_abi_version": (C.c_int, []),
    "kc_device_count": (C.c_int, []),
    "kc_dwa_create": (C.c_int, [C.POINTER(DwaParams), C.POINTER(_vp)]),
    "kc_dwa_destroy": (None, [_vp]),
    "kc_dwa_set_stream": (C.c_int, [_vp, _vp]),
    "kc_dwa_set_resolution": (C.c_int, [_vp, C.c_double]),
    "kc_dwa_set_weights": (C.c_int, [_vp, C.POINTER(Weights)]),
    "kc_dwa_set_option": (C.c_int, [_vp, C.c_char_p, C.c_double]),
    "kc_dwa_get_option": (C.c_int, [_vp, C.c_char_p, C.POINTER(C.c_double)]),
    "kc_set_host_threads": (C.c_int, [C.c_int]),
    "kc_trig_selfcheck": (C.c_int, [C.POINTER(C.c_int64)]),
    "kc_trig_table": (C.c_int, [C.c_double, _dp, _sz, _sz, C.c_double, _dp]),
    "kc_dwa_sample_window": (C.c_int, [_vp, C.c_int, C.POINTER(Limits), C.c_double, C.c_double, C.c_double,
                                       C.c_int, C.c_int, C.POINTER(_sz), _dp, _dp, _dp, _sz]),
    "kc_dwa_set_samples": (C.c_int, [_vp, _sz, _dp, _dp, _dp]),
    "kc_dwa_set_shard": (C.c_int, [_vp, _sz, _sz]),
    "kc_dwa_set_shard_rule": (C.c_int, [_vp, C.c_int, C.c_int, C.c_int]),
    "kc_shard_plan": (C.c_int, [_ip, _sz, C.c_int, C.c_int, _ip]),
    "kc_shard_merge": (C.c_int, [C.POINTER(C.c_int64), _sz, C.c_int, C.c_int, _ip, _sz, C.POINTER(Result)]),
    "kc_dwa_owns_sample": (C.c_int, [_vp, C.c_int64, C.POINTER(C.c_int)]),
    "kc_dwa_set_scan": (C.c_int, [_vp, C.POINTER(State), _dp, _dp, _sz, C.c_float]),
    "kc_dwa_set_points": (C.c_int, [_vp, C.POINTER(State), _fp, _sz, C.c_float]),
    "kc_dwa_set_points_sensor_frame": (C.c_int, [_vp, C.POINTER(State), _fp, _sz, C.c_float]),
    "kc_dwa_set_path": (C.c_int, [_vp, _fp, _fp, _fp, _fp, _sz, C.c_float]),
    "kc_dwa_set_tracked_window": (C.c_int, [_vp, _sz, _sz]),
    "kc_dwa_set_grid_device": (C.c_int, [_vp, C.POINTER(State), _vp, C.c_int, C.c_int, C.c_float, C.c_int, C.c_int,
                                         C.c_float]),
    "kc_dwa_set_grid_from_mapper": (C.c_int, [_vp, C.POINTER(State), _vp, C.c_float]),
    "kc_dwa_set_tracked_segment": (C.c_int, [_vp, _fp, _fp, _fp, _fp, _sz, C.c_float]),
    "kc_dwa_set_tracked_segment_xyz": (C.c_int, [_vp, _fp, _fp, _sz, C.c_float]),
    "kc_dwa_rollout": (C.c_int, [_vp, C.POINTER(State), _sz]),
    "kc_dwa_check_poses": (C.c_int, [_vp, _dp, _dp, _dp, _sz, C.POINTER(C.c_uint8)]),
    "kc_dwa_evaluate": (C.c_int, [_vp]),
    "kc_dwa_fetch_result": (C.c_int, [_vp, C.POINTER(Result)]),
    "kc_dwa_cycle": (C.c_int, [_vp, C.POINTER(State), _sz, C.POINTER(Result)]),
    "kc_dwa_find_best_path": (C.c_int, [_vp, C.POINTER(State), C.POINTER(StepInputs), C.POINTER(Result)]),
    "kc_dwa_get_best": (C.c_int, [_vp, _fp, _fp, _fp, _fp, _fp]),
    "kc_dwa_get_sample_velocity": (C.c_int, [_vp, C.c_int64, _dp, _dp, _dp]),
    "kc_dwa_get_samples": (C.c_int, [_vp, _fp, _fp, _ip, _fp, _sz, C.POINTER(_sz)]),
    "kc_dwa_get_freeze_steps": (C.c_int, [_vp, _ip, _sz, C.POINTER(_sz)]),
    "kc_cost_evaluate": (C.c_int, [_vp, _fp, _fp, _fp, _fp, _fp, _sz, _sz, _fp, C.POINTER(Result)]),
    "kc_cost_upload": (C.c_int, [_vp, _fp, _fp, _fp, _fp, _fp, _sz, _sz]),
    "kc_cost_evaluate_resident": (C.c_int, [_vp, _fp, C.POINTER(Result)]),
    "kc_dwa_result_device": (C.c_int, [_vp, C.POINTER(_vp)]),
    "kc_dwa_publish_result": (C.c_int, [_vp]),
    "kc_dwa_count_admissible_before": (C.c_int, [_vp, C.c_int64, C.POINTER(C.c_int64)]),
    "kc_comm_unique_id": (C.c_int, [C.POINTER(C.c_uint8)]),
    "kc_comm_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_uint8), C.c_int, C.POINTER(_vp)]),
    "kc_comm_create_shm": (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.c_int, C.POINTER(_vp)]),
    "kc_comm_transport": (C.c_int, [_vp]),
    "kc_comm_query": (C.c_int, [_vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "kc_comm_destroy": (None, [_vp]),
    "kc_comm_rank": (C.c_int, [_vp]),
    "kc_comm_world": (C.c_int, [_vp]),
    "kc_dwa_allreduce_best": (C.c_int, [_vp, _vp]),
    "kc_dwa_cycle_sharded": (C.c_int, [_vp, _vp, C.POINTER(State), _sz, C.POINTER(Result)]),
    "kc_dwa_exchange_best": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_float, C.c_int64, C.POINTER(Result)]),
    "kc_dwa_global_index": (C.c_int, [_vp, _vp, C.c_int64, C.POINTER(C.c_int64)]),
    "kc_key_cost": (C.c_float, [C.c_int64]),
    "kc_key_index": (C.c_int64, [C.c_int64]),
    "kc_key_pack": (C.c_int64, [C.c_float, C.c_int64]),
    "kc_dwa_timing_enable": (C.c_int, [_vp, C.c_int]),
    "kc_dwa_timing_get": (C.c_int, [_vp, C.POINTER(C.c_char_p), _fp, _sz, C.POINTER(_sz)]),
    "kc_mapper_create": (C.c_int, [C.c_int, C.c_int, C.c_float, _fp, C.c_float, _sz, C.c_int, C.POINTER(_vp)]),
    "kc_mapper_destroy": (None, [_vp]),
    "kc_mapper_set_stream": (C.c_int, [_vp, _vp]),
    "kc_mapper_scan_to_grid": (C.c_int, [_vp, _dp, _dp, _sz, _ip]),
    "kc_mapper_scan_to_grid_device": (C.c_int, [_vp, _dp, _dp, _sz]),
    "kc_mapper_grid_device": (C.c_int, [_vp, C.POINTER(_vp)]),
    "kc_mapper_enable_bayes": (C.c_int, [_vp, C.c_void_p]),
    "kc_mapper_scan_to_grid_bayes": (C.c_int, [_vp, _dp, _dp, _sz, _ip, _fp]),
    "kc_mapper_scan_to_grid_bayes_device": (C.c_int, [_vp, _dp, _dp, _sz]),
    "kc_mapper_prob_device": (C.c_int, [_vp, C.POINTER(_vp), C.POINTER(_vp)]),
    "kc_mapper_warp_previous": (C.c_int, [_vp, _fp, C.c_double]),
    "kc_mapper_get_previous_prob": (C.c_int, [_vp, _fp]),
    "kc_mapper_set_previous_prob": (C.c_int, [_vp, _fp]),
    "kc_mapper_sync": (C.c_int, [_vp]),
    "kc_mapper_timing_enable": (C.c_int, [_vp, C.c_int]),
    "kc_mapper_timing_get": (C.c_int, [_vp, C.POINTER(C.c_char_p), _fp, _sz, C.POINTER(_sz)]),
    "kc_zone_create": (C.c_int, [C.c_int, _fp, C.c_int, _fp, _fp, C.c_float, C.c_float, C.c_float, _dp, _sz,
                                 C.c_float, C.c_float, C.c_float, C.c_int, C.POINTER(_vp)]),
    "kc_zone_destroy": (None, [_vp]),
    "kc_zone_check": (C.c_int, [_vp, _dp, _sz, C.c_int, C.POINTER(C.c_float)]),
    "kc_zone_check_cloud": (C.c_int, [_vp, C.c_void_p, _sz, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "kc_zone_indices": (C.c_int, [_vp, C.c_int, C.POINTER(C.c_int64), _sz, C.POINTER(_sz)]),
    "kc_cloud_create": (C.c_int, [_sz, _sz, C.c_int, C.POINTER(_vp)]),
    "kc_cloud_destroy": (None, [_vp]),
    "kc_cloud_to_laserscan": (C.c_int, [_vp, C.c_void_p, _sz, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                        C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                        C.c_double, C.c_int, _dp, _dp, _sz, C.POINTER(_sz)]),
    "kc_cloud_to_laserscan_typed": (C.c_int, [_vp, C.c_void_p, _sz, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                              C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_double,
                                              C.c_int, _dp, _dp, _sz, C.POINTER(_sz)]),
    "kc_zone_check_cloud_typed": (C.c_int, [_vp, C.c_void_p, _sz, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_int, C.c_int, C.c_int, _fp]),
    "kc_cloud_last_rebinned": (C.c_int, [_vp, C.POINTER(_sz)]),
    "kc_cloud_timing_enable": (C.c_int, [_vp, C.c_int]),
    "kc_cloud_timing_get": (C.c_int, [_vp, C.POINTER(C.c_char_p), _fp, _sz, C.POINTER(_sz)]),
}

_lib = None


def lib():
    """Load libkompass_hip.so (raises if it was not built -- no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise KompassHipError(
            f"{LIB_PATH} is missing: build it with `make -C {_HERE}` "
            "(or __graft_entry__.build()); there is no CPU fallback")
    L = C.CDLL(str(LIB_PATH))
    # KOMPASS_HIP_LIB_OLD=1 (with KOMPASS_HIP_LIB): an OLDER build of the library in a same-box A/B -- entries it does
    # not export yet are skipped instead of failing the load (tools only; the tests never set it)
    tolerant = os.environ.get("KOMPASS_HIP_LIB_OLD") == "1"
    for name, (res, args) in SIGNATURES.items():
        try:
            f = getattr(L, name)  # AttributeError if the symbol is not exported
        except AttributeError:
            if tolerant:
                continue
            raise
        f.restype = res
        f.argtypes = args
    _lib = L
    _bind_fast(L)
    return L


# The four calls of a controller cycle with fresh inputs (window, points, segment, cycle) sit on the critical
# path in front of the cycle kernel's launch: a second prototype of each takes plain ADDRESSES (c_void_p from an
# int: no ctypes pointer objects per call), the wrappers below reuse one State / Result per context and hand
# float32 C-contiguous arrays over where they lie.
_fast = {}


def _bind_fast(L):
    vp, i, d, f, z = C.c_void_p, C.c_int, C.c_double, C.c_float, C.c_size_t
    protos = {
        "kc_dwa_set_points": (vp, vp, vp, z, f),
        "kc_dwa_set_tracked_segment": (vp, vp, vp, vp, vp, z, f),
        "kc_dwa_set_tracked_segment_xyz": (vp, vp, vp, z, f),
        "kc_dwa_cycle": (vp, vp, z, vp),
        "kc_dwa_find_best_path": (vp, vp, vp, vp),
        "kc_dwa_sample_window": (vp, i, vp, d, d, d, i, i, vp, vp, vp, vp, z),
    }
    for name, args in protos.items():
        try:
            _fast[name] = C.CFUNCTYPE(C.c_int, *args)((name, L))
        except AttributeError:
            if os.environ.get("KOMPASS_HIP_LIB_OLD") != "1":
                raise


def _addr32(a):
    """(float32 C-contiguous array, its address): `a` itself when it already is one."""
    if not (type(a) is np.ndarray and a.dtype == np.float32 and a.flags.c_contiguous):
        a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.__array_interface__["data"][0]


def _check(rc):
    if rc != KC_OK:
        msg = lib().kc_last_error().decode("utf-8", "replace")
        exc = _ERR_TO_EXC.get(rc, KompassHipError)
        if exc is RuntimeError:
            exc = KompassHipError
        raise exc(f"[kc {rc}] {msg}")


def device_count() -> int:
    return lib().kc_device_count()


def set_host_threads(n: int):
    """Threads of the process-wide host pool behind the roll-out's libm trig table."""
    _check(lib().kc_set_host_threads(int(n)))


def trig_selfcheck() -> int:
    """The restated sincos (csrc/kc_trig_exact.h) against the installed libm on the library's fixed argument
    set (host only); returns the number of arguments compared, raises when any differs."""
    n = C.c_int64(0)
    _check(lib().kc_trig_selfcheck(C.byref(n)))
    return int(n.value)


def trig_table(yaw0: float, omega, n_steps: int, dt: float):
    """{cos, sin}(yaw_k) of every omega row, formed on the device: [n_steps, n_rows, 2] float64."""
    om = np.ascontiguousarray(omega, dtype=np.float64)
    out = np.empty((int(n_steps), len(om), 2), dtype=np.float64)
    _check(lib().kc_trig_table(float(yaw0), om.ctypes.data_as(_dp), len(om), int(n_steps), float(dt),
                               out.ctypes.data_as(_dp)))
    return out


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _pf(a):
    return a.ctypes.data_as(_fp) if a is not None else None


def _pd(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def make_limits(vx=(1.0, 10.0, 10.0), vy=(1.0, 10.0, 10.0), omega=(np.pi, 1.0, 10.0, 10.0)) -> Limits:
    return Limits(vx[0], vx[1], vx[2], vy[0], vy[1], vy[2], omega[0], omega[1], omega[2], omega[3])


def make_weights(path=1.0, goal=1.0, obstacles=1.0, smoothness=1.0, jerk=1.0) -> Weights:
    return Weights(path, goal, obstacles, smoothness, jerk)


COMM_ID_BYTES = 128


def comm_unique_id() -> bytes:
    """ncclGetUniqueId through the C ABI: call on one rank, send the bytes to every rank."""
    buf = (C.c_uint8 * COMM_ID_BYTES)()
    _check(lib().kc_comm_unique_id(buf))
    return bytes(buf)


SHARD_BLOCKS, SHARD_ROWS = 0, 1
COMM_RCCL, COMM_SHM = 0, 1


def shard_plan(rows, world: int, mode: int = SHARD_ROWS):
    """kc_shard_plan: owner rank of every sample from its trig row (pure host function)."""
    rows = np.ascontiguousarray(rows, np.int32)
    owner = np.zeros(len(rows), np.int32)
    _check(lib().kc_shard_plan(rows.ctypes.data_as(_ip), len(rows), int(world), int(mode),
                               owner.ctypes.data_as(_ip)))
    return owner


def shard_merge(record, words_per_rank: int, world: int, mode: int, owner, n_total: int) -> Result:
    """kc_shard_merge: the reduced exchange record -> result (pure host function)."""
    rec = np.ascontiguousarray(record, np.int64)
    own = None if owner is None else np.ascontiguousarray(owner, np.int32)
    r = Result()
    _check(lib().kc_shard_merge(rec.ctypes.data_as(C.POINTER(C.c_int64)), int(words_per_rank), int(world), int(mode),
                                None if own is None else own.ctypes.data_as(_ip), int(n_total), C.byref(r)))
    return r


class Comm:
    """Owner of one kc_comm: an RCCL communicator inside libkompass_hip.so, or -- shm_name given --
    the shared-memory rehearsal transport for ranks that share a GPU (kc_comm_create_shm)."""

    def __init__(self, rank: int, world: int, unique_id: bytes = None, device: int = 0, shm_name: str = None):
        self.h = _vp()
        if shm_name is not None:
            _check(lib().kc_comm_create_shm(int(rank), int(world), shm_name.encode(), int(device), C.byref(self.h)))
        else:
            assert len(unique_id) == COMM_ID_BYTES
            buf = (C.c_uint8 * COMM_ID_BYTES).from_buffer_copy(unique_id)
            _check(lib().kc_comm_create(int(rank), int(world), buf, int(device), C.byref(self.h)))
        self.rank, self.world = int(rank), int(world)

    @property
    def transport(self) -> str:
        return "shm" if lib().kc_comm_transport(self.h) == COMM_SHM else "rccl"

    def query(self):
        """(n_ranks, user_rank, device) as the transport itself reports them (RCCL: ncclCommCount /
        ncclCommUserRank / ncclCommCuDevice)."""
        n, r, d = C.c_int(0), C.c_int(0), C.c_int(0)
        _check(lib().kc_comm_query(self.h, C.byref(n), C.byref(r), C.byref(d)))
        return n.value, r.value, d.value

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            lib().kc_comm_destroy(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class DwaContext:
    """Owner of one kc_dwa context (one HIP stream, persistent device buffers)."""

    def __init__(self, shape, dims, sensor_pos=(0, 0, 0), sensor_rot_xyzw=(0, 0, 0, 1), octree_res=0.1,
                 time_step=0.1, max_samples=1024, max_points=64, max_segment=512, max_obstacles=1024,
                 acc_limits=(1.0, 1.0, 1.0), device=0):
        p = DwaParams()
        p.shape = int(shape)
        d = list(dims) + [0.0] * (3 - len(dims))
        for i in range(3):
            p.dims[i] = float(np.float32(d[i]))
            p.sensor_pos[i] = float(np.float32(sensor_pos[i]))
            p.acc_limits[i] = float(np.float32(acc_limits[i]))
        p.ndims = len(dims)
        for i in range(4):
            p.sensor_rot_xyzw[i] = float(np.float32(sensor_rot_xyzw[i]))
        p.octree_res = float(octree_res)
        p.time_step = float(time_step)
        p.max_samples, p.max_points = int(max_samples), int(max_points)
        p.max_segment, p.max_obstacles = int(max_segment), int(max_obstacles)
        p.device = int(device)
        self.params = p
        self.h = _vp()
        _check(lib().kc_dwa_create(C.byref(p), C.byref(self.h)))
        self._P = 0
        self._st = State()                      # reused by the per-cycle calls
        self._st_addr = C.addressof(self._st)
        self._n = _sz(0)
        self._n_addr = C.addressof(self._n)
        self._step = None

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            lib().kc_dwa_destroy(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- configuration ------------------------------------------------------
    def set_stream(self, stream_ptr):
        _check(lib().kc_dwa_set_stream(self.h, _vp(stream_ptr) if stream_ptr else None))

    def set_resolution(self, res):
        _check(lib().kc_dwa_set_resolution(self.h, float(res)))

    def set_option(self, name: str, value):
        _check(lib().kc_dwa_set_option(self.h, name.encode(), float(value)))

    def get_option(self, name: str) -> float:
        v = C.c_double(0.0)
        _check(lib().kc_dwa_get_option(self.h, name.encode(), C.byref(v)))
        return v.value

    def set_weights(self, w: Weights):
        _check(lib().kc_dwa_set_weights(self.h, C.byref(w)))

    def sample_window(self, ctr_type, limits: Limits, cur_vel, max_lin, max_ang, want_list=True):
        n = _sz(0)
        if want_list:
            cap = int(self.params.max_samples)
            vx, vy, om = np.zeros(cap), np.zeros(cap), np.zeros(cap)
            _check(lib().kc_dwa_sample_window(self.h, ctr_type, C.byref(limits), cur_vel[0], cur_vel[1],
                                              cur_vel[2], max_lin, max_ang, C.byref(n), _pd(vx), _pd(vy),
                                              _pd(om), cap))
            k = n.value
            return vx[:k].copy(), vy[:k].copy(), om[:k].copy()
        rc = _fast["kc_dwa_sample_window"](self.h, ctr_type, C.addressof(limits), cur_vel[0], cur_vel[1], cur_vel[2],
                                           max_lin, max_ang, self._n_addr, None, None, None, 0)
        if rc != KC_OK:
            _check(rc)
        return self._n.value

    def set_samples(self, vx, vy, omega):
        vx, vy, omega = _f64(vx), _f64(vy), _f64(omega)
        _check(lib().kc_dwa_set_samples(self.h, len(vx), _pd(vx), _pd(vy), _pd(omega)))

    def set_shard(self, first, count):
        _check(lib().kc_dwa_set_shard(self.h, int(first), int(count)))

    def set_shard_rule(self, rank, world, mode=SHARD_BLOCKS):
        """This context keeps rank `rank`'s share of every list it is given (mode < 0: all of it)."""
        _check(lib().kc_dwa_set_shard_rule(self.h, int(rank), int(world), int(mode)))

    def owns_sample(self, raw_index) -> bool:
        v = C.c_int(0)
        _check(lib().kc_dwa_owns_sample(self.h, int(raw_index), C.byref(v)))
        return bool(v.value)

    def set_scan(self, state, ranges, angles, max_sensor_range=10.0):
        r, a = _f64(ranges), _f64(angles)
        st = State(*state)
        _check(lib().kc_dwa_set_scan(self.h, C.byref(st), _pd(r), _pd(a), len(r), float(max_sensor_range)))

    def _state(self, state):
        st = self._st
        st.x, st.y, st.yaw, st.speed = state
        return self._st_addr

    def set_points(self, state, xyz, max_sensor_range=10.0, global_frame=True):
        """updateSensorData(cloud, global_frame): world-frame points, or (global_frame=False) sensor-frame points."""
        p, addr = _addr32(xyz)
        if global_frame:
            rc = _fast["kc_dwa_set_points"](self.h, self._state(state), addr, p.size // 3, max_sensor_range)
        else:
            st = State(*state)
            rc = lib().kc_dwa_set_points_sensor_frame(self.h, C.byref(st), _pf(p), p.size // 3, float(max_sensor_range))
        if rc != KC_OK:
            _check(rc)

    def set_grid_device(self, state, dev_grid_ptr, grid_height, grid_width, resolution, central=None,
                        max_sensor_range=10.0):
        """OCCUPIED cells of a device-resident LocalMapper grid -> sensor data (8f rank 4)."""
        st = State(*state)
        if central is None:  # local_mapper.h:26-27
            central = (int(round(grid_height // 2)) - 1, int(round(grid_width // 2)) - 1)
        _check(lib().kc_dwa_set_grid_device(self.h, C.byref(st), _vp(dev_grid_ptr), int(grid_height),
                                            int(grid_width), float(np.float32(resolution)), int(central[0]),
                                            int(central[1]), float(max_sensor_range)))

    def set_grid_from_mapper(self, state, mapper, max_sensor_range=10.0):
        """Same, from a MapperContext: the scan may still be in flight (stream-ordered)."""
        st = State(*state)
        _check(lib().kc_dwa_set_grid_from_mapper(self.h, C.byref(st), mapper.h, float(max_sensor_range)))

    def set_path(self, path_xyz, acc_at_point, total_length):
        """The whole interpolated reference path, resident on the device (once per path)."""
        pts = _f32(path_xyz).reshape(-1, 3)
        x, y, z = _f32(pts[:, 0]), _f32(pts[:, 1]), _f32(pts[:, 2])
        acc = _f32(acc_at_point)
        assert len(acc) == len(x)
        _check(lib().kc_dwa_set_path(self.h, _pf(x), _pf(y), _pf(z), _pf(acc), len(x),
                                     float(np.float32(total_length))))

    def set_tracked_window(self, start, size):
        """Tracked segment = points [start, start + size) of the resident path."""
        _check(lib().kc_dwa_set_tracked_window(self.h, int(start), int(size)))

    def set_tracked_segment(self, seg_xyz, acc_at_seg, ref_path_length):
        """Segment points as ONE (S, 3) array (kc_dwa_set_tracked_segment_xyz: no column copies here)."""
        seg, a_seg = _addr32(seg_xyz)
        acc, a_acc = _addr32(acc_at_seg)
        n = seg.size // 3
        assert acc.size == n
        rc = _fast["kc_dwa_set_tracked_segment_xyz"](self.h, a_seg, a_acc, n, float(np.float32(ref_path_length)))
        if rc != KC_OK:
            _check(rc)

    def set_tracked_segment_columns(self, x, y, z, acc_at_seg, ref_path_length):
        """The same from separate x / y / z arrays (what the C ABI takes: contiguous float32 arrays are passed
        where they lie, without the column copies of set_tracked_segment)."""
        (x, ax), (y, ay), (z, az), (acc, aa) = _addr32(x), _addr32(y), _addr32(z), _addr32(acc_at_seg)
        assert len(acc) == len(x)
        rc = _fast["kc_dwa_set_tracked_segment"](self.h, ax, ay, az, aa, len(x), float(np.float32(ref_path_length)))
        if rc != KC_OK:
            _check(rc)

    # -- cycle --------------------------------------------------------------
    def rollout(self, state, P):
        st = State(*state)
        self._P = int(P)
        _check(lib().kc_dwa_rollout(self.h, C.byref(st), int(P)))

    def check_poses(self, x, y, yaw):
        x, y, yaw = _f64(x), _f64(y), _f64(yaw)
        hit = np.zeros(len(x), np.uint8)
        _check(lib().kc_dwa_check_poses(self.h, _pd(x), _pd(y), _pd(yaw), len(x),
                                        hit.ctypes.data_as(C.POINTER(C.c_uint8))))
        return hit.astype(bool)

    def evaluate(self):
        _check(lib().kc_dwa_evaluate(self.h))

    def fetch_result(self) -> Result:
        r = Result()
        _check(lib().kc_dwa_fetch_result(self.h, C.byref(r)))
        return r

    def cycle(self, state, P) -> Result:
        self._P = int(P)
        r = Result()
        rc = _fast["kc_dwa_cycle"](self.h, self._state(state), self._P, C.addressof(r))
        if rc != KC_OK:
            _check(rc)
        return r

    def find_best_path(self, state, P, *, window=None, points=None, scan=None, max_sensor_range=10.0,
                       segment=None) -> Result:
        """One reference controller cycle in ONE call (kc_dwa_find_best_path = DWA::findBestPath, dwa.h:183-230):
        window = (ctr_type, Limits, (vx, vy, omega), max_linear_samples, max_angular_samples), points = (n, 3)
        float32 array or scan = (ranges, angles) float64 arrays, segment = (seg_xyz (S, 3) float32, acc_at_seg,
        ref_path_length).  An omitted part keeps what the context holds.  The arrays of the previous call are
        recognised by identity (same objects: their addresses are reused)."""
        si = self._step
        if si is None:
            si = self._step = StepInputs()
            self._step_addr = C.addressof(si)
            self._step_keep = [None] * 6   # the arrays whose addresses sit in the structure (kept alive)
        keep = self._step_keep
        if window is not None:
            ctr, lim, cur, ml, ma = window
            if keep[5] is not lim:
                keep[5] = lim
                si.limits = C.addressof(lim)
            si.ctr_type, si.max_linear_samples, si.max_angular_samples = ctr, ml, ma
            si.cur_vx, si.cur_vy, si.cur_omega = cur
        else:
            si.limits = None
            keep[5] = None
        if points is not None:
            if keep[0] is not points:
                p, addr = _addr32(points)
                keep[0] = points if p is points else None  # (a converted copy lives only for this call)
                self._step_tmp = p
                si.points_xyz, si.n_points = addr, p.size // 3
                si.scan_ranges = si.scan_angles = None
        elif scan is not None:
            r, ang = _f64(scan[0]), _f64(scan[1])
            self._step_tmp = (r, ang)
            keep[0] = None
            si.points_xyz = None
            si.scan_ranges, si.scan_angles, si.n_beams = r.ctypes.data, ang.ctypes.data, len(r)
        else:
            si.points_xyz = si.scan_ranges = si.scan_angles = None
            keep[0] = None
        si.max_sensor_range = max_sensor_range
        if segment is not None:
            seg, acc, ref_len = segment
            if keep[1] is not seg or keep[2] is not acc:
                s2, a_seg = _addr32(seg)
                a2, a_acc = _addr32(acc)
                keep[1] = seg if s2 is seg else None
                keep[2] = acc if a2 is acc else None
                self._step_tmp2 = (s2, a2)
                si.seg_xyz, si.acc_at_seg, si.seg_size = a_seg, a_acc, s2.size // 3
                si.seg_x = si.seg_y = si.seg_z = None
            si.ref_path_length = ref_len
        else:
            si.seg_size = 0
            keep[1] = keep[2] = None
        self._P = si.num_points = int(P)
        r = Result()
        rc = _fast["kc_dwa_find_best_path"](self.h, self._state(state), self._step_addr, C.addressof(r))
        if rc != KC_OK:
            _check(rc)
        return r

    def get_best(self):
        P = self._P
        px, py = np.zeros(P, np.float32), np.zeros(P, np.float32)
        v = [np.zeros(P - 1, np.float32) for _ in range(3)]
        _check(lib().kc_dwa_get_best(self.h, _pf(px), _pf(py), _pf(v[0]), _pf(v[1]), _pf(v[2])))
        return px, py, v

    def get_samples(self, with_costs=False, with_paths=True):
        P = self._P
        n = _sz(0)
        _check(lib().kc_dwa_get_samples(self.h, None, None, None, None, 0, C.byref(n)))
        k = n.value
        if with_paths:
            px, py = np.zeros((max(k, 1), P), np.float32), np.zeros((max(k, 1), P), np.float32)
        else:
            px = py = None
        raw = np.zeros(max(k, 1), np.int32)
        costs = np.zeros(max(k, 1), np.float32) if with_costs else None
        _check(lib().kc_dwa_get_samples(self.h, _pf(px), _pf(py), raw.ctypes.data_as(_ip), _pf(costs), k,
                                        C.byref(n)))
        out = (px[:k], py[:k], raw[:k]) if with_paths else (None, None, raw[:k])
        return out + (costs[:k],) if with_costs else out

    def get_freeze_steps(self):
        n = _sz(0)
        _check(lib().kc_dwa_get_freeze_steps(self.h, None, 0, C.byref(n)))
        st = np.zeros(max(n.value, 1), np.int32)
        _check(lib().kc_dwa_get_freeze_steps(self.h, st.ctypes.data_as(_ip), n.value, C.byref(n)))
        return st[:n.value]

    def get_sample_velocity(self, raw_index):
        vx, vy, om = C.c_double(0), C.c_double(0), C.c_double(0)
        _check(lib().kc_dwa_get_sample_velocity(self.h, int(raw_index), C.byref(vx), C.byref(vy), C.byref(om)))
        return vx.value, vy.value, om.value

    def cost_evaluate(self, paths_x, paths_y, vel=None):
        px, py = _f32(paths_x), _f32(paths_y)
        N, P = px.shape
        self._P = P
        costs = np.zeros(max(N, 1), np.float32)
        r = Result()
        v = [_f32(a) for a in vel] if vel is not None else [None, None, None]
        _check(lib().kc_cost_evaluate(self.h, _pf(px), _pf(py), _pf(v[0]), _pf(v[1]), _pf(v[2]), N, P,
                                      _pf(costs), C.byref(r)))
        return r, costs[:N]

    def cost_upload(self, paths_x, paths_y, vel=None):
        px, py = _f32(paths_x), _f32(paths_y)
        N, P = px.shape
        self._P, self._N = P, N
        v = [_f32(a) for a in vel] if vel is not None else [None, None, None]
        _check(lib().kc_cost_upload(self.h, _pf(px), _pf(py), _pf(v[0]), _pf(v[1]), _pf(v[2]), N, P))

    def cost_evaluate_resident(self, with_costs=True):
        r = Result()
        costs = np.zeros(max(self._N, 1), np.float32) if with_costs else None
        _check(lib().kc_cost_evaluate_resident(self.h, _pf(costs), C.byref(r)))
        return (r, costs[:self._N]) if with_costs else r

    def allreduce_best(self, comm: "Comm"):
        """ONE ncclAllReduce(int64, min) of the key record + hand-off of the reduced record."""
        _check(lib().kc_dwa_allreduce_best(self.h, comm.h))

    def cycle_sharded(self, comm: "Comm", state, P) -> Result:
        st = State(*state)
        self._P = int(P)
        r = Result()
        _check(lib().kc_dwa_cycle_sharded(self.h, comm.h, C.byref(st), int(P), C.byref(r)))
        return r

    def exchange_best(self, comm: "Comm", found, cost, raw_index, status=0) -> Result:
        """The exchange of a sharded cycle whose last cost terms the HOST added (custom cost callbacks): this
        rank's own best {found, cost, global raw index} in, the global result out (kc_dwa_exchange_best)."""
        r = Result()
        _check(lib().kc_dwa_exchange_best(self.h, comm.h, int(status), int(bool(found)), float(np.float32(cost)),
                                          int(raw_index), C.byref(r)))
        return r

    def global_index(self, comm: "Comm", raw_index) -> int:
        v = C.c_int64(0)
        _check(lib().kc_dwa_global_index(self.h, comm.h, int(raw_index), C.byref(v)))
        return v.value

    def publish_result(self):
        """After an in-place reduction of the device record: hand it to the host
        through pinned memory (fetch_result then returns the reduced key)."""
        _check(lib().kc_dwa_publish_result(self.h))

    def result_device_ptr(self) -> int:
        p = _vp()
        _check(lib().kc_dwa_result_device(self.h, C.byref(p)))
        return p.value

    def count_admissible_before(self, raw_index) -> int:
        c = C.c_int64(0)
        _check(lib().kc_dwa_count_admissible_before(self.h, int(raw_index), C.byref(c)))
        return c.value

    def timing_enable(self, on=True):
        _check(lib().kc_dwa_timing_enable(self.h, int(bool(on))))

    def timings(self):
        names = (C.c_char_p * 32)()
        ms = (C.c_float * 32)()
        n = _sz(0)
        _check(lib().kc_dwa_timing_get(self.h, names, ms, 32, C.byref(n)))
        return [(names[i].decode(), float(ms[i])) for i in range(n.value)]


class MapperContext:
    """Owner of one kc_mapper context (LocalMapper scan -> grid)."""

    def __init__(self, grid_height, grid_width, resolution, laserscan_position=(0, 0, 0),
                 laserscan_orientation=0.0, max_scan_size=4096, device=0):
        pos = _f32(laserscan_position)
        self.H, self.W = int(grid_height), int(grid_width)
        self.h = _vp()
        _check(lib().kc_mapper_create(self.H, self.W, float(np.float32(resolution)), _pf(pos),
                                      float(np.float32(laserscan_orientation)), int(max_scan_size),
                                      int(device), C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            lib().kc_mapper_destroy(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_ptr):
        _check(lib().kc_mapper_set_stream(self.h, _vp(stream_ptr) if stream_ptr else None))

    def scan_to_grid(self, angles, ranges):
        """-> int32 [H, W] with the Eigen column-major storage undone."""
        a, r = _f64(angles), _f64(ranges)
        g = np.empty(self.H * self.W, np.int32)
        _check(lib().kc_mapper_scan_to_grid(self.h, _pd(a), _pd(r), len(a), g.ctypes.data_as(_ip)))
        return g.reshape(self.W, self.H).T

    def scan_to_grid_device(self, angles, ranges):
        a, r = _f64(angles), _f64(ranges)
        _check(lib().kc_mapper_scan_to_grid_device(self.h, _pd(a), _pd(r), len(a)))

    # ---- M3: Bayesian update (LocalMapper's second ctor) ----------------------
    def enable_bayes(self, p_prior=0.5, p_occupied=0.6, p_empty=0.4, range_sure=1.0, range_max=20.0,
                     wall_size=0.2):
        params = (C.c_float * 6)(*[float(np.float32(v)) for v in
                                   (p_prior, p_occupied, p_empty, range_sure, range_max, wall_size)])
        _check(lib().kc_mapper_enable_bayes(self.h, C.cast(params, C.c_void_p)))

    def scan_to_grid_baysian(self, angles, ranges):
        """-> (int32 [H, W] occupancy, float32 [H, W] probabilities)."""
        a, r = _f64(angles), _f64(ranges)
        g = np.empty(self.H * self.W, np.int32)
        pr = np.empty(self.H * self.W, np.float32)
        _check(lib().kc_mapper_scan_to_grid_bayes(self.h, _pd(a), _pd(r), len(a), g.ctypes.data_as(_ip),
                                                  _pf(pr)))
        return g.reshape(self.W, self.H).T, pr.reshape(self.W, self.H).T

    def scan_to_grid_baysian_device(self, angles, ranges):
        a, r = _f64(angles), _f64(ranges)
        _check(lib().kc_mapper_scan_to_grid_bayes_device(self.h, _pd(a), _pd(r), len(a)))

    def prob_device_ptrs(self):
        p, q = _vp(), _vp()
        _check(lib().kc_mapper_prob_device(self.h, C.byref(p), C.byref(q)))
        return p.value, q.value

    def get_previous_grid_in_current_pose(self, position, orientation):
        pos = _f32(np.asarray(position, np.float32)[:2])
        _check(lib().kc_mapper_warp_previous(self.h, _pf(pos), float(orientation)))

    def previous_prob(self):
        pr = np.empty(self.H * self.W, np.float32)
        _check(lib().kc_mapper_get_previous_prob(self.h, _pf(pr)))
        return pr.reshape(self.W, self.H).T

    def set_previous_prob(self, prob=None):
        """prob [H, W] -> previous grid; None feeds the last scan's probabilities back (device copy)."""
        if prob is None:
            _check(lib().kc_mapper_set_previous_prob(self.h, None))
            return
        flat = np.ascontiguousarray(np.asarray(prob, np.float32).T).reshape(-1)
        if flat.size != self.H * self.W:
            raise ValueError("previous grid must be [grid_height, grid_width]")
        _check(lib().kc_mapper_set_previous_prob(self.h, _pf(flat)))

    def sync(self):
        _check(lib().kc_mapper_sync(self.h))

    def grid_device_ptr(self) -> int:
        p = _vp()
        _check(lib().kc_mapper_grid_device(self.h, C.byref(p)))
        return p.value

    def timing_enable(self, on=True):
        _check(lib().kc_mapper_timing_enable(self.h, int(bool(on))))

    def timings(self):
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        n = _sz(0)
        _check(lib().kc_mapper_timing_get(self.h, names, ms, 16, C.byref(n)))
        return [(names[i].decode(), float(ms[i])) for i in range(n.value)]


class CloudContext:
    """Owner of one kc_cloud context (raw point cloud -> laserscan)."""

    def __init__(self, max_bytes=1 << 20, max_bins=4096, device=0):
        self.h = _vp()
        _check(lib().kc_cloud_create(int(max_bytes), int(max_bins), int(device), C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            lib().kc_cloud_destroy(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def to_laserscan(self, data, point_step, row_step, height, width, x_offset, y_offset, z_offset,
                     max_range, min_z, max_z, angle_step=None, num_bins=None, device_ptr=None, nbytes=None,
                     field_type=7):
        """pointCloudToLaserScanFromRaw: (ranges, angles) with angle_step,
        ranges with num_bins.  `data`: bytes / int8 array on the host, or pass
        device_ptr + nbytes for a buffer that already lives on the device."""
        by_step = angle_step is not None
        nb = int(np.ceil(2.0 * np.pi / angle_step)) if by_step else int(num_bins)
        cap = max(nb, 1)
        ranges = np.zeros(cap, np.float64)
        angles = np.zeros(cap, np.float64)
        n = _sz(0)
        if device_ptr is not None:
            ptr, size, on_dev = int(device_ptr), int(nbytes), 1
        else:
            buf = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.int8)
                                       if not isinstance(data, np.ndarray) else data.view(np.int8).reshape(-1))
            ptr, size, on_dev = buf.ctypes.data, buf.size, 0
        _check(lib().kc_cloud_to_laserscan_typed(self.h, ptr, size, on_dev, int(point_step), int(row_step), int(height),
                                                 int(width), int(x_offset), int(y_offset), int(z_offset), int(field_type),
                                                 float(max_range), float(min_z), float(max_z),
                                                 float(angle_step) if by_step else 0.0, nb, _pd(ranges), _pd(angles),
                                                 cap, C.byref(n)))
        return (ranges[:n.value], angles[:n.value]) if by_step else ranges[:n.value]

    def last_rebinned(self) -> int:
        n = _sz(0)
        _check(lib().kc_cloud_last_rebinned(self.h, C.byref(n)))
        return n.value

    def timing_enable(self, on=True):
        _check(lib().kc_cloud_timing_enable(self.h, int(bool(on))))

    def timings(self):
        names = (C.c_char_p * 16)()
        ms = (C.c_float * 16)()
        n = _sz(0)
        _check(lib().kc_cloud_timing_get(self.h, names, ms, 16, C.byref(n)))
        return [(names[i].decode(), float(ms[i])) for i in range(n.value)]


class ZoneContext:
    """Owner of one kc_zone context (CriticalZoneChecker)."""

    def __init__(self, shape, dims, sensor_pos, sensor_rot_xyzw, critical_angle, critical_distance,
                 slowdown_distance, angles, min_height, max_height, range_max, device=0):
        d, sp, sr = _f32(dims), _f32(sensor_pos), _f32(sensor_rot_xyzw)
        a = _f64(angles)
        self.n = len(a)
        self.h = _vp()
        _check(lib().kc_zone_create(int(shape), _pf(d), len(d), _pf(sp), _pf(sr), float(np.float32(critical_angle)),
                                    float(np.float32(critical_distance)), float(np.float32(slowdown_distance)),
                                    _pd(a), len(a), float(np.float32(min_height)), float(np.float32(max_height)),
                                    float(np.float32(range_max)), int(device), C.byref(self.h)))

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            lib().kc_zone_destroy(self.h)
            self.h = _vp()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, ranges, forward) -> float:
        r = _f64(ranges)
        f = C.c_float(0)
        _check(lib().kc_zone_check(self.h, _pd(r), len(r), int(bool(forward)), C.byref(f)))
        return float(f.value)

    def check_cloud(self, data, point_step, row_step, height, width, x_offset, y_offset, z_offset, forward,
                    field_type=7) -> float:
        buf = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.int8) if not isinstance(data, np.ndarray)
                                   else data.view(np.int8).reshape(-1))
        f = C.c_float(0)
        _check(lib().kc_zone_check_cloud_typed(self.h, buf.ctypes.data, buf.size, int(point_step), int(row_step),
                                               int(height), int(width), int(x_offset), int(y_offset), int(z_offset),
                                               int(field_type), int(bool(forward)), C.byref(f)))
        return float(f.value)

    def indices(self, forward):
        out = (C.c_int64 * max(self.n, 1))()
        n = _sz(0)
        _check(lib().kc_zone_indices(self.h, int(bool(forward)), out, self.n, C.byref(n)))
        return np.array(out[:n.value], dtype=np.int64)
