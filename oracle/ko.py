"""ctypes front-end of the CPU oracle (oracle/kompass_oracle.c).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product (kompass-core_amd/) never imports it.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from pathlib import Path as _P

import numpy as np

_HERE = _P(__file__).resolve().parent
_LIB = _HERE / "libkompass_oracle.so"

ACKERMANN, DIFFERENTIAL_DRIVE, OMNI = 0, 1, 2
CYLINDER, BOX, SPHERE = 0, 1, 2
UNEXPLORED, EMPTY, OCCUPIED = -1, 0, 100


def build(force: bool = False) -> _P:
    src = _HERE / "kompass_oracle.c"
    hdr = _HERE / "kompass_oracle.h"
    stale = (not _LIB.exists()) or _LIB.stat().st_mtime < max(
        src.stat().st_mtime, hdr.stat().st_mtime
    )
    if force or stale:
        subprocess.check_call(["make", "-s", "-C", str(_HERE), "-B"])
    return _LIB


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


_fp = C.POINTER(C.c_float)
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


class CostCtx(C.Structure):
    _fields_ = [
        ("seg_x", _fp), ("seg_y", _fp), ("seg_z", _fp),
        ("seg_size", C.c_size_t), ("seg_start_idx", C.c_size_t),
        ("path_acc", _fp), ("path_acc_size", C.c_size_t),
        ("ref_path_length", C.c_float),
        ("obs_x", _fp), ("obs_y", _fp), ("n_obs", C.c_size_t),
        ("max_obstacles_dist", C.c_float),
        ("acc_limits", C.c_float * 3),
        ("w", Weights),
    ]


class DwaConfig(C.Structure):
    _fields_ = [
        ("limits", Limits), ("ctr_type", C.c_int),
        ("time_step", C.c_double), ("prediction_horizon", C.c_double),
        ("control_horizon", C.c_double),
        ("max_linear_samples", C.c_int), ("max_angular_samples", C.c_int),
        ("shape", C.c_int), ("dims", C.c_float * 3), ("ndims", C.c_int),
        ("sensor_pos", C.c_float * 3), ("sensor_rot_xyzw", C.c_float * 4),
        ("octree_res", C.c_double), ("weights", Weights),
    ]


class DwaResult(C.Structure):
    _fields_ = [
        ("found", C.c_int), ("cost", C.c_float), ("index", C.c_long),
        ("raw_index", C.c_long), ("n_generated", C.c_long),
        ("n_admissible", C.c_long), ("P", C.c_size_t),
        ("seg_start", C.c_size_t), ("seg_size", C.c_size_t),
    ]


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    build()
    L = C.CDLL(str(_LIB))
    vp = C.c_void_p
    sz = C.c_size_t
    sig = {
        "ko_path_new": (vp, [_fp, _fp, _fp, sz]),
        "ko_path_free": (None, [vp]),
        "ko_path_interpolate_linear": (C.c_int, [vp, C.c_double]),
        "ko_path_segment": (None, [vp, C.c_double, sz]),
        "ko_path_size": (sz, [vp]),
        "ko_path_x": (_fp, [vp]), "ko_path_y": (_fp, [vp]), "ko_path_z": (_fp, [vp]),
        "ko_path_curvature": (_fp, [vp]), "ko_path_acc": (_fp, [vp]),
        "ko_path_acc_size": (sz, [vp]),
        "ko_path_total_length": (C.c_float, [vp]),
        "ko_path_num_segments": (sz, [vp]),
        "ko_path_segment_start": (sz, [vp, sz]),
        "ko_path_segment_end": (sz, [vp, sz]),
        "ko_num_trajectories": (sz, [C.c_int, C.c_int, C.c_int]),
        "ko_num_points_per_trajectory": (sz, [C.c_double, C.c_double]),
        "ko_sample_velocities": (C.c_long, [C.c_int, C.POINTER(Limits), C.c_double, C.c_double,
                                            C.c_double, C.c_double, C.c_int, C.c_int, _dp, _dp, _dp, sz]),
        "ko_coll_new": (vp, [C.c_int, _fp, C.c_int, _fp, _fp, C.c_double]),
        "ko_coll_free": (None, [vp]),
        "ko_coll_set_resolution": (None, [vp, C.c_double]),
        "ko_coll_update_state": (None, [vp, C.c_double, C.c_double, C.c_double]),
        "ko_coll_update_scan": (C.c_int, [vp, _dp, _dp, sz]),
        "ko_coll_update_points": (C.c_int, [vp, _fp, sz, C.c_int]),
        "ko_coll_check": (C.c_int, [vp]),
        "ko_coll_check_at": (C.c_int, [vp, C.c_double, C.c_double, C.c_double]),
        "ko_coll_radius": (C.c_float, [vp]),
        "ko_coll_num_voxels": (sz, [vp]),
        "ko_rollout": (C.c_long, [vp, C.POINTER(State), C.c_double, sz, _dp, _dp, _dp, sz,
                                  _fp, _fp, _fp, _fp, _fp, _ip]),
        "ko_rollout_mode": (C.c_long, [vp, C.POINTER(State), C.c_double, sz, _dp, _dp, _dp, sz, C.c_int, sz,
                                       _fp, _fp, _fp, _fp, _fp, _ip]),
        "ko_full_cycle_mode": (C.c_long, [vp, C.POINTER(CostCtx), C.POINTER(State), C.c_double, sz, _dp, _dp, _dp, sz,
                                          C.c_int, C.c_int, sz, _fp, _fp, _fp, _fp, _fp, C.POINTER(C.c_uint8), _fp]),
        "ko_min_trajectory_cost": (C.c_long, [C.POINTER(CostCtx), _fp, _fp, _fp, _fp, _fp, sz, sz,
                                              sz, sz, _fp, _fp]),
        "ko_path_cost": (C.c_float, [C.POINTER(CostCtx), _fp, _fp, sz]),
        "ko_goal_cost": (C.c_float, [C.POINTER(CostCtx), _fp, _fp, sz]),
        "ko_obstacle_cost": (C.c_float, [C.POINTER(CostCtx), _fp, _fp, sz]),
        "ko_smoothness_cost": (C.c_float, [C.POINTER(CostCtx), _fp, _fp, _fp, sz]),
        "ko_jerk_cost": (C.c_float, [C.POINTER(CostCtx), _fp, _fp, _fp, sz]),
        "ko_segment_length": (C.c_float, [_fp, _fp, _fp, sz]),
        "ko_obstacles_from_scan": (None, [_fp, _fp, C.POINTER(State), _dp, _dp, sz, _fp, _fp]),
        "ko_obstacles_from_points": (None, [_fp, _fp, C.POINTER(State), _fp, sz, _fp, _fp]),
        "ko_dwa_new": (vp, [C.POINTER(DwaConfig)]),
        "ko_dwa_free": (None, [vp]),
        "ko_dwa_set_path": (C.c_int, [vp, _fp, _fp, _fp, sz]),
        "ko_dwa_set_state": (None, [vp, C.c_double, C.c_double, C.c_double, C.c_double]),
        "ko_dwa_is_goal_reached": (C.c_int, [vp]),
        "ko_dwa_set_max_range": (None, [vp, C.c_float]),
        "ko_dwa_compute_scan": (C.c_int, [vp, C.c_double, C.c_double, C.c_double, _dp, _dp, sz,
                                          C.POINTER(DwaResult)]),
        "ko_dwa_compute_points": (C.c_int, [vp, C.c_double, C.c_double, C.c_double, _fp, sz,
                                            C.POINTER(DwaResult)]),
        "ko_dwa_best_path_x": (_fp, [vp]), "ko_dwa_best_path_y": (_fp, [vp]),
        "ko_dwa_best_vel": (_fp, [vp, C.c_int]),
        "ko_dwa_samples_x": (_fp, [vp]), "ko_dwa_samples_y": (_fp, [vp]),
        "ko_dwa_costs": (_fp, [vp]), "ko_dwa_raw_index": (_ip, [vp]),
        "ko_dwa_path": (vp, [vp]),
        "ko_dwa_max_segment_size": (sz, [vp]),
        "ko_dwa_closest_index": (sz, [vp]),
        "ko_czc_create": (C.c_void_p, [C.c_int, _fp, _fp, _fp, C.c_float, C.c_float, C.c_float, _dp, C.c_size_t,
                                       C.c_float, C.c_float, C.c_float]),
        "ko_czc_destroy": (None, [C.c_void_p]),
        "ko_czc_check": (C.c_float, [C.c_void_p, _dp, C.c_int]),
        "ko_czc_check_cloud": (C.c_float, [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int,
                                           C.c_int, C.c_int, C.c_int, C.c_int]),
        "ko_czc_indices": (C.c_size_t, [C.c_void_p, C.c_int, C.POINTER(C.c_size_t), C.c_size_t]),
        "ko_pointcloud_to_laserscan": (C.c_long, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int,
                                                 C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                                 C.c_double, C.c_int, _dp, _dp, C.c_size_t]),
        "ko_pointcloud_to_laserscan_typed": (C.c_long, [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_int,
                                                       C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double,
                                                       C.c_double, C.c_int, C.c_int, _dp, _dp, C.c_size_t]),
        "ko_czc_set_field_type": (None, [C.c_void_p, C.c_int]),
        "ko_mapper_scan_to_grid": (C.c_int, [C.c_int, C.c_int, C.c_float, _fp, C.c_float, _dp, _dp,
                                             sz, _ip]),
        "ko_bmap_create": (C.c_void_p, [C.c_int, C.c_int, C.c_float, _fp, C.c_float, C.c_float, C.c_float,
                                        C.c_float, C.c_float, C.c_float, C.c_float]),
        "ko_bmap_destroy": (None, [C.c_void_p]),
        "ko_bmap_scan": (C.c_int, [C.c_void_p, _dp, _dp, sz, _ip, _fp]),
        "ko_bmap_warp": (C.c_int, [C.c_void_p, _fp, C.c_double]),
        "ko_bmap_warp_matrix": (None, [C.c_void_p, _fp, C.c_double, _fp]),
        "ko_bmap_previous": (_fp, [C.c_void_p]),
        "ko_bmap_set_previous": (None, [C.c_void_p, _fp]),
        "ko_baseline_cycle": (C.c_long, [vp, C.POINTER(CostCtx), C.POINTER(State), C.c_double, sz,
                                         _dp, _dp, _dp, sz, C.c_int, _fp, C.POINTER(C.c_long)]),
        "ko_full_cycle": (C.c_long, [vp, C.POINTER(CostCtx), C.POINTER(State), C.c_double, sz,
                                     _dp, _dp, _dp, sz, C.c_int, _fp, _fp, C.POINTER(C.c_uint8), _fp]),
        "ko_costs_mt": (None, [C.POINTER(CostCtx), _fp, _fp, _fp, _fp, _fp, sz, sz, C.c_int, _fp]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


# --------------------------------------------------------------------------
# numpy helpers
# --------------------------------------------------------------------------
def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _pf(a):
    return a.ctypes.data_as(_fp)


def _pd(a):
    return a.ctypes.data_as(_dp)


def _pi(a):
    return a.ctypes.data_as(_ip)


def _arr(ptr, n, dtype=np.float32):
    if n == 0:
        return np.zeros(0, dtype=dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).astype(dtype, copy=True)


def make_limits(vx=(1.0, 10.0, 10.0), vy=(1.0, 10.0, 10.0), omega=(np.pi, 1.0, 10.0, 10.0)) -> Limits:
    """vx/vy = (max_vel, max_acc, max_dec); omega = (max_angle, max_omega, max_acc, max_dec)."""
    return Limits(vx[0], vx[1], vx[2], vy[0], vy[1], vy[2], omega[0], omega[1], omega[2], omega[3])


def make_weights(path=1.0, goal=1.0, obstacles=1.0, smoothness=1.0, jerk=1.0) -> Weights:
    return Weights(path, goal, obstacles, smoothness, jerk)


class Path:
    """Path::Path restatement handle."""

    def __init__(self, points=None, _handle=None, _own=True):
        L = lib()
        self._own = _own
        if _handle is not None:
            self.h = _handle
            return
        pts = _f32(points).reshape(-1, 3)
        x, y, z = _f32(pts[:, 0]), _f32(pts[:, 1]), _f32(pts[:, 2])
        self.h = L.ko_path_new(_pf(x), _pf(y), _pf(z), len(x))
        if not self.h:
            raise ValueError("At least two points are required to create a path.")

    def __del__(self):
        if getattr(self, "_own", False) and getattr(self, "h", None):
            lib().ko_path_free(self.h)
            self.h = None

    def interpolate(self, max_dist: float):
        lib().ko_path_interpolate_linear(self.h, float(max_dist))
        return self

    def segment(self, seg_len: float, max_pts: int):
        lib().ko_path_segment(self.h, float(seg_len), int(max_pts))
        return self

    @property
    def size(self):
        return lib().ko_path_size(self.h)

    @property
    def x(self):
        return _arr(lib().ko_path_x(self.h), self.size)

    @property
    def y(self):
        return _arr(lib().ko_path_y(self.h), self.size)

    @property
    def z(self):
        return _arr(lib().ko_path_z(self.h), self.size)

    @property
    def curvature(self):
        return _arr(lib().ko_path_curvature(self.h), self.size)

    @property
    def acc(self):
        return _arr(lib().ko_path_acc(self.h), lib().ko_path_acc_size(self.h))

    @property
    def total_length(self):
        return lib().ko_path_total_length(self.h)

    @property
    def num_segments(self):
        return lib().ko_path_num_segments(self.h)

    def segment_range(self, s):
        L = lib()
        return L.ko_path_segment_start(self.h, s), L.ko_path_segment_end(self.h, s)


def sample_velocities(ctr_type, limits: Limits, cur_vel, dt, max_lin, max_ang):
    L = lib()
    ang = max_ang + 1 - (max_ang % 2)
    cap = int(L.ko_num_trajectories(ctr_type, max_lin, ang)) + 16
    vx = np.zeros(cap)
    vy = np.zeros(cap)
    om = np.zeros(cap)
    n = L.ko_sample_velocities(ctr_type, C.byref(limits), cur_vel[0], cur_vel[1], cur_vel[2],
                               dt, max_lin, max_ang, _pd(vx), _pd(vy), _pd(om), cap)
    if n < 0:
        raise RuntimeError("sample buffer too small")
    return vx[:n].copy(), vy[:n].copy(), om[:n].copy()


class Collision:
    def __init__(self, shape, dims, sensor_pos=(0, 0, 0), sensor_rot_xyzw=(0, 0, 0, 1), res=0.1):
        d = _f32(dims)
        p = _f32(sensor_pos)
        r = _f32(sensor_rot_xyzw)
        self.h = lib().ko_coll_new(shape, _pf(d), len(d), _pf(p), _pf(r), float(res))
        if not self.h:
            raise ValueError("Invalid robot geometry type")

    def __del__(self):
        if getattr(self, "h", None):
            lib().ko_coll_free(self.h)
            self.h = None

    def update_state(self, x, y, yaw):
        lib().ko_coll_update_state(self.h, x, y, yaw)

    def update_scan(self, ranges, angles):
        r, a = _f64(ranges), _f64(angles)
        rc = lib().ko_coll_update_scan(self.h, _pd(r), _pd(a), len(r))
        if rc:
            raise RuntimeError(f"unsupported sensor frame ({rc})")

    def update_points(self, xyz, global_frame=True):
        p = _f32(xyz).reshape(-1, 3)
        rc = lib().ko_coll_update_points(self.h, _pf(p), len(p), int(global_frame))
        if rc:
            raise RuntimeError(f"unsupported sensor frame ({rc})")

    def check(self):
        return bool(lib().ko_coll_check(self.h))

    def check_at(self, x, y, yaw):
        return bool(lib().ko_coll_check_at(self.h, x, y, yaw))

    @property
    def num_voxels(self):
        return lib().ko_coll_num_voxels(self.h)


def rollout(coll, start, dt, P, vx, vy, omega, with_vel=False):
    """Returns (paths_x[Na,P], paths_y[Na,P], raw_index[Na], vel or None)."""
    L = lib()
    vx, vy, omega = _f64(vx), _f64(vy), _f64(omega)
    n = len(vx)
    px = np.zeros((max(n, 1), P), np.float32)
    py = np.zeros((max(n, 1), P), np.float32)
    raw = np.zeros(max(n, 1), np.int32)
    st = State(*start) if not isinstance(start, State) else start
    if with_vel:
        v = [np.zeros((max(n, 1), P - 1), np.float32) for _ in range(3)]
        vp = [_pf(a) for a in v]
    else:
        v = None
        vp = [None, None, None]
    na = L.ko_rollout(coll.h if coll is not None else None, C.byref(st), dt, P, _pd(vx), _pd(vy),
                      _pd(omega), n, _pf(px), _pf(py), vp[0], vp[1], vp[2], _pi(raw))
    out_v = [a[:na].copy() for a in v] if with_vel else None
    return px[:na].copy(), py[:na].copy(), raw[:na].copy(), out_v


class CostInputs:
    """Keeps the numpy buffers behind a ko_cost_ctx alive."""

    def __init__(self, seg_xyz, seg_start_idx, path_acc, ref_path_length, obstacles_xy=None,
                 max_obstacles_dist=10.0 / 3.0, acc_limits=(1, 1, 1), weights: Weights | None = None):
        seg = _f32(seg_xyz).reshape(-1, 3)
        self.sx, self.sy, self.sz = _f32(seg[:, 0]), _f32(seg[:, 1]), _f32(seg[:, 2])
        self.acc = _f32(path_acc)
        obs = _f32(obstacles_xy).reshape(-1, 2) if obstacles_xy is not None else np.zeros((0, 2), np.float32)
        self.ox, self.oy = _f32(obs[:, 0]), _f32(obs[:, 1])
        cx = CostCtx()
        cx.seg_x, cx.seg_y, cx.seg_z = _pf(self.sx), _pf(self.sy), _pf(self.sz)
        cx.seg_size = len(self.sx)
        cx.seg_start_idx = int(seg_start_idx)
        cx.path_acc = _pf(self.acc)
        cx.path_acc_size = len(self.acc)
        cx.ref_path_length = float(ref_path_length)
        cx.obs_x, cx.obs_y = _pf(self.ox), _pf(self.oy)
        cx.n_obs = len(self.ox)
        cx.max_obstacles_dist = float(np.float32(max_obstacles_dist))
        for i in range(3):
            cx.acc_limits[i] = float(np.float32(acc_limits[i]))
        cx.w = weights if weights is not None else make_weights()
        self.cx = cx


def min_trajectory_cost(ci: CostInputs, paths_x, paths_y, vel=None):
    """cost_evaluator.cpp:49-109 -> (argmin, min_cost, costs[N])."""
    px, py = _f32(paths_x), _f32(paths_y)
    N, P = px.shape
    costs = np.zeros(N, np.float32)
    mc = C.c_float(0)
    if vel is not None:
        v = [_f32(a) for a in vel]
        vp = [_pf(a) for a in v]
        sv = v[0].shape[1]
    else:
        vp = [None, None, None]
        sv = 0
    idx = lib().ko_min_trajectory_cost(C.byref(ci.cx), _pf(px), _pf(py), vp[0], vp[1], vp[2], N, P, P,
                                       sv, _pf(costs), C.byref(mc))
    return int(idx), float(mc.value), costs


def obstacles_from_scan(sensor_pos, sensor_rot_xyzw, state, ranges, angles):
    p, r = _f32(sensor_pos), _f32(sensor_rot_xyzw)
    rg, an = _f64(ranges), _f64(angles)
    ox = np.zeros(len(rg), np.float32)
    oy = np.zeros(len(rg), np.float32)
    st = State(*state)
    lib().ko_obstacles_from_scan(_pf(p), _pf(r), C.byref(st), _pd(rg), _pd(an), len(rg), _pf(ox), _pf(oy))
    return ox, oy


def obstacles_from_points(sensor_pos, sensor_rot_xyzw, state, xyz):
    p, r = _f32(sensor_pos), _f32(sensor_rot_xyzw)
    pts = _f32(xyz).reshape(-1, 3)
    ox = np.zeros(len(pts), np.float32)
    oy = np.zeros(len(pts), np.float32)
    st = State(*state)
    lib().ko_obstacles_from_points(_pf(p), _pf(r), C.byref(st), _pf(pts), len(pts), _pf(ox), _pf(oy))
    return ox, oy


class DWA:
    """controllers/dwa.{h,cpp} restatement (host glue + sampler + evaluator)."""

    def __init__(self, limits: Limits, ctr_type, time_step, prediction_horizon, control_horizon,
                 max_linear_samples, max_angular_samples, shape, dims, sensor_pos=(0, 0, 0),
                 sensor_rot_xyzw=(0, 0, 0, 1), octree_res=0.1, weights: Weights | None = None):
        cfg = DwaConfig()
        cfg.limits = limits
        cfg.ctr_type = ctr_type
        cfg.time_step = time_step
        cfg.prediction_horizon = prediction_horizon
        cfg.control_horizon = control_horizon
        cfg.max_linear_samples = max_linear_samples
        cfg.max_angular_samples = max_angular_samples
        cfg.shape = shape
        d = list(dims) + [0.0] * (3 - len(dims))
        for i in range(3):
            cfg.dims[i] = float(np.float32(d[i]))
            cfg.sensor_pos[i] = float(np.float32(sensor_pos[i]))
        cfg.ndims = len(dims)
        for i in range(4):
            cfg.sensor_rot_xyzw[i] = float(np.float32(sensor_rot_xyzw[i]))
        cfg.octree_res = octree_res
        cfg.weights = weights if weights is not None else make_weights()
        self.cfg = cfg
        self.h = lib().ko_dwa_new(C.byref(cfg))

    def __del__(self):
        if getattr(self, "h", None):
            lib().ko_dwa_free(self.h)
            self.h = None

    def set_path(self, points):
        pts = _f32(points).reshape(-1, 3)
        x, y, z = _f32(pts[:, 0]), _f32(pts[:, 1]), _f32(pts[:, 2])
        if lib().ko_dwa_set_path(self.h, _pf(x), _pf(y), _pf(z), len(x)):
            raise ValueError("At least two points are required to create a path.")

    def set_state(self, x, y, yaw, speed=0.0):
        lib().ko_dwa_set_state(self.h, x, y, yaw, speed)

    def is_goal_reached(self):
        return bool(lib().ko_dwa_is_goal_reached(self.h))

    def set_max_range(self, r):
        lib().ko_dwa_set_max_range(self.h, r)

    @property
    def path(self):
        return Path(_handle=lib().ko_dwa_path(self.h), _own=False)

    def compute(self, vel, scan=None, points=None):
        res = DwaResult()
        L = lib()
        if scan is not None:
            r, a = _f64(scan[0]), _f64(scan[1])
            rc = L.ko_dwa_compute_scan(self.h, vel[0], vel[1], vel[2], _pd(r), _pd(a), len(r), C.byref(res))
        else:
            p = _f32(points).reshape(-1, 3)
            rc = L.ko_dwa_compute_points(self.h, vel[0], vel[1], vel[2], _pf(p), len(p), C.byref(res))
        if rc == -1:
            raise ValueError("Pointer to global path is NULL. Cannot use DWA local planner without "
                             "setting a global path")
        if rc:
            raise RuntimeError(f"oracle DWA failed ({rc})")
        out = {k: getattr(res, k) for k, _ in DwaResult._fields_}
        P, na = res.P, res.n_admissible
        if res.found:
            out["path_x"] = _arr(L.ko_dwa_best_path_x(self.h), P)
            out["path_y"] = _arr(L.ko_dwa_best_path_y(self.h), P)
            out["vel"] = [_arr(L.ko_dwa_best_vel(self.h, c), P - 1) for c in range(3)]
        if na > 0:
            out["samples_x"] = _arr(L.ko_dwa_samples_x(self.h), na * P).reshape(na, P)
            out["samples_y"] = _arr(L.ko_dwa_samples_y(self.h), na * P).reshape(na, P)
            out["costs"] = _arr(L.ko_dwa_costs(self.h), na)
            out["raw"] = _arr(L.ko_dwa_raw_index(self.h), na, np.int32)
        return out


def scan_to_grid(H, W, res, position, orientation, angles, ranges):
    """LocalMapper::scanToGrid -> int32 [H, W] (Eigen column-major restored)."""
    p = _f32(position)
    a, r = _f64(angles), _f64(ranges)
    g = np.zeros(H * W, np.int32)
    lib().ko_mapper_scan_to_grid(H, W, float(np.float32(res)), _pf(p), float(np.float32(orientation)),
                                 _pd(a), _pd(r), len(a), _pi(g))
    return g.reshape(W, H).T.copy()  # column-major (i + j*H) -> [i, j]


class BayesMapper:
    """LocalMapper built with the Bayesian ctor (local_mapper.h:58-103):
    scanToGridBaysian + getPreviousGridInCurrentPose.  Grids come back as
    [H, W] arrays (Eigen's column-major storage restored)."""

    def __init__(self, H, W, res, position, orientation, p_prior=0.5, p_occupied=0.6, p_empty=0.4,
                 range_sure=1.0, range_max=20.0, wall_size=0.2):
        self.H, self.W = int(H), int(W)
        p = _f32(position)
        f = lambda v: float(np.float32(v))
        self.h = lib().ko_bmap_create(self.H, self.W, f(res), _pf(p), f(orientation), f(p_prior), f(p_occupied),
                                      f(p_empty), f(range_sure), f(range_max), f(wall_size))
        if not self.h:
            raise ValueError("invalid mapper arguments")

    def __del__(self):
        if getattr(self, "h", None):
            lib().ko_bmap_destroy(self.h)
            self.h = None

    def _unpack(self, flat):
        return flat.reshape(self.W, self.H).T.copy()

    def scan_to_grid_baysian(self, angles, ranges):
        a, r = _f64(angles), _f64(ranges)
        g = np.zeros(self.H * self.W, np.int32)
        pr = np.zeros(self.H * self.W, np.float32)
        lib().ko_bmap_scan(self.h, _pd(a), _pd(r), len(a), _pi(g), _pf(pr))
        return self._unpack(g), self._unpack(pr)

    def get_previous_grid_in_current_pose(self, position, orientation):
        p = _f32(np.asarray(position, np.float32)[:2])
        if lib().ko_bmap_warp(self.h, _pf(p), float(orientation)) != 0:
            raise MemoryError
        return self.previous()

    def warp_matrix(self, position, orientation):
        p = _f32(np.asarray(position, np.float32)[:2])
        inv = np.zeros(9, np.float32)
        lib().ko_bmap_warp_matrix(self.h, _pf(p), float(orientation), _pf(inv))
        return inv.reshape(3, 3)

    def previous(self):
        return self._unpack(_arr(lib().ko_bmap_previous(self.h), self.H * self.W))

    def set_previous(self, prob):
        flat = np.ascontiguousarray(np.asarray(prob, np.float32).T).reshape(-1)  # [i, j] -> i + j*H
        assert flat.size == self.H * self.W
        lib().ko_bmap_set_previous(self.h, _pf(flat))


FIELD_INT8, FIELD_UINT8, FIELD_INT16, FIELD_UINT16, FIELD_INT32, FIELD_UINT32, FIELD_FLOAT32, FIELD_FLOAT64 = range(1, 9)


def pointcloud_to_laserscan(data, point_step, row_step, height, width, x_offset, y_offset, z_offset,
                            max_range, min_z, max_z, angle_step=None, num_bins=None, field_type=FIELD_FLOAT32):
    """pointCloudToLaserScanFromRaw: (ranges, angles) for the angle_step
    overload, ranges for the num_bins overload."""
    buf = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.int8) if not isinstance(data, np.ndarray)
                               else data.view(np.int8).reshape(-1))
    by_step = angle_step is not None
    nb = int(np.ceil(2.0 * np.pi / angle_step)) if by_step else int(num_bins)
    cap = max(nb, 1)
    ranges = np.zeros(cap, np.float64)
    angles = np.zeros(cap, np.float64)
    n = lib().ko_pointcloud_to_laserscan_typed(buf.ctypes.data, buf.size, point_step, row_step, height, width,
                                               x_offset, y_offset, z_offset, float(max_range), float(min_z),
                                               float(max_z), float(angle_step) if by_step else 0.0, nb, int(field_type),
                                               _pd(ranges), _pd(angles), cap)
    if n < 0:
        raise ValueError("invalid point cloud arguments")
    return (ranges[:n], angles[:n]) if by_step else ranges[:n]


class CriticalZone:
    """CriticalZoneChecker (utils/critical_zone_check.cpp), CPU semantics."""

    def __init__(self, shape, dims, sensor_pos, sensor_rot_xyzw, critical_angle, critical_distance,
                 slowdown_distance, angles, min_height, max_height, range_max, field_type=FIELD_FLOAT32):
        d, sp, sr = _f32(dims), _f32(sensor_pos), _f32(sensor_rot_xyzw)
        self.angles = _f64(angles)
        self.h = lib().ko_czc_create(int(shape), _pf(d), _pf(sp), _pf(sr), float(np.float32(critical_angle)),
                                     float(np.float32(critical_distance)), float(np.float32(slowdown_distance)),
                                     _pd(self.angles), len(self.angles), float(np.float32(min_height)),
                                     float(np.float32(max_height)), float(np.float32(range_max)))
        if not self.h:
            raise ValueError("SlowDown distance must be greater than the Critical distance / invalid shape")
        lib().ko_czc_set_field_type(self.h, int(field_type))

    def __del__(self):
        if getattr(self, "h", None):
            lib().ko_czc_destroy(self.h)
            self.h = None

    def check(self, ranges, forward):
        r = _f64(ranges)
        assert len(r) >= len(self.angles)
        return float(lib().ko_czc_check(self.h, _pd(r), int(bool(forward))))

    def check_cloud(self, data, point_step, row_step, height, width, x_offset, y_offset, z_offset, forward):
        buf = np.ascontiguousarray(np.frombuffer(bytes(data), dtype=np.int8) if not isinstance(data, np.ndarray)
                                   else data.view(np.int8).reshape(-1))
        return float(lib().ko_czc_check_cloud(self.h, buf.ctypes.data, buf.size, point_step, row_step, height, width,
                                              x_offset, y_offset, z_offset, int(bool(forward))))

    def indices(self, forward):
        out = (C.c_size_t * max(len(self.angles), 1))()
        n = lib().ko_czc_indices(self.h, int(bool(forward)), out, len(self.angles))
        return np.array(out[:n], dtype=np.int64)


def baseline_cycle(coll, ci: CostInputs, start, dt, P, vx, vy, omega, threads=1):
    vx, vy, omega = _f64(vx), _f64(vy), _f64(omega)
    st = State(*start)
    mc = C.c_float(0)
    na = C.c_long(0)
    idx = lib().ko_baseline_cycle(coll.h if coll is not None else None, C.byref(ci.cx), C.byref(st), dt, P,
                                  _pd(vx), _pd(vy), _pd(omega), len(vx), threads, C.byref(mc), C.byref(na))
    return int(idx), float(mc.value), int(na.value)


def host_threads() -> int:
    """CPUs this process may really use (affinity mask capped by the cgroup quota)."""
    import os

    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def rollout_mode(coll, start, dt, P, vx, vy, omega, drop_samples=True, num_ctrl_points=0):
    """rollout() with both values of drop_samples_ (trajectory_sampler.cpp:157-168); always returns the
    velocity profiles: (paths_x[Na,P], paths_y[Na,P], raw_index[Na], [vx, vy, omega] each [Na,P-1])."""
    L = lib()
    vx, vy, omega = _f64(vx), _f64(vy), _f64(omega)
    n = len(vx)
    px = np.zeros((max(n, 1), P), np.float32)
    py = np.zeros((max(n, 1), P), np.float32)
    raw = np.zeros(max(n, 1), np.int32)
    st = State(*start) if not isinstance(start, State) else start
    v = [np.zeros((max(n, 1), P - 1), np.float32) for _ in range(3)]
    na = L.ko_rollout_mode(coll.h if coll is not None else None, C.byref(st), dt, P, _pd(vx), _pd(vy), _pd(omega), n,
                           int(bool(drop_samples)), int(num_ctrl_points), _pf(px), _pf(py), _pf(v[0]), _pf(v[1]),
                           _pf(v[2]), _pi(raw))
    return px[:na].copy(), py[:na].copy(), raw[:na].copy(), [a[:na].copy() for a in v]


def full_cycle_mode(coll, ci: CostInputs | None, start, dt, P, vx, vy, omega, drop_samples=True, num_ctrl_points=0,
                    threads=None):
    """full_cycle() with both values of drop_samples_; also returns the velocity profiles (`vel`)."""
    vx, vy, omega = _f64(vx), _f64(vy), _f64(omega)
    n = len(vx)
    px = np.zeros((max(n, 1), P), np.float32)
    py = np.zeros((max(n, 1), P), np.float32)
    v = [np.zeros((max(n, 1), P - 1), np.float32) for _ in range(3)]
    adm = np.zeros(max(n, 1), np.uint8)
    costs = np.zeros(max(n, 1), np.float32)
    st = State(*start)
    lib().ko_full_cycle_mode(coll.h if coll is not None else None, C.byref(ci.cx) if ci is not None else None,
                             C.byref(st), dt, P, _pd(vx), _pd(vy), _pd(omega), n, int(threads or host_threads()),
                             int(bool(drop_samples)), int(num_ctrl_points), _pf(px), _pf(py), _pf(v[0]), _pf(v[1]),
                             _pf(v[2]), adm.ctypes.data_as(C.POINTER(C.c_uint8)), _pf(costs))
    raw = np.flatnonzero(adm[:n]).astype(np.int32)
    c = costs[raw]
    ok = np.flatnonzero(c < np.finfo(np.float32).max)
    if len(ok):
        idx = int(ok[np.argmin(c[ok])])
        cost = float(c[idx])
    else:
        idx, cost = -1, 0.0
    return dict(px=px[raw], py=py[raw], raw=raw, costs=c, index=idx, cost=cost, vel=[a[raw] for a in v])


def full_cycle(coll, ci: CostInputs | None, start, dt, P, vx, vy, omega, threads=None):
    """Every sample rolled out and scored independently by `threads` workers
    (the per-sample arithmetic of rollout() + min_trajectory_cost()), compacted
    here in generation order.  -> dict(px, py, raw, costs, index, cost) like
    tests/helpers.oracle_cycle."""
    vx, vy, omega = _f64(vx), _f64(vy), _f64(omega)
    n = len(vx)
    px = np.zeros((max(n, 1), P), np.float32)
    py = np.zeros((max(n, 1), P), np.float32)
    adm = np.zeros(max(n, 1), np.uint8)
    costs = np.zeros(max(n, 1), np.float32)
    st = State(*start)
    lib().ko_full_cycle(coll.h if coll is not None else None, C.byref(ci.cx) if ci is not None else None,
                        C.byref(st), dt, P, _pd(vx), _pd(vy), _pd(omega), n, int(threads or host_threads()),
                        _pf(px), _pf(py), adm.ctypes.data_as(C.POINTER(C.c_uint8)), _pf(costs))
    raw = np.flatnonzero(adm[:n]).astype(np.int32)
    c = costs[raw]
    # cost_evaluator.cpp:101-104: strict `<` against FLT_MAX, lowest index wins ties
    ok = np.flatnonzero(c < np.finfo(np.float32).max)
    if len(ok):
        idx = int(ok[np.argmin(c[ok])])
        cost = float(c[idx])
    else:
        idx, cost = -1, 0.0
    return dict(px=px[raw], py=py[raw], raw=raw, costs=c, index=idx, cost=cost)


def costs_mt(ci: CostInputs, paths_x, paths_y, vel=None, threads=None):
    """Per-sample total cost of caller-provided trajectories (threaded)
    -> (argmin, min_cost, costs[N]) like min_trajectory_cost."""
    px, py = _f32(paths_x), _f32(paths_y)
    N, P = px.shape
    costs = np.zeros(N, np.float32)
    v = [_f32(a) for a in vel] if vel is not None else [None, None, None]
    vp = [_pf(a) if a is not None else None for a in v]
    lib().ko_costs_mt(C.byref(ci.cx), _pf(px), _pf(py), vp[0], vp[1], vp[2], N, P,
                      int(threads or host_threads()), _pf(costs))
    ok = np.flatnonzero(costs < np.finfo(np.float32).max)
    if len(ok):
        idx = int(ok[np.argmin(costs[ok])])
        return idx, float(costs[idx]), costs
    return -1, 0.0, costs
