"""Seeded synthetic inputs of the BASELINE.json configurations.

Plain numpy; shared by tests/ and bench.py so the HIP path and the CPU oracle
are always fed the same bytes.  Nothing here computes a result.
"""
from __future__ import annotations

import math

import numpy as np

ACKERMANN, DIFFERENTIAL_DRIVE, OMNI = 0, 1, 2
CYLINDER, BOX, SPHERE = 0, 1, 2

# dwa_test.cpp:173-177,206-207 (robot + limits used by the reference's own
# closed-loop scenarios)
ROBOT_CYLINDER = dict(shape=CYLINDER, dims=[0.1, 0.4])
LIMITS = dict(vx=(1.0, 2.0, 2.0), vy=(1.0, 2.0, 2.0), omega=(2.0, 2.0, 3.0, 3.0))


def costmap_points(n_side: int, res: float = 0.05, seed: int = 0, p_occ: float = 0.02,
                   free_radius: float = 1.0, border: int = 2) -> np.ndarray:
    """Occupied cell centres (x, y, 0) of an n_side x n_side costmap centred on
    the robot: Bernoulli(p_occ) outside a free disc, plus a solid border."""
    rng = np.random.default_rng(seed)
    occ = rng.random((n_side, n_side)) < p_occ
    c = (np.arange(n_side) - n_side / 2.0 + 0.5) * res
    X, Y = np.meshgrid(c, c, indexing="ij")
    occ &= (X * X + Y * Y) > free_radius * free_radius
    occ[:border, :] = True
    occ[-border:, :] = True
    occ[:, :border] = True
    occ[:, -border:] = True
    pts = np.stack([X[occ], Y[occ], np.zeros(int(occ.sum()))], axis=1)
    return np.ascontiguousarray(pts, dtype=np.float32)


def window(cur, limits, dt):
    """Dynamic window of trajectory_sampler.cpp:328-372 (plain formula)."""
    vmax, acc, dec = limits
    return max(-vmax, cur - dec * dt), min(vmax, cur + acc * dt)


def lattice_nonholonomic(n_vx: int, n_omega: int, cur_vel=(0.5, 0.0, 0.0), dt: float = 0.1,
                         limits=LIMITS):
    """Exactly n_vx * n_omega (vx, 0, omega) samples inside the dynamic window,
    vx-major / omega-minor like trajectory_sampler.cpp:207-217; the vx axis is
    kept away from |vx| < 0.01 so no row is skipped."""
    lo, hi = window(cur_vel[0], limits["vx"], dt)
    lo = max(lo, 0.02)
    olo, ohi = window(cur_vel[2], (limits["omega"][1], limits["omega"][2], limits["omega"][3]), dt)
    vxs = lo + (hi - lo) * np.arange(n_vx) / max(n_vx - 1, 1)
    oms = olo + (ohi - olo) * np.arange(n_omega) / max(n_omega - 1, 1)
    vx = np.repeat(vxs, n_omega)
    om = np.tile(oms, n_vx)
    return vx.astype(np.float64), np.zeros_like(vx), om.astype(np.float64)


def lattice_omni(n_vx: int, n_vy: int, n_omega: int, cur_vel=(0.5, 0.0, 0.0), dt: float = 0.1,
                 limits=LIMITS):
    """Per vx: the (vx, vy, 0) block then the (vx, 0, omega) block
    (trajectory_sampler.cpp:256-272); n_vx * (n_vy + n_omega) samples."""
    lo, hi = window(cur_vel[0], limits["vx"], dt)
    lo = max(lo, 0.02)
    ylo, yhi = window(cur_vel[1], limits["vy"], dt)
    olo, ohi = window(cur_vel[2], (limits["omega"][1], limits["omega"][2], limits["omega"][3]), dt)
    vxs = lo + (hi - lo) * np.arange(n_vx) / max(n_vx - 1, 1)
    vys = ylo + (yhi - ylo) * np.arange(n_vy) / max(n_vy - 1, 1)
    oms = olo + (ohi - olo) * np.arange(n_omega) / max(n_omega - 1, 1)
    vx, vy, om = [], [], []
    for v in vxs:
        vx += [v] * n_vy
        vy += list(vys)
        om += [0.0] * n_vy
        vx += [v] * n_omega
        vy += [0.0] * n_omega
        om += list(oms)
    return np.array(vx), np.array(vy), np.array(om)


def straight_segment(n_points: int, spacing: float = 0.01, y: float = 0.0):
    """Tracked segment of a straight reference path along +x starting at the
    robot: (xyz [S,3] float32, acc_at_seg [S] float32, ref_len, seg)."""
    s = np.arange(n_points, dtype=np.float64) * spacing
    xyz = np.stack([s, np.full_like(s, y), np.zeros_like(s)], axis=1).astype(np.float32)
    return xyz, s.astype(np.float32)


def arc_segment(n_points: int, radius: float = 10.0, spacing: float = 0.01):
    """Tracked segment of the 3/4-circle path of controller_test_helpers.h:63-72
    starting at the robot pose (0, 0, yaw 0): circle centred at (0, radius)."""
    s = np.arange(n_points, dtype=np.float64) * spacing
    th = s / radius
    xyz = np.stack([radius * np.sin(th), radius * (1 - np.cos(th)), np.zeros_like(s)], axis=1)
    return xyz.astype(np.float32), s.astype(np.float32)


def dense_scan(n_beams: int, scale: float = 1.0):
    """benchmark_runner.cpp:112-121 generator: angles -pi + i*2pi/n, ranges
    5 + 2 sin(20 angle), optionally scaled to fill a larger grid."""
    ang = -math.pi + np.arange(n_beams, dtype=np.float64) * (2 * math.pi / n_beams)
    rng = (5.0 + 2.0 * np.sin(20.0 * ang)) * scale
    return ang, rng


# name -> parameters of the BASELINE.json configs (kernel-level sample counts)
CONFIGS = {
    "cfg1": dict(ctr=DIFFERENTIAL_DRIVE, n_vx=8, n_om=16, P=20, map_side=200, seg=201, path="straight",
                 weights=(1.0, 1.0, 1.0, 0.0, 0.0)),
    "cfg2": dict(ctr=DIFFERENTIAL_DRIVE, n_vx=64, n_om=128, P=50, map_side=500, seg=501, path="straight",
                 weights=(1.0, 1.0, 1.0, 0.0, 0.0)),
    "cfg3": dict(ctr=ACKERMANN, n_vx=128, n_om=256, P=100, map_side=1000, seg=1001, path="arc",
                 weights=(1.0, 1.0, 1.0, 0.0, 0.0)),
    "cfg5": dict(ctr=OMNI, n_vx=256, n_vy=64, n_om=192, P=50, map_side=500, seg=501, path="straight",
                 weights=(1.0, 1.0, 1.0, 1.0, 1.0)),
}


# Scenes (the costmap a config runs on).  "survey" is SURVEY.md 8(d)'s scene: Bernoulli(0.02) clutter
# outside a 1 m free disc + solid border -- it leaves 5 % of cfg2's samples admissible and none of
# cfg3's (10 s horizon), so the cost stage has little to do.  "mid" thins the clutter until roughly half
# of the samples survive (cfg2 56 %, cfg3 40 %, cfg5 55 %); "open" keeps only what lies beyond 10 m
# (border band), every sample admissible.
SCENES = {
    "survey": dict(p_occ=0.02, free_radius=1.0),
    "mid": dict(p_occ=0.005, free_radius=1.0),
    "open": dict(p_occ=0.02, free_radius=10.0),
}
MID_P_OCC = {"cfg3": 0.002}


def scene_points(name: str, scene: str = "survey", seed: int = 0) -> np.ndarray:
    c = CONFIGS[name]
    sc = dict(SCENES[scene])
    if scene == "mid":
        sc["p_occ"] = MID_P_OCC.get(name, sc["p_occ"])
    return costmap_points(c["map_side"], 0.05, seed, **sc)


def make_controller_inputs(name: str, seed: int = 0, scale: float = 1.0, scene: str = "survey"):
    """Everything one controller cycle of a BASELINE config consumes.
    scale < 1 shrinks the sample lattice (parity tests at oracle-friendly size)."""
    c = CONFIGS[name]
    nvx = max(2, int(round(c["n_vx"] * scale)))
    nom = max(3, int(round(c["n_om"] * scale)))
    if c["ctr"] == OMNI:
        nvy = max(3, int(round(c["n_vy"] * scale)))
        vx, vy, om = lattice_omni(nvx, nvy, nom)
    else:
        vx, vy, om = lattice_nonholonomic(nvx, nom)
    seg, acc = (straight_segment if c["path"] == "straight" else arc_segment)(c["seg"])
    pts = scene_points(name, scene, seed)
    return dict(
        name=name, scene=scene, ctr=c["ctr"], vx=vx, vy=vy, omega=om, P=c["P"], dt=0.1,
        state=(0.0, 0.0, 0.0, 0.0), points=pts, octree_res=0.05,
        seg_xyz=seg, acc_at_seg=acc, ref_len=12.0 if c["path"] == "straight" else 47.12389,
        weights=c["weights"], max_range=10.0, robot=ROBOT_CYLINDER,
        acc_limits=(LIMITS["vx"][1], LIMITS["vy"][1], LIMITS["omega"][2]),
    )


# ---------------------------------------------------------------------------
# The reference's own published benchmark workloads (inputs restated from
# src/kompass_cpp/benchmarks/benchmark_runner.cpp; data only)
# ---------------------------------------------------------------------------
def ref_cost5k_samples(n_samples: int = 5001, horizon: float = 10.0, dt: float = 0.01):
    """generate_heavy_trajectory_samples (benchmark_runner.cpp:36-89): one straight
    trajectory, then pairs with a lateral-velocity / a heading fluctuation of growing
    amplitude.  -> paths_x, paths_y [N, P] float32, vel (vx, vy, omega) [N, P-1] float32."""
    P = int(horizon / dt)
    v1, max_fl = 1.0, 0.5
    i = np.arange(P, dtype=np.float64)
    px, py, vx, vy, om = [], [], [], [], []

    def push(x, y, a, b, c):
        px.append(x.astype(np.float32)); py.append(y.astype(np.float32))
        vx.append(np.full(P - 1, a, np.float64).astype(np.float32) if np.isscalar(a) else a[:P - 1].astype(np.float32))
        vy.append(np.full(P - 1, b, np.float64).astype(np.float32) if np.isscalar(b) else b[:P - 1].astype(np.float32))
        om.append(np.full(P - 1, c, np.float64).astype(np.float32) if np.isscalar(c) else c[:P - 1].astype(np.float32))

    push(dt * v1 * i, np.zeros(P), v1, 0.0, 0.0)
    pairs = (n_samples - 1) // 2
    step = max_fl / (pairs if pairs > 0 else 1)
    for p in range(1, pairs + 1):
        amp = p * step
        fl = amp * np.sin(2 * math.pi * i / P)
        push(dt * v1 * i, dt * fl * i, v1, fl, 0.0)
        fa = amp * np.cos(2 * math.pi * i / P)
        push(dt * v1 * i * np.cos(fa), dt * v1 * i * np.sin(fa), v1, 0.0, fa)
    return (np.stack(px), np.stack(py), [np.stack(vx), np.stack(vy), np.stack(om)])


REF_COST5K = dict(path_points=[[0.0, 0.0, 0.0], [5.0, 0.0, 0.0], [10.0, 0.0, 0.0]], interpolation=0.01,
                  segment_length=1000.0, max_segment_points=1000,
                  acc_limits=(3.0, 3.0, 5.0),   # x_p(1,3,5), y_p(1,3,5), a_p(3.14,3,5,8): max accelerations
                  weights=(1.0, 1.0, 0.0, 1.0, 1.0))  # path, goal, (no obstacles set), smoothness, jerk
REF_MAPPER400 = dict(height=400, width=400, res=0.05, beams=3600)
