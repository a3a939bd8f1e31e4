"""COST_PARITY_JSON dump in the reference's schema (row J1 of VERDICT r1).

The reference's CPU/GPU parity harness runs its cost_evaluator_test twice (CPU
build, GPU build), each run leaving `{schema_version: 1, backend, tests: {name:
{costs: [...]}}}` at $COST_PARITY_JSON (src/kompass_cpp/tests/
cost_evaluator_test.cpp:159-207), and tests/test_cost_parity.py:132-188 compares
the two dumps cost by cost (relative 1e-4).  This module restates the inputs of
the twelve named cases (:217-461, helpers :34-142) once and evaluates them with
either backend of this repo:

    backend "cpu": the CPU oracle (oracle/ko.py)               -- the checker
    backend "hip": kc_cost_evaluate through the C ABI (MI355X) -- the product

    COST_PARITY_JSON=/tmp/hip.json python tests/cost_parity.py --backend hip
    COST_PARITY_JSON=/tmp/cpu.json python tests/cost_parity.py --backend cpu
    python tests/cost_parity.py --compare /tmp/cpu.json /tmp/hip.json

`--label gpu` writes the backend string the reference's own comparison script
expects from its device build.  tests/test_cost_parity_json.py runs all of it.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "kompass-core_amd", ROOT / "tests"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

REL_TOL = 1e-4  # tests/test_cost_parity.py:32 of the reference
WEIGHT_NAMES = ("path", "goal", "obstacles", "smoothness", "jerk")


def _straight(length=10.0, interp=1.0, seg=5.0):
    return ("straight", length, interp, seg)


def _circle34(R, n, interp, seg):
    return ("circle34", R, n, interp, seg)


Z5 = [(0.0, 0.0, 0.0)] * 5
_R = 2.0

# name -> evaluations, each (weight name, reference path, trajectory points, velocities or None, obstacles or None);
# order inside a case = order of the evalCost() calls of that Boost test case
CASES = {
    "goal_cost_on_straight_path": [("goal", _straight(), [(4.0, 0.0, 0.0)] * 5, None, None)],
    "goal_cost_arc_remaining_on_curved_path": [
        ("goal", _circle34(_R, 60, 0.05, 20.0), [(_R * math.cos(0.5), _R * math.sin(0.5), 0.0)] * 5, None, None),
        ("goal", _circle34(_R, 60, 0.05, 20.0), [(1.5, -0.5, 0.0)] * 5, None, None)],
    "goal_cost_tie_breaker": [("goal", _straight(), [(4.0, 0.1, 0.0)] * 5, None, None),
                              ("goal", _straight(), [(4.0, 0.5, 0.0)] * 5, None, None)],
    "path_cost_centered_sample": [("path", _straight(), [(float(i), 0.0, 0.0) for i in range(5)], None, None)],
    "path_cost_constant_lateral_offset": [("path", _straight(), [(float(i), 0.5, 0.0) for i in range(5)], None, None)],
    "smoothness_cost_constant_velocity": [("smoothness", _straight(), Z5, [(1.0, 0, 0)] * 4, None)],
    "smoothness_cost_single_step_change": [("smoothness", _straight(), Z5,
                                            [(0, 0, 0), (1, 0, 0), (1, 0, 0), (1, 0, 0)], None)],
    "jerk_cost_constant_acceleration": [("jerk", _straight(), Z5,
                                         [(0.1, 0, 0), (0.2, 0, 0), (0.3, 0, 0), (0.4, 0, 0)], None)],
    "jerk_cost_known_second_diff": [("jerk", _straight(), Z5, [(0, 0, 0), (1, 0, 0), (3, 0, 0), (6, 0, 0)], None)],
    "obstacles_cost_at_max_range": [("obstacles", _straight(), Z5, None, [(20.0, 0.0, 0.0)])],
    "obstacles_cost_at_zero_distance": [("obstacles", _straight(), Z5, None, [(0.0, 0.0, 0.0)])],
    "obstacles_cost_at_half_range": [("obstacles", _straight(), Z5, None, [(5.0, 0.0, 0.0)])],
}


def _reference_path(spec):
    """Path::Path + interpolate + segment (host glue; inputs of both backends)."""
    from oracle import ko

    if spec[0] == "straight":
        _, length, interp, seg = spec
        p = ko.Path([[0, 0, 0], [length, 0, 0]])
    else:
        _, R, n, interp, seg = spec
        mx = 3.0 * math.pi / 2.0
        p = ko.Path([[R * math.cos(i / (n - 1) * mx), R * math.sin(i / (n - 1) * mx), 0.0] for i in range(n)])
        interp = np.float32(interp)
    p.interpolate(interp)
    p.segment(seg, 10000)
    s0, s1 = p.segment_range(0)
    seg_xyz = np.stack([p.x[s0:s1 + 1], p.y[s0:s1 + 1], p.z[s0:s1 + 1]], axis=1).astype(np.float32)
    return dict(seg=seg_xyz, s0=s0, acc=np.asarray(p.acc, np.float32).copy(), total=float(p.total_length))


def _arrays(pts, vels):
    pts = np.asarray(pts, np.float32).reshape(-1, 3)
    n = len(pts)
    v = np.asarray(vels if vels is not None else [(0.0, 0.0, 0.0)] * (n - 1), np.float32).reshape(n - 1, 3)
    return pts[:, 0][None, :].copy(), pts[:, 1][None, :].copy(), [v[:, k][None, :].copy() for k in range(3)]


def eval_cpu(wname, spec, pts, vels, obstacles):
    """evalCost() with the CPU oracle."""
    from oracle import ko

    ref = _reference_path(spec)
    w = {k: 0.0 for k in WEIGHT_NAMES}
    w[wname] = 1.0
    obs_xy = None
    if obstacles:
        ox, oy = ko.obstacles_from_points((0, 0, 0), (0, 0, 0, 1), (0, 0, 0, 0), obstacles)
        obs_xy = np.stack([ox, oy], axis=1)
    ci = ko.CostInputs(ref["seg"], ref["s0"], ref["acc"], ref["total"], obs_xy,
                       max_obstacles_dist=np.float32(30.0) / np.float32(3.0), acc_limits=(1.0, 1.0, 1.0),
                       weights=ko.make_weights(**w))
    px, py, vel = _arrays(pts, vels)
    idx, cost, _ = ko.min_trajectory_cost(ci, px, py, vel)
    assert idx == 0, "CostEvaluator did not find a trajectory"
    return float(cost)


def eval_hip(wname, spec, pts, vels, obstacles):
    """evalCost() through the C ABI: kc_dwa_set_tracked_segment / kc_dwa_set_points / kc_cost_evaluate."""
    import kompass_hip as kh
    import synthetic as syn

    ref = _reference_path(spec)
    w = {k: 0.0 for k in WEIGHT_NAMES}
    w[wname] = 1.0
    ctx = kh.DwaContext(syn.CYLINDER, [0.1, 0.4], max_samples=4, max_points=8, acc_limits=(1, 1, 1),
                        max_segment=len(ref["seg"]))
    ctx.set_weights(kh.make_weights(**w))
    s0 = ref["s0"]
    ctx.set_tracked_segment(ref["seg"], ref["acc"][s0:s0 + len(ref["seg"])], ref["total"])
    if obstacles:
        ctx.set_points((0, 0, 0, 0), np.float32(obstacles), 30.0)
    px, py, vel = _arrays(pts, vels)
    r, costs = ctx.cost_evaluate(px, py, vel)
    assert r.found and r.index == 0, "CostEvaluator did not find a trajectory"
    ctx.close()
    return float(costs[0])


def records(backend):
    fn = eval_hip if backend == "hip" else eval_cpu
    return {name: {"costs": [fn(*ev) for ev in evs]} for name, evs in CASES.items()}


def dump(backend, path, label=None):
    doc = {"schema_version": 1, "backend": label or backend, "tests": records(backend)}
    with open(path, "w") as f:
        json.dump(doc, f, indent=2)
        f.write("\n")
    return doc


def compare(cpu, dev, rel_tol=REL_TOL):
    """The comparison of the reference's tests/test_cost_parity.py:132-188; returns the rows
    (name, index, cpu, device, delta, rel), worst first, and the failures."""
    only_cpu = set(cpu["tests"]) - set(dev["tests"])
    only_dev = set(dev["tests"]) - set(cpu["tests"])
    assert not only_cpu and not only_dev, (sorted(only_cpu), sorted(only_dev))
    rows = []
    for name in sorted(cpu["tests"]):
        a, b = cpu["tests"][name]["costs"], dev["tests"][name]["costs"]
        assert len(a) == len(b), (name, len(a), len(b))
        for i, (c, g) in enumerate(zip(a, b)):
            delta = abs(c - g)
            rows.append((name, i, c, g, delta, delta / max(abs(c), 1e-6)))
    rows.sort(key=lambda r: r[5], reverse=True)
    return rows, [r for r in rows if r[5] > rel_tol]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backend", choices=["cpu", "hip"], default="hip")
    ap.add_argument("--label", default=None, help="backend string to write (e.g. 'gpu' for the reference's own script)")
    ap.add_argument("--compare", nargs=2, metavar=("CPU_JSON", "DEVICE_JSON"))
    a = ap.parse_args()
    if a.compare:
        rows, bad = compare(json.load(open(a.compare[0])), json.load(open(a.compare[1])))
        print(f"{'Test':<45} {'Idx':>3} {'CPU':>14} {'device':>14} {'delta':>12} {'rel':>10}")
        for name, i, c, g, d, r in rows:
            print(f"{name:<45} {i:>3} {c:>14.6f} {g:>14.6f} {d:>12.3e} {r:>10.2e}{'' if r <= REL_TOL else '  <-- DRIFT'}")
        sys.exit(1 if bad else 0)
    out = os.environ.get("COST_PARITY_JSON", "")
    if not out:  # like the reference fixture: no variable, no file
        print("COST_PARITY_JSON is not set: nothing written", file=sys.stderr)
        return
    doc = dump(a.backend, out, a.label)
    print(f"wrote {out}: backend={doc['backend']}, {len(doc['tests'])} tests")


if __name__ == "__main__":
    main()
