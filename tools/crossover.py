"""Diagnostic: cycle time of both cost kernels against the admissible count (cfg2 lattice,
obstacles nearer than r removed)."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kompass-core_amd"))
import numpy as np
import kompass_hip as kh, synthetic as syn

inp = syn.make_controller_inputs("cfg2", seed=0)
base = syn.CONFIGS["cfg2"]
vx, vy, om = syn.lattice_nonholonomic(base["n_vx"], base["n_om"])
P, S, O = inp["P"], len(inp["seg_xyz"]), len(inp["points"])
ctx = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, 0), (0, 0, 0, 1),
                    inp["octree_res"], inp["dt"], max_samples=len(vx), max_points=P,
                    max_segment=S, max_obstacles=O, acc_limits=inp["acc_limits"], device=0)
ctx.set_weights(kh.make_weights(*inp["weights"]))
ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
ctx.set_samples(vx, vy, om)
pts = np.asarray(inp["points"], dtype=np.float32).reshape(-1, 3)
for r in (0.0, 1.2, 1.5, 1.8, 2.1, 2.5, 3.0, 4.0):
    sel = pts[np.hypot(pts[:, 0], pts[:, 1]) > r] if r > 0 else pts
    ctx.set_points(inp["state"], sel, inp["max_range"])
    for _ in range(30): res = ctx.cycle((0.0, 0.0, 0.001, 0.0), P)
    ts = []
    for _ in range(300):
        t0 = time.perf_counter(); res = ctx.cycle((0.0, 0.0, 0.001, 0.0), P); ts.append(time.perf_counter() - t0)
    print(f"{os.environ.get('KC_COST_KERNEL','auto'):5s} r>{r:3.1f}: admissible {res.n_admissible:5d}  cycle {np.median(ts)*1e6:7.1f} us")
