"""Diagnostic: cycle time when every sample is admissible (robot in open space) at cfg2."""
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

def timed(fn, n=300):
    for _ in range(20): fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
    ts.sort(); return ts[len(ts) // 2] * 1e6, r

pts = np.asarray(inp["points"], dtype=np.float32).reshape(-1, 3)
for name, sel in (("cfg2 as benched", pts),
                  ("obstacles farther than 6 m only", pts[np.hypot(pts[:, 0], pts[:, 1]) > 6.0]),
                  ("obstacles farther than 10 m only", pts[np.hypot(pts[:, 0], pts[:, 1]) > 10.0])):
    ctx.set_points(inp["state"], sel, inp["max_range"])
    t, r = timed(lambda: ctx.cycle((0.0, 0.0, 0.001, 0.0), P))
    print(f"{name:36s}: {len(sel):5d} pts, cycle {t:7.1f} us, admissible {r.n_admissible} / {r.n_samples}")
