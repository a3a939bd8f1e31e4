"""Where the host side of a fresh-input controller step goes (bench.py's fresh_inputs leg, call by call):
python tools/fresh_breakdown.py [scene]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "kompass-core_amd")]
import numpy as np
import kompass_hip as kh, synthetic as syn

scene = sys.argv[1] if len(sys.argv) > 1 else "survey"
opts = dict(kv.split("=") for kv in sys.argv[2:])
cfg = "cfg2"
inp = syn.make_controller_inputs(cfg, seed=0, scene=scene)
base = syn.CONFIGS[cfg]
P, S = inp["P"], len(inp["seg_xyz"])
ctx = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, 0), (0, 0, 0, 1), inp["octree_res"], inp["dt"],
                    max_samples=8704, max_points=P, max_segment=S, max_obstacles=len(inp["points"]), acc_limits=inp["acc_limits"])
for k, v in opts.items():
    ctx.set_option(k, float(v))
ctx.set_weights(kh.make_weights(*inp["weights"]))
lim = kh.make_limits(syn.LIMITS["vx"], syn.LIMITS["vy"], syn.LIMITS["omega"])
pose = lambda i: (0.0, 0.0, 1e-3 * ((i % 7) - 3), 0.0)
cur = lambda i: (0.5 + 0.002 * ((i % 5) - 2), 0.0, 0.01 * ((i % 3) - 1))
seg = np.asarray(inp["seg_xyz"], np.float32)
sx, sy, sz = (np.ascontiguousarray(seg[:, k]) for k in range(3))
pts = np.ascontiguousarray(inp["points"], np.float32)
names = ["sample_window", "set_points", "set_tracked_segment", "cycle"]
acc = {k: [] for k in names + ["total"]}
for i in range(1200):
    st = pose(i)
    t0 = time.perf_counter()
    ctx.sample_window(base["ctr"], lim, cur(i), 91, 91, want_list=False)
    t1 = time.perf_counter()
    ctx.set_points(st, pts, inp["max_range"])
    t2 = time.perf_counter()
    ctx.set_tracked_segment_columns(sx, sy, sz, inp["acc_at_seg"], inp["ref_len"])
    t3 = time.perf_counter()
    r = ctx.cycle(st, P)
    t4 = time.perf_counter()
    if i >= 200:
        for k, a, b in zip(names, (t0, t1, t2, t3), (t1, t2, t3, t4)):
            acc[k].append((b - a) * 1e6)
        acc["total"].append((t4 - t0) * 1e6)
print(scene, opts, "admissible", r.n_admissible, "| us per call (median / mean):")
for k in names + ["total"]:
    print(f"  {k:22s} {np.median(acc[k]):7.1f} / {np.mean(acc[k]):7.1f}")
ctx.close()
