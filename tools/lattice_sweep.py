"""Cycle time against the SIZE of the window lattice (max_linear_samples x max_angular_samples: the reference's default is
20 x 20), fresh inputs every cycle through kc_dwa_find_best_path, clutter and mid scenes: where does the single-launch
cycle apply, what do small lattices cost.  python tools/lattice_sweep.py [scene]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "kompass-core_amd")]
import numpy as np
import kompass_hip as kh, synthetic as syn

scene = sys.argv[1] if len(sys.argv) > 1 else "survey"
opts = dict(kv.split("=") for kv in sys.argv[2:])
inp = syn.make_controller_inputs("cfg2", seed=0, scene=scene)
base = syn.CONFIGS["cfg2"]
lim = kh.make_limits(syn.LIMITS["vx"], syn.LIMITS["vy"], syn.LIMITS["omega"])
P, S = inp["P"], len(inp["seg_xyz"])
seg = np.asarray(inp["seg_xyz"], np.float32)
sacc = np.ascontiguousarray(inp["acc_at_seg"], np.float32)
pts = np.ascontiguousarray(inp["points"], np.float32)
for L, A in ((11, 11), (21, 21), (31, 31), (45, 45), (61, 61), (91, 91), (101, 121), (131, 131), (181, 181)):
    ctx = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, 0), (0, 0, 0, 1), inp["octree_res"], inp["dt"],
                        max_samples=(L + 2) * (A + 2), max_points=P, max_segment=S, max_obstacles=len(pts), acc_limits=inp["acc_limits"])
    ctx.set_weights(kh.make_weights(*inp["weights"]))
    for k, v in opts.items():
        ctx.set_option(k, float(v))
    lat, ks = [], {}
    for i in range(500):
        st = (0.0, 0.0, 1e-3 * ((i % 7) - 3), 0.0)
        if i == 350:
            ctx.timing_enable(True)
        t = time.perf_counter()
        r = ctx.find_best_path(st, P, window=(base["ctr"], lim, (0.5, 0.0, 0.0), L, A), points=pts, max_sensor_range=inp["max_range"],
                               segment=(seg, sacc, inp["ref_len"]))
        dt = time.perf_counter() - t
        if 100 <= i < 350:
            lat.append(dt)
        if i >= 350:
            for nm, ms in ctx.timings():
                if not nm.startswith("host:"):
                    ks.setdefault(nm, []).append(ms)
    print("%3d x %3d: %5d samples, %5d admissible, cycle p50 %.1f us, single launch %d, kernels (last) %s" %
          (L, A, r.n_samples, r.n_admissible, np.percentile(lat, 50) * 1e6, ctx.get_option("last_cycle_single_launch"),
           {k: round(float(np.mean(v)) * 1e3, 1) for k, v in ks.items()}), flush=True)
    ctx.close()
