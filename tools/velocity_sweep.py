"""The fresh-input cycle (kc_dwa_find_best_path, cfg2-sized window) against the robot's CURRENT velocity, held still:
what a cycle costs at each corner of the velocity range (the workload changes with it: longer roll-outs, more poses
near obstacles), to tell that apart from what a MOVING window costs (tools/window_sweep.py).
python tools/velocity_sweep.py [scene] [option=value ...]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "kompass-core_amd")]
import numpy as np
import kompass_hip as kh, synthetic as syn

scene = sys.argv[1] if len(sys.argv) > 1 else "survey"
inp = syn.make_controller_inputs("cfg2", seed=0, scene=scene)
base = syn.CONFIGS["cfg2"]
lim = kh.make_limits(syn.LIMITS["vx"], syn.LIMITS["vy"], syn.LIMITS["omega"])
P, S = inp["P"], len(inp["seg_xyz"])
ctx = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, 0), (0, 0, 0, 1), inp["octree_res"], inp["dt"],
                    max_samples=16384, max_points=P, max_segment=S, max_obstacles=len(inp["points"]), acc_limits=inp["acc_limits"])
for kv in sys.argv[2:]:
    ctx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
ctx.set_weights(kh.make_weights(*inp["weights"]))
seg = np.asarray(inp["seg_xyz"], np.float32)
sacc = np.ascontiguousarray(inp["acc_at_seg"], np.float32)
pts = np.ascontiguousarray(inp["points"], np.float32)
for vx in (-0.2, 0.0, 0.3, 0.5, 0.8, 1.0):
    for om in (-1.0, 0.0, 0.6):
        lat, ks = [], {}
        for i in range(500):
            st = (0.0, 0.0, 0.001 * (i % 7), 0.0)
            if i == 350:
                ctx.timing_enable(True)
            t = time.perf_counter()
            r = ctx.find_best_path(st, P, window=(base["ctr"], lim, (vx, 0.0, om), 91, 91), points=pts,
                                   max_sensor_range=inp["max_range"], segment=(seg, sacc, inp["ref_len"]))
            dt = time.perf_counter() - t
            if 100 <= i < 350:
                lat.append(dt * 1e6)
            if i >= 350:
                for nm, ms in ctx.timings():
                    if not nm.startswith("host:"):
                        ks.setdefault(nm, []).append(ms)
        ctx.timing_enable(False)
        print("current velocity (%.1f, %.1f): %5d samples %5d admissible | cycle p50 %.1f us | kernels %s" %
              (vx, om, r.n_samples, r.n_admissible, np.percentile(lat, 50), {k.replace("_kernel", ""): round(float(np.mean(v)) * 1e3, 1) for k, v in ks.items()}), flush=True)
ctx.close()
