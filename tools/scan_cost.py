"""Diagnostic: cost of kc_dwa_set_scan at typical laserscan sizes."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kompass-core_amd"))
import numpy as np
import kompass_hip as kh, synthetic as syn
inp = syn.make_controller_inputs("cfg2", seed=0)
base = syn.CONFIGS["cfg2"]
vx, vy, om = syn.lattice_nonholonomic(base["n_vx"], base["n_om"])
P, S = inp["P"], len(inp["seg_xyz"])
for n in (360, 720, 4096):
    ang, rng = syn.dense_scan(n, 1.0)
    ctx = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, 0.2), (0, 0, 0, 1),
                        inp["octree_res"], inp["dt"], max_samples=len(vx), max_points=P,
                        max_segment=S, max_obstacles=n, acc_limits=inp["acc_limits"], device=0)
    ctx.set_weights(kh.make_weights(*inp["weights"]))
    ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
    ctx.set_samples(vx, vy, om)
    ts, tc = [], []
    for i in range(120):
        r = rng * (1.0 + 0.001 * (i % 5))
        t0 = time.perf_counter(); ctx.set_scan(inp["state"], r, ang, 10.0); t1 = time.perf_counter()
        res = ctx.cycle((0.0, 0.0, 0.001, 0.0), P); t2 = time.perf_counter()
        ts.append(t1 - t0); tc.append(t2 - t1)
    ctx.timing_enable(True)
    acc = {}
    for i in range(20):
        ctx.cycle((0.0, 0.0, 0.001, 0.0), P)
        for k, v in ctx.timings():
            acc.setdefault(k, []).append(v)
    ctx.timing_enable(False)
    print("   kernels (us, HIP events):", {k: round(float(np.mean(v)) * 1e3, 1) for k, v in acc.items()})
    # the same cycle without the obstacle term / without the segment terms
    for label, w in (("no obstacle cost", (1.0, 1.0, 0.0, 0.0, 0.0)), ("obstacle cost only", (0.0, 0.0, 1.0, 0.0, 0.0))):
        ctx.set_weights(kh.make_weights(*w))
        ctx.timing_enable(True)
        acc = {}
        for i in range(20):
            ctx.cycle((0.0, 0.0, 0.001, 0.0), P)
            for k, v in ctx.timings():
                if "cost" in k:
                    acc.setdefault(k, []).append(v)
        ctx.timing_enable(False)
        print("   ", label, {k: round(float(np.mean(v)) * 1e3, 1) for k, v in acc.items()})
    ctx.set_weights(kh.make_weights(*inp["weights"]))
    print(f"set_scan({n:4d} beams) {np.median(ts[20:])*1e6:6.1f} us | following cycle {np.median(tc[20:])*1e6:6.1f} us | admissible {res.n_admissible}")
    ctx.close()
