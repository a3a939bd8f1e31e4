"""Point lists as sensors deliver them: a depth image with holes (30 % of the points NaN) and sensor-frame lists
(global_frame = false), 8 k .. 300 k points: kc_dwa_set_points by variant.  python tools/frame_nan_sweep.py"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "kompass-core_amd")]
import numpy as np
import kompass_hip as kh, synthetic as syn

inp = syn.make_controller_inputs("cfg2", seed=0)
P, S = inp["P"], len(inp["seg_xyz"])
rng = np.random.default_rng(0)
for n in (8_000, 30_000, 100_000, 300_000):
    th, rad = rng.uniform(-np.pi, np.pi, n), rng.uniform(1.5, 9.0, n)
    pts = np.stack([rad * np.cos(th), rad * np.sin(th), rng.uniform(-0.2, 1.5, n)], 1).astype(np.float32)
    holes = pts.copy()
    holes[rng.random(n) < 0.3] = np.nan
    ctx = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0.1, 0, 0.3), (0, 0, 0, 1), inp["octree_res"], inp["dt"], max_samples=len(inp["vx"]),
                        max_points=P, max_segment=S, max_obstacles=n, acc_limits=inp["acc_limits"])
    ctx.set_weights(kh.make_weights(*inp["weights"]))
    ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
    ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
    out = []
    for label, data, gf in (("world frame", pts, True), ("world frame, 30 % NaN", holes, True), ("sensor frame", pts, False), ("sensor frame, 30 % NaN", holes, False)):
        ts = []
        for i in range(60):
            st = (0.0, 0.0, 1e-3 * (i % 7), 0.0)
            t0 = time.perf_counter()
            ctx.set_points(st, data, inp["max_range"], global_frame=gf)
            t1 = time.perf_counter()
            ctx.cycle(st, P)
            if i >= 20:
                ts.append((t1 - t0) * 1e6)
        out.append("%s %.1f" % (label, np.percentile(ts, 50)))
    print("%7d points, set_points p50 us: %s" % (n, " | ".join(out)), flush=True)
    ctx.close()
