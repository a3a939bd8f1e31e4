"""Cycle time when the prediction horizon changes every cycle (adaptPredictionHorizonToCurvature, dwa.cpp:157-206: P <= the
constructed P, new per call): kc_dwa_find_best_path with num_points drawn from [P/2, P].  python tools/horizon_sweep.py [steps]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "kompass-core_amd")]
import numpy as np
import kompass_hip as kh, synthetic as syn

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
inp = syn.make_controller_inputs("cfg2", seed=0, scene="survey")
base = syn.CONFIGS["cfg2"]
lim = kh.make_limits(syn.LIMITS["vx"], syn.LIMITS["vy"], syn.LIMITS["omega"])
P, S = inp["P"], len(inp["seg_xyz"])
ctx = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, 0), (0, 0, 0, 1), inp["octree_res"], inp["dt"],
                    max_samples=16384, max_points=P, max_segment=S, max_obstacles=len(inp["points"]),
                    acc_limits=inp["acc_limits"])
ctx.set_weights(kh.make_weights(*inp["weights"]))
seg = np.asarray(inp["seg_xyz"], np.float32)
sacc = np.ascontiguousarray(inp["acc_at_seg"], np.float32)
pts = np.ascontiguousarray(inp["points"], np.float32)
rng = np.random.default_rng(4)
sx, sy, sz = (np.ascontiguousarray(seg[:, k]) for k in range(3))
fixed = len(sys.argv) > 2 and sys.argv[2] == "fixed"
fixed_p = int(sys.argv[3]) if len(sys.argv) > 3 else P
adm = []
lat, parts = [], []
for i in range(steps + 100):
    p = int(rng.integers(P // 2, P + 1))
    st = (0.0, 0.0, 0.001 * (i % 7), 0.0)
    if fixed:
        p = fixed_p
    t = time.perf_counter()
    ctx.sample_window(base["ctr"], lim, (0.5, 0.0, 0.0), 91, 91, want_list=False)
    t1 = time.perf_counter()
    ctx.set_points(st, pts, inp["max_range"])
    t2 = time.perf_counter()
    ctx.set_tracked_segment_columns(sx, sy, sz, sacc, inp["ref_len"])
    t3 = time.perf_counter()
    r = ctx.cycle(st, p)
    t4 = time.perf_counter()
    if i >= 100:
        lat.append((t4 - t) * 1e6)
        parts.append(((t1 - t) * 1e6, (t2 - t1) * 1e6, (t3 - t2) * 1e6, (t4 - t3) * 1e6))
        adm.append(int(r.n_admissible))
lat = np.array(lat)
print("cycles %d | P in [%d, %d] | us p50 %.1f p90 %.1f p99 %.1f max %.1f mean %.1f | above 1.5 x p50: %d"
      % (len(lat), P // 2, P, np.percentile(lat, 50), np.percentile(lat, 90), np.percentile(lat, 99), lat.max(), lat.mean(),
         int(np.sum(lat > 1.5 * np.percentile(lat, 50)))))
print("by call (window | points | segment | cycle), us median:", np.round(np.median(np.array(parts), axis=0), 1), ("fixed P = %d" % fixed_p) if fixed else "changing P", "| admissible median %d" % int(np.median(adm)))
ctx.close()
