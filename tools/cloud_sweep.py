"""Cycle time when the sensor data changes SIZE and extent every cycle (a real cloud does): random subsets of a scene's
points, 30-100 % of them, shifted a little, through kc_dwa_find_best_path.  python tools/cloud_sweep.py [steps]"""
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
allp = np.ascontiguousarray(inp["points"], np.float32)
rng = np.random.default_rng(3)
clouds = []
for k in range(64):
    m = int(len(allp) * rng.uniform(0.3, 1.0))
    sel = np.sort(rng.choice(len(allp), m, replace=False))
    clouds.append(np.ascontiguousarray(allp[sel] + np.float32([rng.uniform(-0.3, 0.3), rng.uniform(-0.3, 0.3), 0.0])))
lat, sizes = [], []
for i in range(steps + 100):
    pts = clouds[int(rng.integers(0, len(clouds)))]
    seglen = int(rng.integers(S // 2, S + 1))   # the tracked segment changes length too
    t = time.perf_counter()
    r = ctx.find_best_path((0.0, 0.0, 0.001 * (i % 7), 0.0), P, window=(base["ctr"], lim, (0.5, 0.0, 0.0), 91, 91), points=pts,
                           max_sensor_range=inp["max_range"], segment=(seg[:seglen], sacc[:seglen], inp["ref_len"]))
    dt = time.perf_counter() - t
    if i >= 100:
        lat.append(dt * 1e6)
        sizes.append(len(pts))
lat = np.array(lat)
print("cycles %d | points %d..%d | us p50 %.1f p90 %.1f p99 %.1f max %.1f mean %.1f | above 1.5 x p50: %d"
      % (len(lat), min(sizes), max(sizes), np.percentile(lat, 50), np.percentile(lat, 90), np.percentile(lat, 99), lat.max(), lat.mean(),
         int(np.sum(lat > 1.5 * np.percentile(lat, 50)))))
ctx.close()
