"""How often does a moving dynamic window change its index PATTERN (which sample takes which axis value), and what
does such a cycle cost?  A velocity sweep through kc_dwa_find_best_path: cycle times, sample counts, slow cycles.
python tools/window_sweep.py [steps]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "kompass-core_amd")]
import numpy as np
import kompass_hip as kh, synthetic as syn

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
by_call = len(sys.argv) > 2 and sys.argv[2] == "calls"
parts = []
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
sx, sy, sz = (np.ascontiguousarray(seg[:, k]) for k in range(3))
rng = np.random.default_rng(1)
vx, om = 0.3, 0.0
lat, counts = [], []
for i in range(steps + 100):
    # a robot whose velocity wanders (bounded random walk), as the commands of the previous cycles make it
    vx = float(np.clip(vx + rng.normal(0, 0.02), -0.2, 1.0))
    om = float(np.clip(om + rng.normal(0, 0.05), -1.0, 1.0))
    st = (0.0, 0.0, 0.001 * (i % 7), 0.0)
    t = time.perf_counter()
    if by_call:   # the four entries one by one: where a pattern change costs
        ctx.sample_window(base["ctr"], lim, (vx, 0.0, om), 91, 91, want_list=False)
        t1 = time.perf_counter()
        ctx.set_points(st, pts, inp["max_range"])
        ctx.set_tracked_segment_columns(sx, sy, sz, sacc, inp["ref_len"])
        t2 = time.perf_counter()
        r = ctx.cycle(st, P)
        parts.append(((t1 - t) * 1e6, (t2 - t1) * 1e6, (time.perf_counter() - t2) * 1e6))
    else:
        r = ctx.find_best_path(st, P, window=(base["ctr"], lim, (vx, 0.0, om), 91, 91), points=pts,
                               max_sensor_range=inp["max_range"], segment=(seg, sacc, inp["ref_len"]))
    dt = time.perf_counter() - t
    if i >= 100:
        lat.append(dt * 1e6)
        counts.append(int(r.n_samples))
lat = np.array(lat)
changes = int(np.sum(np.diff(counts) != 0))
print("cycles %d | us p50 %.1f p90 %.1f p99 %.1f max %.1f mean %.1f | sample count changed %d times (%d distinct counts) | cycles above 1.5 x p50: %d"
      % (len(lat), np.percentile(lat, 50), np.percentile(lat, 90), np.percentile(lat, 99), lat.max(), lat.mean(), changes,
         len(set(counts)), int(np.sum(lat > 1.5 * np.percentile(lat, 50)))))
if by_call:
    pa = np.array(parts[100:])
    slow = lat > 1.5 * np.percentile(lat, 50)
    print("by call (window | points + segment | cycle), us: fast cycles %s, slow cycles %s"
          % (np.round(np.median(pa[~slow], axis=0), 1), np.round(np.median(pa[slow], axis=0), 1)))
half = lat[len(lat) // 2:]
print("second half: p50 %.1f p90 %.1f p99 %.1f mean %.1f | above 1.5 x p50: %d of %d | pattern hits %d builds %d"
      % (np.percentile(half, 50), np.percentile(half, 90), np.percentile(half, 99), half.mean(), int(np.sum(half > 1.5 * np.percentile(half, 50))),
         len(half), ctx.get_option("pattern_hits"), ctx.get_option("pattern_builds")))
ctx.close()
