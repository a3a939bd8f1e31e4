"""set_points back to back at a config's obstacle list (cfg3: 27 833 points -> the multi-workgroup sensor build).
KC_DEBUG_HOST=1 prints the host's BAR copy time."""
import sys, time
sys.path.insert(0, "/root/repo/kompass-core_amd")
import numpy as np
import kompass_hip as kh, synthetic as syn
name = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
inp = syn.make_controller_inputs(name, seed=0)
P, S = inp["P"], len(inp["seg_xyz"])
ctx = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, 0), (0, 0, 0, 1), inp["octree_res"], inp["dt"], max_samples=len(inp["vx"]),
                    max_points=P, max_segment=S, max_obstacles=len(inp["points"]), acc_limits=inp["acc_limits"])
ctx.set_weights(kh.make_weights(*inp["weights"]))
pts = np.ascontiguousarray(inp["points"], dtype=np.float32)
if len(sys.argv) > 3:   # a subset of the points (every k-th)
    pts = np.ascontiguousarray(pts[:: max(1, len(pts) // int(sys.argv[3]))][: int(sys.argv[3])])
print(name, "points", len(pts))
ctx.set_points(inp["state"], pts, inp["max_range"])
ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
for i in range(5): ctx.cycle((0.0, 0.0, 1e-3 * (i % 7 - 3), 0.0), P)
ts = []
for i in range(reps):
    t0 = time.perf_counter(); ctx.set_points(inp["state"], pts, inp["max_range"]); ts.append(time.perf_counter() - t0)
print("set_points back to back: median %.1f us, min %.1f" % (np.median(ts) * 1e6, np.min(ts) * 1e6))
ts = []
for i in range(reps):
    t0 = time.perf_counter(); ctx.set_points(inp["state"], pts, inp["max_range"]); ctx.cycle((0.0, 0.0, 0.0, 0.0), P); ts.append(time.perf_counter() - t0)
print("set_points + cycle: median %.1f us" % (np.median(ts) * 1e6))
ts = []
for i in range(reps):
    t0 = time.perf_counter(); ctx.cycle((0.0, 0.0, 0.0, 0.0), P); ts.append(time.perf_counter() - t0)
print("cycle alone: median %.1f us" % (np.median(ts) * 1e6))
