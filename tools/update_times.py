import sys, time
sys.path.insert(0, "/root/repo/kompass-core_amd")
import numpy as np
import kompass_hip as kh, synthetic as syn
inp = syn.make_controller_inputs("cfg2", seed=0)
P, S = inp["P"], len(inp["seg_xyz"])
ctx = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, 0), (0, 0, 0, 1), inp["octree_res"], inp["dt"], max_samples=len(inp["vx"]),
                    max_points=P, max_segment=S, max_obstacles=len(inp["points"]), acc_limits=inp["acc_limits"])
ctx.set_weights(kh.make_weights(*inp["weights"]))
ctx.set_points(inp["state"], inp["points"], inp["max_range"])
ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
for i in range(50): ctx.cycle((0.0, 0.0, 1e-3 * (i % 7 - 3), 0.0), P)
def med(fn, n=40):
    ts = []
    for i in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    return np.median(ts) * 1e6
seg = np.ascontiguousarray(inp["seg_xyz"], dtype=np.float32)
print("set_tracked_segment back to back: %.1f us" % med(lambda: ctx.set_tracked_segment(seg, inp["acc_at_seg"], inp["ref_len"])))
def both():
    ctx.set_tracked_segment(seg, inp["acc_at_seg"], inp["ref_len"]); ctx.cycle((0.0, 0.0, 0.0, 0.0), P)
print("segment + cycle: %.1f us" % med(both))
print("cycle alone: %.1f us" % med(lambda: ctx.cycle((0.0, 0.0, 0.0, 0.0), P)))
print("set_points back to back: %.1f us" % med(lambda: ctx.set_points(inp["state"], inp["points"], inp["max_range"])))
