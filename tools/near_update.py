"""python tools/near_update.py [cfg] [scene] [near_table]: cycles whose tracked segment changes EVERY cycle (what a
running controller does), so the near table is rebuilt every time; prints the mean / p50 of set_tracked_segment + cycle."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kompass-core_amd"))
import numpy as np
import kompass_hip as kh, synthetic as syn
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
scene = sys.argv[2] if len(sys.argv) > 2 else "open"
nt = float(sys.argv[3]) if len(sys.argv) > 3 else 128
inp = syn.make_controller_inputs(cfg, seed=0, scene=scene)
P, S = inp["P"], len(inp["seg_xyz"])
ctx = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, 0), (0, 0, 0, 1), inp["octree_res"], inp["dt"],
                    max_samples=len(inp["vx"]), max_points=P, max_segment=S, max_obstacles=len(inp["points"]),
                    acc_limits=inp["acc_limits"])
ctx.set_option("near_table", nt)
ctx.set_weights(kh.make_weights(*inp["weights"]))
ctx.set_points(inp["state"], inp["points"], inp["max_range"])
ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
segs = []
for k in range(4):
    xyz = np.asarray(inp["seg_xyz"], np.float32).copy()
    xyz[:, 1] += np.float32(0.01 * k)
    segs.append(xyz)
lat = []
for i in range(600):
    t = time.perf_counter()
    ctx.set_tracked_segment(segs[i % 4], inp["acc_at_seg"], inp["ref_len"])
    r = ctx.cycle((0.0, 0.0, 1e-3 * (i % 7 - 3), 0.0), P)
    if i >= 100:
        lat.append(time.perf_counter() - t)
lat = np.array(lat) * 1e6
print(f"{cfg} {scene} near_table={nt:g}: segment + cycle mean {lat.mean():.1f} us, p50 {np.percentile(lat, 50):.1f}; admissible {r.n_admissible}")
ctx.close()
