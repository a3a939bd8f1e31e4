"""python tools/shard_cycle.py [cfg] [scene] [world]: the cycle of ONE rank's shard of a fixed lattice (what every GPU of
a `world`-GPU node runs in the strong-scaling split, without the all-reduce), every shard in turn on this GPU."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kompass-core_amd"))
import numpy as np
import kompass_hip as kh, synthetic as syn, sharding
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
scene = sys.argv[2] if len(sys.argv) > 2 else "mid"
world = int(sys.argv[3]) if len(sys.argv) > 3 else 8
inp = syn.make_controller_inputs(cfg, seed=0, scene=scene)
P, S, n = inp["P"], len(inp["seg_xyz"]), len(inp["vx"])
ctx = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, 0), (0, 0, 0, 1), inp["octree_res"], inp["dt"],
                    max_samples=n, max_points=P, max_segment=S, max_obstacles=len(inp["points"]), acc_limits=inp["acc_limits"])
ctx.set_weights(kh.make_weights(*inp["weights"]))
ctx.set_points(inp["state"], inp["points"], inp["max_range"])
ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
out = []
for g in range(world):
    first, count = sharding.shard_range(n, g, world)
    ctx.set_shard(first, count)
    lat = []
    for i in range(260):
        t = time.perf_counter()
        r = ctx.cycle((0.0, 0.0, 1e-3 * (i % 7 - 3), 0.0), P)
        if i >= 60:
            lat.append(time.perf_counter() - t)
    lat = np.array(lat) * 1e6
    out.append((g, count, int(r.n_admissible), float(np.mean(lat)), float(np.percentile(lat, 50)), ctx.get_option("last_cycle_single_launch")))
for o in out:
    print(f"{cfg} {scene} rank {o[0]}/{world}: {o[1]} samples, {o[2]} admissible, cycle mean {o[3]:.1f} us p50 {o[4]:.1f} single_launch {o[5]:g}")
print(f"slowest rank: {max(o[3] for o in out):.1f} us")
ctx.close()
