"""python tools/run_cycles.py cfg scene w_path,w_goal,w_obs,w_smooth,w_jerk [cycles]: plain cycles of one scene (for
rocprofv3 counter passes)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kompass-core_amd"))
import kompass_hip as kh, synthetic as syn
cfg, scene = sys.argv[1], sys.argv[2]
inp = syn.make_controller_inputs(cfg, seed=0, scene=scene)
w = tuple(float(v) for v in sys.argv[3].split(",")) if len(sys.argv) > 3 and sys.argv[3] else inp["weights"]
n = int(sys.argv[4]) if len(sys.argv) > 4 else 50
P, S = inp["P"], len(inp["seg_xyz"])
ctx = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, 0), (0, 0, 0, 1), inp["octree_res"], inp["dt"],
                    max_samples=len(inp["vx"]), max_points=P, max_segment=S, max_obstacles=len(inp["points"]),
                    acc_limits=inp["acc_limits"])
ctx.set_weights(kh.make_weights(*w))
ctx.set_points(inp["state"], inp["points"], inp["max_range"])
ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
for i in range(n):
    r = ctx.cycle((0.0, 0.0, 1e-3 * (i % 7 - 3), 0.0), P)
print(cfg, scene, w, "admissible", r.n_admissible, "cost", r.cost)
