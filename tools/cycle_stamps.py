"""Phase clocks of the single-launch cycle kernel (library built with -DKC_PHASE_STAMPS into lib_stamps/):
python tools/cycle_stamps.py [cfg] [scene]"""
import os, sys, pathlib
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kompass-core_amd"))
os.environ["KC_DEBUG_STAMPS"] = "1"
import numpy as np
import kompass_hip as kh, synthetic as syn
kh.LIB_PATH = pathlib.Path(ROOT) / "kompass-core_amd" / "lib_stamps" / "libkompass_hip.so"
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
scene = sys.argv[2] if len(sys.argv) > 2 else "survey"
inp = syn.make_controller_inputs(cfg, seed=0, scene=scene)
P, S = inp["P"], len(inp["seg_xyz"])
Pc = int(os.environ.get("KC_TOOL_P", P))   # (a shorter horizon for this run: more samples survive)
ctx = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, 0), (0, 0, 0, 1), inp["octree_res"], inp["dt"],
                    max_samples=len(inp["vx"]), max_points=P, max_segment=S, max_obstacles=len(inp["points"]),
                    acc_limits=inp["acc_limits"])
w = tuple(float(v) for v in sys.argv[3].split(",")) if len(sys.argv) > 3 else inp["weights"]
ctx.set_weights(kh.make_weights(*w))
ctx.set_points(inp["state"], inp["points"], inp["max_range"])
ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
for i in range(20):
    res = ctx.cycle((0.0, 0.0, 1e-3 * (i % 7 - 3), 0.0), Pc)
print(cfg, scene, "admissible", res.n_admissible, "single launch", ctx.get_option("last_cycle_single_launch"), flush=True)
ctx.close()
