"""Phase clocks of the long-list cost kernel on a ring-shaped laserscan scene.
Needs a library built with -DKC_PHASE_STAMPS (make OUT=lib_stamps HIPFLAGS=...)."""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kompass-core_amd"))
os.environ["KC_DEBUG_STAMPS"] = "1"
import pathlib
import numpy as np
import kompass_hip as kh, synthetic as syn
kh.LIB_PATH = pathlib.Path(ROOT) / "kompass-core_amd" / "lib_stamps" / "libkompass_hip.so"
inp = syn.make_controller_inputs("cfg2", seed=0)
base = syn.CONFIGS["cfg2"]
vx, vy, om = syn.lattice_nonholonomic(base["n_vx"], base["n_om"])
P, S = inp["P"], len(inp["seg_xyz"])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 720
w = tuple(float(v) for v in sys.argv[2].split(",")) if len(sys.argv) > 2 else inp["weights"]
ang, rng = syn.dense_scan(n, 1.0)
ctx = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, 0.2), (0, 0, 0, 1),
                    inp["octree_res"], inp["dt"], max_samples=len(vx), max_points=P,
                    max_segment=S, max_obstacles=n, acc_limits=inp["acc_limits"], device=0)
ctx.set_weights(kh.make_weights(*w))
ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
ctx.set_samples(vx, vy, om)
ctx.set_scan(inp["state"], rng, ang, 10.0)
for i in range(10):
    res = ctx.cycle((0.0, 0.0, 0.001, 0.0), P)
print("admissible", res.n_admissible, flush=True)
ctx.close()
