"""Phase clocks of the cycle kernel inside a fresh-input step (sensor update + window + segment every cycle);
needs kompass-core_amd/lib_stamps (make OUT=lib_stamps HIPFLAGS_EXTRA=-DKC_PHASE_STAMPS)."""
import os, sys, pathlib
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kompass-core_amd"))
os.environ["KC_DEBUG_STAMPS"] = "1"
import numpy as np
import kompass_hip as kh, synthetic as syn
kh.LIB_PATH = pathlib.Path(ROOT) / "kompass-core_amd" / "lib_stamps" / "libkompass_hip.so"
scene = sys.argv[1] if len(sys.argv) > 1 else "survey"
opts = dict(kv.split("=") for kv in sys.argv[2:])
cfg = "cfg2"
inp = syn.make_controller_inputs(cfg, seed=0, scene=scene)
base = syn.CONFIGS[cfg]
P, S = inp["P"], len(inp["seg_xyz"])
ctx = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, 0), (0, 0, 0, 1), inp["octree_res"], inp["dt"],
                    max_samples=8704, max_points=P, max_segment=S, max_obstacles=len(inp["points"]), acc_limits=inp["acc_limits"])
for k, v in opts.items():
    ctx.set_option(k, float(v))
ctx.set_weights(kh.make_weights(*inp["weights"]))
lim = kh.make_limits(syn.LIMITS["vx"], syn.LIMITS["vy"], syn.LIMITS["omega"])
seg = np.asarray(inp["seg_xyz"], np.float32)
sx, sy, sz = (np.ascontiguousarray(seg[:, k]) for k in range(3))
pts = np.ascontiguousarray(inp["points"], np.float32)
for i in range(30):
    st = (0.0, 0.0, 1e-3 * ((i % 7) - 3), 0.0)
    ctx.sample_window(base["ctr"], lim, (0.5 + 0.002 * ((i % 5) - 2), 0.0, 0.01 * ((i % 3) - 1)), 91, 91, want_list=False)
    ctx.set_points(st, pts, inp["max_range"])
    ctx.set_tracked_segment_columns(sx, sy, sz, inp["acc_at_seg"], inp["ref_len"])
    r = ctx.cycle(st, P)
print(scene, opts, "admissible", r.n_admissible, flush=True)
ctx.close()
