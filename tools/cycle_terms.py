"""Cycle time of one config / scene by cost term: python tools/cycle_terms.py cfg2 mid [option=value ...]
(scene "scan": bench.py's laserscan_room leg -- 1440 beams, ranges 4 + 1.5 cos 5a; "scanN": N beams)"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kompass-core_amd"))
import numpy as np
import kompass_hip as kh, synthetic as syn
cfg, scene = sys.argv[1], sys.argv[2]
beams = int(scene[4:] or 1440) if scene.startswith("scan") else 0
inp = syn.make_controller_inputs(cfg, seed=0, scene="survey" if beams else scene)
P, S = inp["P"], len(inp["seg_xyz"])
ctx = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, 0), (0, 0, 0, 1), inp["octree_res"], inp["dt"],
                    max_samples=len(inp["vx"]), max_points=P, max_segment=S, max_obstacles=max(len(inp["points"]), beams, 16),
                    acc_limits=inp["acc_limits"])
for kv in sys.argv[3:]:
    ctx.set_option(kv.split("=")[0], float(kv.split("=")[1]))
ctx.timing_enable(True)
if beams:
    ang = np.linspace(-np.pi, np.pi, beams, endpoint=False)
    ctx.set_scan(inp["state"], 4.0 + 1.5 * np.cos(5 * ang), ang, inp["max_range"])
else:
    ctx.set_points(inp["state"], inp["points"], inp["max_range"])
ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
print('default weights', inp['weights'])
for w in (tuple(inp['weights']), (1, 1, 1, 0, 0), (1, 1, 0, 0, 0), (0, 0, 1, 0, 0), (1, 0, 0, 0, 0), (0, 1, 0, 0, 0), (0, 0, 0, 1, 1), (0, 0, 0, 1, 0)):
    ctx.set_weights(kh.make_weights(*[float(v) for v in w]))
    for i in range(100): r = ctx.cycle((0.0, 0.0, 1e-3 * (i % 7 - 3), 0.0), P)
    t0 = time.perf_counter()
    for i in range(1000): r = ctx.cycle((0.0, 0.0, 1e-3 * (i % 7 - 3), 0.0), P)
    print(cfg, scene, w, "admissible", r.n_admissible, ": %.1f us" % ((time.perf_counter() - t0) / 1000 * 1e6),
          {k.replace("_kernel", ""): round(ms * 1e3, 1) for k, ms in ctx.timings() if not k.startswith("host:")})
