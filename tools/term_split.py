"""Which cost term costs what: cycle / kernel times of a config + scene with the weights switched.
python tools/term_split.py cfg5 mid"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "kompass-core_amd")]
import numpy as np
import kompass_hip as kh, synthetic as syn
cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
scene = sys.argv[2] if len(sys.argv) > 2 else "mid"
inp = syn.make_controller_inputs(cfg, seed=0, scene=scene)
P, S = inp["P"], len(inp["seg_xyz"])
for name, w in (("all", inp["weights"]), ("segment only", (1, 1, 0, 0, 0)), ("obstacles only", (0, 0, 1, 0, 0)), ("none", (0, 0, 0, 0, 0))):
    ctx = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, 0), (0, 0, 0, 1), inp["octree_res"], inp["dt"],
                        max_samples=len(inp["vx"]), max_points=P, max_segment=S, max_obstacles=len(inp["points"]),
                        acc_limits=inp["acc_limits"])
    ctx.set_weights(kh.make_weights(*w))
    ctx.set_points(inp["state"], inp["points"], inp["max_range"])
    ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
    ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
    pose = lambda i: (0.0, 0.0, 1e-3 * ((i % 7) - 3), 0.0)
    for i in range(50): r = ctx.cycle(pose(i), P)
    t0 = time.perf_counter()
    for i in range(300): r = ctx.cycle(pose(i), P)
    t = (time.perf_counter() - t0) / 300 * 1e6
    ctx.timing_enable(True)
    acc = {}
    for i in range(40):
        ctx.cycle(pose(i), P)
        for n, ms in ctx.timings():
            if not n.startswith("host:"):
                acc.setdefault(n, []).append(ms * 1e3)
    k = {n: round(float(np.median(v)), 1) for n, v in acc.items()}
    print(f"{cfg} {scene} {name:15s} {t:7.1f} us/cycle  admissible {r.n_admissible}  {k}", flush=True)
    ctx.close()
