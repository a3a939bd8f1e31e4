"""Resident-input cycle of cfg2 on the three scenes for ONE build of the library: us per cycle (p50) and the cycle
kernel by HIP events.  One line, for tools/ab_libs.sh:  tools/ab_libs.sh 3 "python tools/cycle_time.py" base new"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "kompass-core_amd")]
import numpy as np
import kompass_hip as kh, synthetic as syn

cfg = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
opts = dict(kv.split("=") for kv in sys.argv[2:])
out = []
for scene in ("survey", "mid", "open"):
    inp = syn.make_controller_inputs(cfg, seed=0, scene=scene)
    P, S = inp["P"], len(inp["seg_xyz"])
    ctx = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, 0), (0, 0, 0, 1), inp["octree_res"], inp["dt"],
                        max_samples=len(inp["vx"]), max_points=P, max_segment=S, max_obstacles=len(inp["points"]),
                        acc_limits=inp["acc_limits"])
    for k, v in opts.items():
        ctx.set_option(k, float(v))
    ctx.set_weights(kh.make_weights(*inp["weights"]))
    ctx.set_points(inp["state"], np.ascontiguousarray(inp["points"], np.float32), inp["max_range"])
    seg = np.asarray(inp["seg_xyz"], np.float32)   # (the entry every build has)
    ctx.set_tracked_segment_columns(np.ascontiguousarray(seg[:, 0]), np.ascontiguousarray(seg[:, 1]),
                                    np.ascontiguousarray(seg[:, 2]), inp["acc_at_seg"], inp["ref_len"])
    ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
    pose = lambda i: (0.0, 0.0, 1e-3 * ((i % 7) - 3), 0.0)
    for i in range(200):
        ctx.cycle(pose(i), P)
    lat = []
    for i in range(1500):
        t = time.perf_counter()
        ctx.cycle(pose(i), P)
        lat.append(time.perf_counter() - t)
    ctx.timing_enable(True)
    k = []
    for i in range(300):
        ctx.cycle(pose(i), P)
        k += [ms for name, ms in ctx.timings() if not name.startswith("host:")]
    ctx.timing_enable(False)
    out.append("%s %.1f (kernels %.1f)" % (scene, np.percentile(lat, 50) * 1e6, np.sum(k) / 300 * 1e3))
    ctx.close()
print(cfg, " | ".join(out))
