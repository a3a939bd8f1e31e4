"""Cycle kernel time against the clutter density of the scene (how many samples survive), cfg2-sized lattice, resident
inputs: a search for cliffs between the scenes the bench prices.  python tools/density_sweep.py [box|cyl] [cfg]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "kompass-core_amd")]
import numpy as np
import kompass_hip as kh, synthetic as syn

box = len(sys.argv) > 1 and sys.argv[1] == "box"
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
dens = [float(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else (0.02, 0.015, 0.012, 0.01, 0.008, 0.0065, 0.005, 0.0035, 0.002, 0.001, 0.0003)
opts = dict(kv.split("=") for kv in sys.argv[4:])
inp = syn.make_controller_inputs(cfg, seed=0, scene="survey")
rb = dict(shape=1, dims=[0.3, 0.2, 0.4]) if box else inp["robot"]
P, S = inp["P"], len(inp["seg_xyz"])
for p_occ in dens:
    pts = syn.costmap_points(syn.CONFIGS[cfg]["map_side"], 0.05, 0, p_occ=p_occ, free_radius=1.0)
    ctx = kh.DwaContext(rb["shape"], rb["dims"], (0, 0, 0), (0, 0, 0, 1), inp["octree_res"], inp["dt"], max_samples=len(inp["vx"]),
                        max_points=P, max_segment=S, max_obstacles=max(len(pts), 16), acc_limits=inp["acc_limits"])
    ctx.set_weights(kh.make_weights(*inp["weights"]))
    for k, v in opts.items():
        ctx.set_option(k, float(v))
    ctx.set_points(inp["state"], pts, inp["max_range"])
    ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
    ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
    pose = lambda i: (0.0, 0.0, 1e-3 * ((i % 7) - 3), 0.0)
    for i in range(100):
        r = ctx.cycle(pose(i), P)
    lat = []
    for i in range(300):
        t = time.perf_counter()
        r = ctx.cycle(pose(i), P)
        lat.append(time.perf_counter() - t)
    ctx.timing_enable(True)
    ks = {}
    for i in range(150):
        ctx.cycle(pose(i), P)
        for nm, ms in ctx.timings():
            if not nm.startswith("host:"):
                ks.setdefault(nm, []).append(ms)
    print("p_occ %.4f: %5d points, %4d admissible, cycle p50 %.1f us, kernels %s, single launch %s"
          % (p_occ, len(pts), r.n_admissible, np.percentile(lat, 50) * 1e6, {k: round(float(np.mean(v)) * 1e3, 1) for k, v in ks.items()},
             ctx.get_option("last_cycle_single_launch")), flush=True)
    ctx.close()
