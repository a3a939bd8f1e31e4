"""LaserScan input: cycle time against the size of the room (ranges 0.8 .. 9 m, round / square / corridor), 1440 beams,
cfg2-sized lattice, set_scan + cycle every step: a search for cliffs of the scan path.  python tools/room_sweep.py [beams]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "kompass-core_amd")]
import numpy as np
import kompass_hip as kh, synthetic as syn

beams = int(sys.argv[1]) if len(sys.argv) > 1 else 1440
only_shape = sys.argv[2] if len(sys.argv) > 2 else None
only_sizes = [float(v) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else None
opts = dict(kv.split("=") for kv in sys.argv[4:])
inp = syn.make_controller_inputs("cfg2", seed=0, scene="survey")
P, S = inp["P"], len(inp["seg_xyz"])
ang = np.linspace(-np.pi, np.pi, beams, endpoint=False)
rng = np.random.default_rng(0)
ctx = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, 0), (0, 0, 0, 1), inp["octree_res"], inp["dt"],
                    max_samples=len(inp["vx"]), max_points=P, max_segment=S, max_obstacles=max(beams, 16), acc_limits=inp["acc_limits"])
ctx.set_weights(kh.make_weights(*inp["weights"]))
ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
for k, v in opts.items():
    ctx.set_option(k, float(v))
for shape in ("round", "square", "corridor"):
    if only_shape and shape != only_shape:
        continue
    for size in (only_sizes or (0.8, 1.2, 1.8, 2.5, 3.5, 5.0, 7.0, 9.0)):
        if shape == "round":
            r0 = np.full(beams, size)
        elif shape == "square":
            r0 = size / np.maximum(np.abs(np.cos(ang)), np.abs(np.sin(ang)))
        else:  # walls at +- size / 3 beside the robot, open ahead and behind up to 9.5 m
            r0 = np.minimum((size / 3) / np.maximum(np.abs(np.sin(ang)), 1e-3), 9.5)
        lat, ks = [], {}
        for i in range(400):
            r = np.minimum(r0 + rng.uniform(0.0, 0.02, beams), 9.9)
            if os.environ.get("KC_TOOL_NO_RETURN"):   # every twentieth beam without a return
                r[::20] = np.inf
            st = (0.0, 0.0, 1e-3 * ((i % 7) - 3), 0.0)
            if i == 250:
                ctx.timing_enable(True)
            t = time.perf_counter()
            ctx.set_scan(st, r, ang, inp["max_range"])
            if i >= 250:
                for nm, ms in ctx.timings():
                    if not nm.startswith("host:"):
                        ks.setdefault(nm, []).append(ms)
            res = ctx.cycle(st, P)
            dt = time.perf_counter() - t
            if 100 <= i < 250:
                lat.append(dt)
            if i >= 250:
                for nm, ms in ctx.timings():
                    if not nm.startswith("host:"):
                        ks.setdefault(nm, []).append(ms)
        ctx.timing_enable(False)
        print("%-8s %.1f m: %4d admissible, set_scan + cycle p50 %.1f us, kernels %s" %
              (shape, size, res.n_admissible, np.percentile(lat, 50) * 1e6, {k: round(float(np.mean(v)) * 1e3, 1) for k, v in ks.items()}), flush=True)
ctx.close()
