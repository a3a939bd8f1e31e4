"""A room-like laserscan scene through the C ABI: set_scan and cycle times, kernel times.
python tools/scan_scene_time.py [beams]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kompass-core_amd"))
import numpy as np
import kompass_hip as kh, synthetic as syn
beams = int(sys.argv[1]) if len(sys.argv) > 1 else 1440
inp = syn.make_controller_inputs("cfg2", seed=0, scene="open")
P, S = inp["P"], len(inp["seg_xyz"])
ctx = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, 0), (0, 0, 0, 1), inp["octree_res"], inp["dt"],
                    max_samples=len(inp["vx"]), max_points=P, max_segment=S, max_obstacles=max(beams, 16),
                    acc_limits=inp["acc_limits"])
wts = inp["weights"]
for k, v in [a.split("=") for a in sys.argv[2:]]:
    if k == "weights":
        wts = tuple(float(t) for t in v.split(","))
    else:
        ctx.set_option(k, float(v))
ctx.set_weights(kh.make_weights(*wts))
ang = np.linspace(-np.pi, np.pi, beams, endpoint=False)
rng = 4.0 + 1.5 * np.cos(5 * ang)
st = (0.0, 0.0, 0.0, 0.0)
ctx.set_scan(st, rng, ang, 10.0)
ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
for i in range(30): r = ctx.cycle((0.0, 0.0, 1e-3 * (i % 7 - 3), 0.0), P)
t0 = time.perf_counter()
for i in range(300): r = ctx.cycle((0.0, 0.0, 1e-3 * (i % 7 - 3), 0.0), P)
tc = (time.perf_counter() - t0) / 300
t0 = time.perf_counter()
for i in range(300): ctx.set_scan(st, rng + 0.001 * (i % 5), ang, 10.0)
ts = (time.perf_counter() - t0) / 300
ctx.timing_enable(True)
ctx.cycle(st, P)
print(f"{beams} beams: cycle {tc * 1e6:.1f} us (admissible {r.n_admissible}, single launch {ctx.get_option('last_cycle_single_launch')}), set_scan {ts * 1e6:.1f} us;",
      {n: round(ms * 1e3, 1) for n, ms in ctx.timings() if not n.startswith('host:')})
