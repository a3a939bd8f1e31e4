"""Host-side breakdown of a fresh-input step with LaserScan input (set_scan + window + segment + cycle):
python tools/scan_fresh_breakdown.py [beams] [opt=value ...]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "kompass-core_amd")]
import numpy as np
import kompass_hip as kh, synthetic as syn

beams = int(sys.argv[1]) if len(sys.argv) > 1 else 1440
opts = dict(kv.split("=") for kv in sys.argv[2:])
cfg = "cfg2"
inp = syn.make_controller_inputs(cfg, seed=0, scene="open")
base = syn.CONFIGS[cfg]
P, S = inp["P"], len(inp["seg_xyz"])
ctx = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, 0), (0, 0, 0, 1), inp["octree_res"], inp["dt"],
                    max_samples=8704, max_points=P, max_segment=S, max_obstacles=max(beams, 16), acc_limits=inp["acc_limits"])
for k, v in opts.items():
    ctx.set_option(k, float(v))
ctx.set_weights(kh.make_weights(*inp["weights"]))
lim = kh.make_limits(syn.LIMITS["vx"], syn.LIMITS["vy"], syn.LIMITS["omega"])
seg = np.asarray(inp["seg_xyz"], np.float32)
sx, sy, sz = (np.ascontiguousarray(seg[:, k]) for k in range(3))
ang = np.linspace(-np.pi, np.pi, beams, endpoint=False)
rbase = 4.0 + 1.5 * np.cos(5 * ang)
names = ["sample_window", "set_scan", "set_tracked_segment", "cycle"]
acc = {k: [] for k in names + ["total"]}
for i in range(900):
    st = (0.001 * (i % 7), 0.0, 0.0, 0.5)
    rng = rbase + 0.01 * (i % 9)
    t0 = time.perf_counter()
    ctx.sample_window(base["ctr"], lim, (0.5, 0.0, 0.001 * (i % 5)), 91, 91, want_list=False)
    t1 = time.perf_counter()
    ctx.set_scan(st, rng, ang, 10.0)
    t2 = time.perf_counter()
    ctx.set_tracked_segment_columns(sx, sy, sz, inp["acc_at_seg"], inp["ref_len"])
    t3 = time.perf_counter()
    r = ctx.cycle(st, P)
    t4 = time.perf_counter()
    if i >= 200:
        for k, a, b in zip(names, (t0, t1, t2, t3), (t1, t2, t3, t4)):
            acc[k].append((b - a) * 1e6)
        acc["total"].append((t4 - t0) * 1e6)
ctx.timing_enable(True)
ctx.set_scan(st, rng, ang, 10.0)
k1 = {n: round(ms * 1e3, 1) for n, ms in ctx.timings() if not n.startswith("host:")}
ctx.cycle(st, P)
k2 = {n: round(ms * 1e3, 1) for n, ms in ctx.timings() if not n.startswith("host:")}
print(beams, "beams", opts, "admissible", r.n_admissible, "| us per call (median):",
      {k: round(float(np.median(acc[k])), 1) for k in names + ["total"]}, "| kernels", k1, k2)
ctx.close()
