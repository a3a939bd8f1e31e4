"""Soak: many controller cycles (fresh sensor data and segment every few cycles,
seven poses), every result compared with the first one computed for the same
inputs.  Catches rare visibility / ordering slips that a parity test of a few
cycles would not.  python tools/soak.py [seconds] [scene]   (scene mid / open: many survivors, the
near table of the tracked segment is rebuilt with every segment update)"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kompass-core_amd"))
import numpy as np
import kompass_hip as kh, synthetic as syn

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
scene = sys.argv[2] if len(sys.argv) > 2 else "survey"
inp = syn.make_controller_inputs("cfg2", seed=0, scene=scene)
base = syn.CONFIGS["cfg2"]
vx, vy, om = syn.lattice_nonholonomic(base["n_vx"], base["n_om"])
P, S, O = inp["P"], len(inp["seg_xyz"]), len(inp["points"])
ctx = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, 0), (0, 0, 0, 1), inp["octree_res"], inp["dt"],
                    max_samples=len(vx), max_points=P, max_segment=S, max_obstacles=O, acc_limits=inp["acc_limits"], device=0)
ctx.set_weights(kh.make_weights(*inp["weights"]))
ctx.set_samples(vx, vy, om)
pts = [np.asarray(inp["points"], np.float32), np.asarray(inp["points"], np.float32)[::2].copy()]
m = kh.MapperContext(300, 300, 0.05, (0, 0, 0), 0.0, 1024)
ang, rng = syn.dense_scan(1024, 0.8)
segs = [np.asarray(inp["seg_xyz"], np.float32), np.asarray(inp["seg_xyz"], np.float32) + np.float32([0.0, 0.03, 0.0])]
first = {}
n = bad = 0
t_end = time.perf_counter() + secs
while time.perf_counter() < t_end:
    variant = (n // 5) % 3
    if n % 5 == 0:
        if variant < 2:
            ctx.set_points(inp["state"], pts[variant], inp["max_range"])
        else:
            m.scan_to_grid_device(ang, rng)
            ctx.set_grid_from_mapper(inp["state"], m, inp["max_range"])
        ctx.set_tracked_segment(segs[(n // 5) % 2], inp["acc_at_seg"], inp["ref_len"])
    lat = (n // 3) % 2   # a new window every three cycles: same trig rows, other speeds (lattice over the BAR)
    if n % 3 == 0:
        ctx.set_samples(vx * (1.0 - 0.05 * lat), vy, om)
    pose = (0.0, 0.0, 1e-3 * ((n % 7) - 3), 0.0)
    r = ctx.cycle(pose, P)
    key = (variant, (n // 5) % 2, lat, n % 7)
    row = ctx.get_best() if r.found else None   # single-launch cycle: the row that came with the pinned record
    got = (bool(r.found), int(r.raw_index), int(r.index), int(r.n_admissible), float(np.float32(r.cost)),
           hash(row[0].tobytes() + row[1].tobytes()) if row else 0)
    if key not in first:
        first[key] = got
    elif first[key] != got:
        bad += 1
        if bad < 10:
            print("MISMATCH", n, key, first[key], got, flush=True)
    n += 1
print(f"{n} cycles, {len(first)} distinct inputs, {bad} mismatches")
sys.exit(1 if bad else 0)
