"""Diagnostic: cost of the per-sensor-update / per-path-update calls at cfg2 (wall time of the
call plus a device sync), next to one controller cycle."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "kompass-core_amd"))
import kompass_hip as kh, synthetic as syn

inp = syn.make_controller_inputs("cfg2", seed=0)
base = syn.CONFIGS["cfg2"]
vx, vy, om = syn.lattice_nonholonomic(base["n_vx"], base["n_om"])
P, S, O = inp["P"], len(inp["seg_xyz"]), len(inp["points"])
ctx = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, 0), (0, 0, 0, 1),
                    inp["octree_res"], inp["dt"], max_samples=len(vx), max_points=P,
                    max_segment=S, max_obstacles=O, acc_limits=inp["acc_limits"], device=0)
ctx.set_weights(kh.make_weights(*inp["weights"]))
ctx.set_samples(vx, vy, om)

def timed(fn, n=50):
    ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); ts.append(time.perf_counter() - t0)
    ts.sort(); return ts[len(ts) // 2] * 1e6

t_pts = timed(lambda: ctx.set_points(inp["state"], inp["points"], inp["max_range"]))
t_seg = timed(lambda: ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"]))
t_cyc = timed(lambda: ctx.cycle((0.0, 0.0, 0.001, 0.0), P), 200)
def full():
    ctx.set_points(inp["state"], inp["points"], inp["max_range"])
    ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
    ctx.cycle((0.0, 0.0, 0.001, 0.0), P)
t_full = timed(full, 100)
print(f"set_points({O} pts) {t_pts:.1f} us | set_tracked_segment({S}) {t_seg:.1f} us | cycle {t_cyc:.1f} us | all three {t_full:.1f} us")
