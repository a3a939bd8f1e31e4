"""The reference's CostEvaluator_5k workload with chosen weights (path goal obstacles smoothness jerk):
kernel time per cost term.  python tools/cost5k_terms.py 1,0,0,0,0 [reps] [near_table] [velocity_group]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [os.path.join(ROOT, "kompass-core_amd"), ROOT]
import numpy as np
import kompass_hip as kh
import synthetic as syn
import bench

wts = [float(v) for v in (sys.argv[1] if len(sys.argv) > 1 else "1,1,0,1,1").split(",")]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
w = bench.ref_cost5k_inputs()
N, P = w["px"].shape
S = len(w["seg"])
ctx = kh.DwaContext(syn.CYLINDER, [0.1, 0.4], max_samples=N, max_points=P, max_segment=S, acc_limits=w["acc_limits"])
if len(sys.argv) > 3:
    ctx.set_option("near_table", int(sys.argv[3]))
if len(sys.argv) > 4:
    ctx.set_option("velocity_group", int(sys.argv[4]))
ctx.set_weights(kh.make_weights(*wts))
ctx.set_tracked_segment(w["seg"], w["acc"][w["s0"]:w["s0"] + S], w["total"])
ctx.cost_upload(w["px"], w["py"], w["vel"])
for _ in range(3):
    ctx.cost_evaluate_resident(with_costs=False)
t0 = time.perf_counter()
for _ in range(reps):
    ctx.cost_evaluate_resident(with_costs=False)
print("weights", wts, "near_table", ctx.get_option("near_table"), "velocity_group", ctx.get_option("velocity_group"), ": %.1f us per evaluation" % ((time.perf_counter() - t0) / reps * 1e6))
