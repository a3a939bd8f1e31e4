"""The CostEvaluator entry (kc_cost_upload + kc_cost_evaluate_resident: caller-provided trajectories with velocity
profiles, the reference's CostEvaluator_5k_Trajs shape) over batch sizes and trajectory lengths: a search for cliffs.
python tools/evaluator_sweep.py"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "kompass-core_amd")]
import numpy as np
import kompass_hip as kh, synthetic as syn
import bench

w = bench.ref_cost5k_inputs()
S = len(w["seg"])
for P in (50, 200, 1000):
    for N in (16, 128, 1000, 5001, 20000):
        horizon = P * 0.01
        px, py, vel = syn.ref_cost5k_samples(n_samples=N, horizon=horizon, dt=0.01)
        N_, P_ = px.shape
        ctx = kh.DwaContext(syn.CYLINDER, [0.1, 0.4], max_samples=N_, max_points=P_, max_segment=S, acc_limits=w["acc_limits"])
        ctx.set_weights(kh.make_weights(*w["weights"]))
        ctx.set_tracked_segment(w["seg"], w["acc"][w["s0"]:w["s0"] + S], w["total"])
        ctx.cost_upload(px, py, vel)
        for _ in range(20):
            ctx.cost_evaluate_resident(with_costs=False)
        ts = []
        for _ in range(100):
            t = time.perf_counter()
            ctx.cost_evaluate_resident(with_costs=False)
            ts.append(time.perf_counter() - t)
        ctx.timing_enable(True)
        ks = {}
        for _ in range(30):
            ctx.cost_evaluate_resident(with_costs=False)
            for nm, ms in ctx.timings():
                if not nm.startswith("host:"):
                    ks.setdefault(nm, []).append(ms)
        print("%5d trajectories x %4d points: %.1f us per evaluation (%.2f ns per point), kernels %s" %
              (N_, P_, np.percentile(ts, 50) * 1e6, np.percentile(ts, 50) * 1e9 / (N_ * P_), {k: round(float(np.mean(v)) * 1e3, 1) for k, v in ks.items()}), flush=True)
        ctx.close()
