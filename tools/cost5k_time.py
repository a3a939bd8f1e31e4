"""CostEvaluator_5k_Trajs (benchmark_runner.cpp:152-185) timing of one build: ms per resident evaluation,
overlapped (two streams) and with the kernels timed one by one.  One line for tools/ab_libs.sh."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "kompass-core_amd")]
import numpy as np
import bench
import kompass_hip as kh
import synthetic as syn

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--opt", action="append", default=[])
a = ap.parse_args()
w = bench.ref_cost5k_inputs()
N, P = w["px"].shape
S = len(w["seg"])
ctx = kh.DwaContext(syn.CYLINDER, [0.1, 0.4], max_samples=N, max_points=P, max_segment=S, acc_limits=w["acc_limits"])
ctx.set_weights(kh.make_weights(*w["weights"]))
ctx.set_tracked_segment(w["seg"], w["acc"][w["s0"]:w["s0"] + S], w["total"])
ctx.cost_upload(w["px"], w["py"], w["vel"])
for o in a.opt:
    k, v = o.split("=")
    ctx.set_option(k, int(v))
for _ in range(30):
    ctx.cost_evaluate_resident(with_costs=False)
lat = []
for _ in range(a.steps):
    t = time.perf_counter()
    r = ctx.cost_evaluate_resident(with_costs=False)
    lat.append(time.perf_counter() - t)
lat = np.array(lat) * 1e3
ctx.timing_enable(True)
kms = {}
for _ in range(30):
    ctx.cost_evaluate_resident(with_costs=False)
    for name, ms in ctx.timings():
        if not name.startswith("host:"):
            kms.setdefault(name, []).append(ms)
print("cost5k ms: mean %.4f p50 %.4f min %.4f max %.4f | alone: %s | winner %d %.9g" % (
    lat.mean(), np.percentile(lat, 50), lat.min(), lat.max(),
    " ".join("%s %.1f" % (k.replace("_kernel", ""), 1e3 * np.mean(v)) for k, v in kms.items()), r.index, r.cost))
