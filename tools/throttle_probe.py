"""Does the steady controller loop get CPU-throttled by the cgroup quota?
Runs cycles for a few seconds per KC_HOST_THREADS setting (child processes) and
reports throttling (cgroup v2 cpu.stat) and the latency tail."""
import os, subprocess, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
CHILD = r'''
import os, sys, time
sys.path.insert(0, os.path.join(%r, "kompass-core_amd"))
import numpy as np
import kompass_hip as kh, synthetic as syn
def stat():
    out = {}
    try:
        for line in open("/sys/fs/cgroup/cpu.stat"):
            k, v = line.split(); out[k] = int(v)
    except OSError:
        pass
    return out
inp = syn.make_controller_inputs("cfg2", seed=0)
base = syn.CONFIGS["cfg2"]
vx, vy, om = syn.lattice_nonholonomic(base["n_vx"], base["n_om"])
P, S, O = inp["P"], len(inp["seg_xyz"]), len(inp["points"])
ctx = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, 0), (0, 0, 0, 1), inp["octree_res"], inp["dt"],
                    max_samples=len(vx), max_points=P, max_segment=S, max_obstacles=O, acc_limits=inp["acc_limits"], device=0)
ctx.set_weights(kh.make_weights(*inp["weights"])); ctx.set_points(inp["state"], inp["points"], inp["max_range"])
ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"]); ctx.set_samples(vx, vy, om)
for i in range(200): ctx.cycle((0.0, 0.0, 1e-3 * (i %% 7), 0.0), P)
s0 = stat(); lat = []
t_end = time.perf_counter() + 3.0
i = 0
while time.perf_counter() < t_end:
    t0 = time.perf_counter(); ctx.cycle((0.0, 0.0, 1e-3 * (i %% 7), 0.0), P); lat.append(time.perf_counter() - t0); i += 1
s1 = stat(); lat = np.array(lat) * 1e6
print("threads=%%s cycles=%%d mean %%.1f p50 %%.1f p99 %%.1f max %%.0f us | throttled +%%d periods, +%%.1f ms" %% (
    os.environ.get("KC_HOST_THREADS", "default"), len(lat), lat.mean(), np.percentile(lat, 50), np.percentile(lat, 99), lat.max(),
    s1.get("nr_throttled", 0) - s0.get("nr_throttled", 0), (s1.get("throttled_usec", 0) - s0.get("throttled_usec", 0)) / 1e3))
''' % ROOT
for th in (None, "12", "8", "6", "4"):
    env = dict(os.environ)
    if th:
        env["KC_HOST_THREADS"] = th
    subprocess.run([sys.executable, "-c", CHILD], env=env)
