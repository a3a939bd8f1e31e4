"""sensor_fused_kernel by role (measurement hook KC_SENSOR_ROLES, HIP events): python tools/sensor_roles.py"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "kompass-core_amd")]
import numpy as np
import kompass_hip as kh, synthetic as syn
inp = syn.make_controller_inputs("cfg2", seed=0)
P, S = inp["P"], len(inp["seg_xyz"])
ctx = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, 0), (0, 0, 0, 1), inp["octree_res"], inp["dt"],
                    max_samples=8704, max_points=P, max_segment=S, max_obstacles=len(inp["points"]), acc_limits=inp["acc_limits"])
ctx.set_weights(kh.make_weights(*inp["weights"]))
ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
pts = np.ascontiguousarray(inp["points"], np.float32)
ctx.set_points(inp["state"], pts, inp["max_range"]); ctx.cycle(inp["state"], P)
ctx.timing_enable(True)
ts = []
for i in range(300):
    ctx.set_points(inp["state"], pts, inp["max_range"])
    for name, ms in ctx.timings():
        if name == "sensor_fused_kernel":
            ts.append(ms * 1e3)
    if os.environ.get("KC_SENSOR_ROLES", "15") == "15":
        ctx.cycle(inp["state"], P)
        ctx.timings()
print("roles %s kb %s nb %s: sensor_fused_kernel %.2f us (events, median; min %.2f)" % (
    os.environ.get("KC_SENSOR_ROLES", "15"), os.environ.get("KC_SENSOR_KB", "-"), os.environ.get("KC_SENSOR_NB", "-"), np.median(ts), np.min(ts)))
