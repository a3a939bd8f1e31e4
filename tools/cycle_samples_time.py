import sys, time
sys.path.insert(0, "/root/repo/kompass-core_amd")
import numpy as np
import kompass_hip as kh, synthetic as syn
for scene in ("survey", "mid", "open"):
    for cs in (32, 16):
        inp = syn.make_controller_inputs("cfg2", seed=0, scene=scene)
        P, S = inp["P"], len(inp["seg_xyz"])
        ctx = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, 0), (0, 0, 0, 1), inp["octree_res"], inp["dt"], max_samples=len(inp["vx"]),
                            max_points=P, max_segment=S, max_obstacles=len(inp["points"]), acc_limits=inp["acc_limits"])
        ctx.set_option("fused_cycle", 2); ctx.set_option("cycle_samples", cs)
        ctx.set_weights(kh.make_weights(*inp["weights"]))
        ctx.set_points(inp["state"], inp["points"], inp["max_range"])
        ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
        ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
        lat = []
        for i in range(1100):
            t = time.perf_counter(); r = ctx.cycle((0.0, 0.0, 1e-3 * (i % 7 - 3), 0.0), P)
            if i >= 100: lat.append(time.perf_counter() - t)
        print(scene, "cycle_samples", cs, "adm", r.n_admissible, "cycle %.1f us" % (np.mean(lat) * 1e6))
        ctx.close()
