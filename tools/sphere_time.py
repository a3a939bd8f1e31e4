import sys, time
sys.path.insert(0, "/root/repo/kompass-core_amd")
import numpy as np
import kompass_hip as kh, synthetic as syn
for shape, dims in ((syn.CYLINDER, [0.2, 0.4]), (syn.SPHERE, [0.2])):
    for mode in (1, 0):
        inp = syn.make_controller_inputs("cfg2", seed=0)
        P, S = inp["P"], len(inp["seg_xyz"])
        ctx = kh.DwaContext(shape, dims, (0, 0, 0), (0, 0, 0, 1), inp["octree_res"], inp["dt"], max_samples=len(inp["vx"]),
                            max_points=P, max_segment=S, max_obstacles=len(inp["points"]), acc_limits=inp["acc_limits"])
        ctx.set_option("fused_cycle", mode)
        ctx.set_weights(kh.make_weights(*inp["weights"]))
        ctx.set_points(inp["state"], inp["points"], inp["max_range"])
        ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
        ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
        lat = []
        for i in range(600):
            t = time.perf_counter(); r = ctx.cycle((0.0, 0.0, 1e-3 * (i % 7 - 3), 0.0), P)
            if i >= 100: lat.append(time.perf_counter() - t)
        print("shape", shape, "fused_cycle", mode, "single", ctx.get_option("last_cycle_single_launch"), "adm", r.n_admissible, "cycle %.1f us" % (np.mean(lat) * 1e6))
        ctx.close()
