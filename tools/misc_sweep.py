"""More searches for cliffs of the fresh-input cycle (kc_dwa_find_best_path, cfg2-sized window): tracked segments of
different lengths / shapes / heights, poses far from the map origin and at every yaw, an omni window.
python tools/misc_sweep.py"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "kompass-core_amd")]
import numpy as np
import kompass_hip as kh, synthetic as syn

inp = syn.make_controller_inputs("cfg2", seed=0, scene="mid")
base = syn.CONFIGS["cfg2"]
lim = kh.make_limits(syn.LIMITS["vx"], syn.LIMITS["vy"], syn.LIMITS["omega"])
P = inp["P"]
pts0 = np.ascontiguousarray(inp["points"], np.float32)


def run(label, seg, sacc, ref_len, pose, pts, ctr=base["ctr"], L=91, A=91, cur=(0.5, 0.0, 0.0)):
    ctx = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, 0), (0, 0, 0, 1), inp["octree_res"], inp["dt"],
                        max_samples=(L + 2) * (A + 2) * (3 if ctr == syn.OMNI else 1), max_points=P, max_segment=len(seg), max_obstacles=len(pts),
                        acc_limits=inp["acc_limits"])
    ctx.set_weights(kh.make_weights(*inp["weights"]))
    lat, ks = [], {}
    for i in range(450):
        st = (pose[0], pose[1], pose[2] + 1e-3 * ((i % 7) - 3), 0.0)
        if i == 300:
            ctx.timing_enable(True)
        t = time.perf_counter()
        r = ctx.find_best_path(st, P, window=(ctr, lim, cur, L, A), points=pts, max_sensor_range=inp["max_range"], segment=(seg, sacc, ref_len))
        dt = time.perf_counter() - t
        if 100 <= i < 300:
            lat.append(dt)
        if i >= 300:
            for nm, ms in ctx.timings():
                if not nm.startswith("host:"):
                    ks.setdefault(nm, []).append(ms)
    print("%-34s %5d samples %5d admissible, cycle p50 %.1f us, single %d, kernels %s" %
          (label, r.n_samples, r.n_admissible, np.percentile(lat, 50) * 1e6, ctx.get_option("last_cycle_single_launch"),
           {k: round(float(np.mean(v)) * 1e3, 1) for k, v in ks.items()}), flush=True)
    ctx.close()


seg0 = np.asarray(inp["seg_xyz"], np.float32)
acc0 = np.ascontiguousarray(inp["acc_at_seg"], np.float32)
for S in (2, 11, 51, 201, 501):
    run("straight segment of %d points" % S, seg0[:S].copy(), acc0[:S].copy(), inp["ref_len"], (0, 0, 0), pts0)
arc, aacc = syn.arc_segment(501)
arc = np.asarray(arc, np.float32)
run("arc segment, 501 points", arc, np.ascontiguousarray(aacc, np.float32), 47.12389, (0, 0, 0), pts0)
hill = seg0.copy()
hill[:, 2] = 0.3 * np.sin(np.linspace(0, 6, len(hill)))
run("segment with varying z", hill, acc0, inp["ref_len"], (0, 0, 0), pts0)
for off in ((0, 0), (100, -50), (3000, 2000)):
    for yaw in (0.0, 1.0, 2.5, -3.0):
        c, s = np.cos(yaw), np.sin(yaw)
        R = np.array([[c, -s, 0], [s, c, 0], [0, 0, 1]], np.float32)
        shift = np.float32([off[0], off[1], 0])
        run("pose (%g, %g) yaw %.1f" % (off[0], off[1], yaw), (seg0 @ R.T + shift).astype(np.float32), acc0, inp["ref_len"],
            (off[0], off[1], yaw), np.ascontiguousarray(pts0 @ R.T + shift, np.float32))
run("omni window 31 x 31 x 9", seg0, acc0, inp["ref_len"], (0, 0, 0), pts0, ctr=syn.OMNI, L=31, A=9, cur=(0.3, 0.1, 0.0))
