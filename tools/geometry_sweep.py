"""The fresh-input cycle (kc_dwa_find_best_path, cfg2-sized window, survey / mid clouds) over the robot: shape, size
(a 5 cm puck ... a 1.2 m platform), voxel size of the collision model (2 ... 25 cm) and where the sensor sits on the body.
python tools/geometry_sweep.py [scene] [boxes]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "kompass-core_amd")]
import numpy as np
import kompass_hip as kh, synthetic as syn

scene = sys.argv[1] if len(sys.argv) > 1 else "survey"
kw = {} if scene == "survey" else {"scene": scene}
inp = syn.make_controller_inputs("cfg2", seed=0, **kw)
base = syn.CONFIGS["cfg2"]
lim = kh.make_limits(syn.LIMITS["vx"], syn.LIMITS["vy"], syn.LIMITS["omega"])
P = inp["P"]
pts = np.ascontiguousarray(inp["points"], np.float32)
seg = np.asarray(inp["seg_xyz"], np.float32)
sacc = np.ascontiguousarray(inp["acc_at_seg"], np.float32)


def run(label, shape, dims, res, sensor=(0, 0, 0), L=91, A=91, opts={}):
    ctx = kh.DwaContext(shape, dims, sensor, (0, 0, 0, 1), res, inp["dt"], max_samples=(L + 2) * (A + 2), max_points=P,
                        max_segment=len(seg), max_obstacles=len(pts), acc_limits=inp["acc_limits"])
    ctx.set_weights(kh.make_weights(*inp["weights"]))
    for k, v in opts.items():
        ctx.set_option(k, v)
    lat, ks = [], {}
    for i in range(420):
        st = (0.0, 0.0, 1e-3 * ((i % 7) - 3), 0.0)
        if i == 300:
            ctx.timing_enable(True)
        t = time.perf_counter()
        r = ctx.find_best_path(st, P, window=(base["ctr"], lim, (0.5, 0.0, 0.0), L, A), points=pts,
                               max_sensor_range=inp["max_range"], segment=(seg, sacc, inp["ref_len"]))
        dt = time.perf_counter() - t
        if 100 <= i < 300:
            lat.append(dt)
        if i >= 300:
            for nm, ms in ctx.timings():
                ks.setdefault(nm, []).append(ms)
    print("%-44s %5d admissible, cycle p50 %6.1f p90 %6.1f us, single %d, %s" %
          (label, r.n_admissible, np.percentile(lat, 50) * 1e6, np.percentile(lat, 90) * 1e6,
           ctx.get_option("last_cycle_single_launch"),
           {k.replace("_kernel", ""): round(float(np.mean(v)) * 1e3, 1) for k, v in ks.items() if np.mean(v) * 1e3 >= 1.0}), flush=True)
    ctx.close()


if len(sys.argv) > 2 and sys.argv[2] == "boxes":   # long boxes: circles along the axis (option box_cover) on / off
    for d in ([1.5, 0.2, 0.5], [1.2, 0.4, 0.5], [1.0, 0.5, 0.5], [0.3, 1.2, 0.4]):
        for res in (0.05, 0.1):
            for cover in (1, 0):
                run("box %s, voxels %.2f, box_cover %d" % (d, res, cover), kh.BOX, d, res, opts={"box_cover": cover})
    sys.exit(0)
for res in (0.02, 0.05, 0.1, 0.25):
    run("cylinder r 0.1 h 0.4, voxels %.2f" % res, kh.CYLINDER, [0.1, 0.4], res)
for r_ in (0.05, 0.3, 0.6, 1.2):
    run("cylinder r %.2f h 0.4, voxels 0.1" % r_, kh.CYLINDER, [r_, 0.4], 0.1)
for d in ([0.1, 0.1, 0.1], [0.6, 0.4, 0.3], [1.5, 0.2, 0.5], [2.0, 2.0, 1.0]):
    run("box %s, voxels 0.1" % d, kh.BOX, d, 0.1)
for r_ in (0.05, 0.15, 0.5, 1.0):
    run("sphere r %.2f, voxels 0.1" % r_, kh.SPHERE, [r_], 0.1)
for res in (0.02, 0.05, 0.25):
    run("sphere r 0.15, voxels %.2f" % res, kh.SPHERE, [0.15], res)
for s in ((0.2, 0.0, 0.3), (-0.3, 0.1, -0.1), (0.0, 0.0, 1.0)):
    run("cylinder r 0.1 h 0.4, sensor at %s" % (s,), kh.CYLINDER, [0.1, 0.4], 0.1, sensor=s)
    run("sphere r 0.15, sensor at %s" % (s,), kh.SPHERE, [0.15], 0.1, sensor=s)
