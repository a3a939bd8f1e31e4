"""Dense 3-D clouds: kc_dwa_set_points + kc_dwa_cycle against the size of the cloud (10 k .. 500 k points over a
20 m x 20 m room with furniture-like clusters, z in -0.2 .. 1.8 m), cfg2 lattice; the one-launch sensor build serves up
to 32 k points, the two-launch build beyond.  python tools/big_cloud_sweep.py [shape] [option=value ...]   (KC_TOOL_SIZES=10000,66000: those sizes only)"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "kompass-core_amd")]
import numpy as np
import kompass_hip as kh, synthetic as syn

shape = sys.argv[1] if len(sys.argv) > 1 else "cylinder"
opts = dict(kv.split("=") for kv in sys.argv[2:])
inp = syn.make_controller_inputs("cfg2", seed=0)
robot = {"cylinder": inp["robot"], "box": dict(shape=syn.BOX, dims=[0.6, 0.4, 0.5]), "sphere": dict(shape=syn.SPHERE, dims=[0.3])}[shape]
P, S = inp["P"], len(inp["seg_xyz"])
rng = np.random.default_rng(0)
sizes = (10_000, 30_000, 33_000, 66_000, 130_000, 260_000, 500_000) if "KC_TOOL_SIZES" not in os.environ else tuple(int(v) for v in os.environ["KC_TOOL_SIZES"].split(","))
for n in sizes:
    # walls + clusters, nothing within 1 m of the robot
    k = n // 2
    wall = np.stack([rng.uniform(-10, 10, k), rng.choice([-10.0, 10.0], k) + rng.normal(0, 0.02, k), rng.uniform(-0.2, 1.8, k)], 1)
    wall[::2, [0, 1]] = wall[::2][:, [1, 0]]
    centres = rng.uniform(-8, 8, (40, 2))
    centres = centres[np.hypot(centres[:, 0], centres[:, 1]) > 2.0]
    c = centres[rng.integers(0, len(centres), n - k)]
    clus = np.concatenate([c + rng.normal(0, 0.25, (n - k, 2)), rng.uniform(-0.2, 1.8, (n - k, 1))], 1)
    pts = np.ascontiguousarray(np.concatenate([wall, clus]), np.float32)
    ctx = kh.DwaContext(robot["shape"], robot["dims"], (0, 0, 0.3), (0, 0, 0, 1), inp["octree_res"], inp["dt"], max_samples=len(inp["vx"]),
                        max_points=P, max_segment=S, max_obstacles=len(pts), acc_limits=inp["acc_limits"])
    for k_, v_ in opts.items():
        ctx.set_option(k_, float(v_))
    ctx.set_weights(kh.make_weights(*inp["weights"]))
    ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
    ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
    tp, tc, ks = [], [], {}
    for i in range(130):
        st = (0.0, 0.0, 1e-3 * (i % 7), 0.0)
        if i == 100:
            ctx.timing_enable(True)
        t0 = time.perf_counter()
        ctx.set_points(st, pts, inp["max_range"])
        if i >= 100:
            for nm, ms in ctx.timings():
                if not nm.startswith("host:"):
                    ks.setdefault(nm, []).append(ms)
        t1 = time.perf_counter()
        r = ctx.cycle(st, P)
        t2 = time.perf_counter()
        if 30 <= i < 100:
            tp.append((t1 - t0) * 1e6)
            tc.append((t2 - t1) * 1e6)
        if i >= 100:
            for nm, ms in ctx.timings():
                if not nm.startswith("host:"):
                    ks.setdefault(nm, []).append(ms)
    print("%7d points: set_points p50 %7.1f us, cycle p50 %6.1f us, %5d admissible, sensor on host %d | kernels %s" %
          (n, np.percentile(tp, 50), np.percentile(tc, 50), r.n_admissible, ctx.get_option("sensor_on_host"),
           {k.replace("_kernel", ""): round(float(np.mean(v)) * 1e3, 1) for k, v in ks.items()}), flush=True)
    ctx.close()
