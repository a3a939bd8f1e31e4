"""Soak of the device-side sensor update: random clouds (1 .. 60 k points, random extent / origin / z layers /
non-finite points, occasionally a tiny cloud or an empty one) on ONE context, each followed by a cycle; a second
context builds every update on the host (`sensor_on_host`).  Any difference in the admissible set, the costs or the
winner stops the run.  python tools/soak_sensor.py [iterations] [seed] [cylinder|box|sphere]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [os.path.join(ROOT, "kompass-core_amd"), os.path.join(ROOT, "tests"), ROOT]
import numpy as np
import kompass_hip as kh, synthetic as syn

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
inp = syn.make_controller_inputs("cfg2", seed=3, scale=0.25)
shape = sys.argv[3] if len(sys.argv) > 3 else "cylinder"
inp["robot"] = {"cylinder": inp["robot"], "box": dict(shape=syn.BOX, dims=[0.3, 0.2, 0.4]), "sphere": dict(shape=syn.SPHERE, dims=[0.22])}[shape]
sensor_z = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
P, S = inp["P"], len(inp["seg_xyz"])


def ctx(host):
    c = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, sensor_z), (0, 0, 0, 1), inp["octree_res"], inp["dt"],
                      max_samples=len(inp["vx"]), max_points=P, max_segment=S, max_obstacles=70000,
                      acc_limits=inp["acc_limits"])
    c.set_option("sensor_on_host", 1 if host else 0)
    c.set_weights(kh.make_weights(*inp["weights"]))
    c.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
    c.set_samples(inp["vx"], inp["vy"], inp["omega"])
    return c


a, b = ctx(False), ctx(True)
t0 = time.time()
sizes = {}
for it in range(iters):
    kind = rng.integers(0, 10)
    n = int({0: rng.integers(0, 4), 1: rng.integers(1, 300), 2: rng.integers(3000, 5000)}.get(int(kind), rng.integers(300, 60000)))
    ext = float(rng.choice([1.5, 4.0, 12.0, 30.0]))
    org = rng.uniform(-50, 50, 2) if rng.random() < 0.3 else np.zeros(2)
    pts = np.zeros((n, 3), np.float32)
    r = 0.7 + ext * np.sqrt(rng.random(n))
    th = rng.random(n) * 2 * np.pi
    pts[:, 0], pts[:, 1] = org[0] + r * np.cos(th), org[1] + r * np.sin(th)
    pts[:, 2] = rng.choice([-0.3, 0.0, 0.1, 0.5], n)
    if n > 10 and rng.random() < 0.2:
        pts[rng.integers(0, n, 3), rng.integers(0, 3, 3)] = rng.choice([np.nan, np.inf, -np.inf])
    st = (float(org[0] + rng.uniform(-0.2, 0.2)), float(org[1] + rng.uniform(-0.2, 0.2)), float(rng.uniform(-3, 3)), 0.0)
    ra = rb = None
    for c in (a, b):
        c.set_points(st, pts, 10.0)
        r_ = c.cycle(st, P)
        if c is a: ra = r_
        else: rb = r_
    key = lambda r: (r.found, r.index, r.raw_index, r.n_admissible, np.float32(r.cost).view(np.uint32) if r.found else 0)
    if key(ra) != key(rb):
        print("MISMATCH at", it, "n", n, "ext", ext, "origin", org, key(ra), key(rb))
        sys.exit(1)
    if it % 97 == 0:
        ca, cb = a.get_samples(with_costs=True), b.get_samples(with_costs=True)
        if not (np.array_equal(ca[2], cb[2]) and np.array_equal(ca[3].view(np.uint32), cb[3].view(np.uint32))):
            print("COST MISMATCH at", it, "n", n)
            sys.exit(1)
    sizes[min(n // 4096, 15)] = sizes.get(min(n // 4096, 15), 0) + 1
    if it % 500 == 499:
        print(f"{it + 1} updates + cycles, {time.time() - t0:.0f} s, 0 mismatches", flush=True)
print("done:", iters, "updates, 0 mismatches; sizes by 4096:", dict(sorted(sizes.items())))
