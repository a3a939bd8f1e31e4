"""Soak of device trig: two contexts -- the kernels' own cos / sin rows (+ the table riding in the
sensor launch) against the host's libm table -- driven with the same random inputs: random poses (yaw over
[-pi, pi] and, every so often, far outside), a new random velocity window every few cycles, fresh sensor data
every second cycle (so that both the riding table and the in-kernel rows are exercised), both robot footprints.
Any difference in the admissible count, the winner, its cost bits or (every 500 cycles) any per-sample cost stops
the run.   python tools/soak_trig.py [seconds] [cfg] [seed]"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [os.path.join(ROOT, "kompass-core_amd"), ROOT]
import numpy as np
import kompass_hip as kh, synthetic as syn

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 30.0
cfg = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 1)
t_end = time.time() + secs
total = rides = 0
for shape in ("cylinder", "box"):
    inp = syn.make_controller_inputs(cfg, seed=0, scene="mid")
    if shape == "box":
        inp = dict(inp, robot=dict(shape=syn.BOX, dims=[0.5, 0.34, 0.3]))
    P, S = inp["P"], len(inp["seg_xyz"])
    ctxs = []
    for dev in (1, 0):
        c = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, 0), (0, 0, 0, 1), inp["octree_res"], inp["dt"],
                          max_samples=len(inp["vx"]), max_points=P, max_segment=S, max_obstacles=len(inp["points"]),
                          acc_limits=inp["acc_limits"])
        c.set_option("device_trig", dev)
        c.set_weights(kh.make_weights(*inp["weights"]))
        c.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
        c.set_samples(inp["vx"], inp["vy"], inp["omega"])
        c.set_points(inp["state"], inp["points"], inp["max_range"])
        ctxs.append(c)
    half = t_end - (t_end - time.time()) / (2 if shape == "cylinder" else 1)
    i = 0
    while time.time() < half:
        yaw = rng.uniform(-np.pi, np.pi) if i % 17 else rng.uniform(-1.0, 1.0) * 10.0 ** rng.integers(0, 7)
        st = (float(rng.uniform(-0.3, 0.3)), float(rng.uniform(-0.3, 0.3)), float(yaw), 0.0)
        if i % 5 == 0:   # a new window: other velocities, other omegas (same pattern or not)
            k = rng.uniform(0.3, 1.5)
            for c in ctxs:
                c.set_samples(inp["vx"] * k, inp["vy"] * k, inp["omega"] * rng.uniform(0.2, 2.0) if False else inp["omega"] * k)
        if i % 2 == 0:
            for c in ctxs:
                c.set_points(st, inp["points"], inp["max_range"])
        ra, rb = (c.cycle(st, P) for c in ctxs)
        a = (ra.found, ra.index, ra.raw_index, ra.n_admissible, np.float32(ra.cost).view(np.uint32))
        b = (rb.found, rb.index, rb.raw_index, rb.n_admissible, np.float32(rb.cost).view(np.uint32))
        if a != b:
            print("MISMATCH", shape, i, st, a, b)
            sys.exit(1)
        if i % 500 == 7:
            ca, cb = (c.get_samples(with_costs=True) for c in ctxs)
            if not (np.array_equal(ca[2], cb[2]) and np.array_equal(ca[3].view(np.uint32), cb[3].view(np.uint32))
                    and np.array_equal(ca[0].view(np.uint32), cb[0].view(np.uint32))):
                print("MISMATCH in samples", shape, i, st)
                sys.exit(1)
        i += 1
    total += i
    rides += int(ctxs[0].get_option("trig_rides"))
    for c in ctxs:
        c.close()
print(f"{cfg}: {total} cycle pairs, 0 mismatches; {rides} tables rode in a sensor launch")
