"""Soak of the long-box pose gate (CollDev::cover): random boxes (aspect 2 .. 12, long axis along x or y), voxel
sizes, clouds, poses and lattices; one context looks the circles up, a second one takes the single look-up
(option box_cover = 0).  Any difference in the admissible set, the costs or the winner stops the run.
python tools/soak_boxes.py [iterations] [seed]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [os.path.join(ROOT, "kompass-core_amd"), ROOT]
import numpy as np
import kompass_hip as kh, synthetic as syn

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
covered = 0
for it in range(iters):
    inp = syn.make_controller_inputs("cfg2", seed=int(rng.integers(0, 50)), scale=float(rng.choice([0.2, 0.35])),
                                     scene=str(rng.choice(["survey", "mid"])))
    B = float(rng.uniform(0.04, 0.3))
    A = B * float(rng.uniform(2.0, 12.0))
    dims = [2 * A, 2 * B, 0.4] if rng.random() < 0.6 else [2 * B, 2 * A, 0.4]
    res = float(rng.choice([0.03, 0.05, 0.08]))
    pts = np.asarray(inp["points"], np.float32)
    pts = pts[rng.random(len(pts)) < rng.uniform(0.05, 1.0)].copy()
    if len(pts) == 0:
        continue
    st = (float(rng.uniform(-1, 1)), float(rng.uniform(-1, 1)), float(rng.uniform(-3.1, 3.1)), 0.0)
    out = []
    for cover in (1, 0):
        ctx = kh.DwaContext(syn.BOX, dims, (0, 0, 0), (0, 0, 0, 1), res, inp["dt"], max_samples=len(inp["vx"]), max_points=inp["P"],
                            max_segment=len(inp["seg_xyz"]), max_obstacles=len(pts), acc_limits=inp["acc_limits"])
        ctx.set_option("box_cover", cover)
        if it % 3 == 2:
            ctx.set_option("drop_samples", 0)
        ctx.set_weights(kh.make_weights(*inp["weights"]))
        ctx.set_points(st, pts, inp["max_range"])
        ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
        ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
        r = ctx.cycle(st, inp["P"])
        px, py, raw, costs = ctx.get_samples(with_costs=True)
        out.append((r.as_dict(), raw.copy(), costs.copy(), ctx.get_option("last_cycle_single_launch")))
        ctx.close()
    (a, ra, ca, sa), (b, rb, cb, sb) = out
    if a != b or not np.array_equal(ra, rb) or not np.array_equal(ca.view(np.uint32), cb.view(np.uint32)):
        print("MISMATCH at", it, dims, res, st, a, b, len(ra), len(rb), flush=True)
        sys.exit(1)
    covered += int((max(dims[:2]) - min(dims[:2])) / 2 / res >= 4.5)
    if it % 50 == 49:
        print("%d boxes, %d of them with circles, last: dims %s voxels %.2f admissible %d single launch %d/%d" %
              (it + 1, covered, np.round(dims, 2), res, a["n_admissible"], sa, sb), flush=True)
print("no difference in %d boxes (%d with circles)" % (iters, covered))
