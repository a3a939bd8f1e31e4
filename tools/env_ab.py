"""Same-process A/B of context-creation environment switches: python tools/env_ab.py cfg3 mid KC_TRIG_STAGES=0 KC_TRIG_STAGES=1 ...
Each variant gets its own context (the variables are read at creation); blocks of cycles alternate between the
contexts, the median over the blocks of each variant's mean cycle time is printed."""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "kompass-core_amd")]
import numpy as np
import kompass_hip as kh, synthetic as syn

cfg, scene = sys.argv[1], sys.argv[2]
variants = sys.argv[3:]
inp = syn.make_controller_inputs(cfg, seed=0, scene=scene)
P, S = inp["P"], len(inp["seg_xyz"])
ctxs = []
for v in variants:
    kv = dict(x.split("=") for x in v.split(",") if x)
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update(kv)
    ctx = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, 0), (0, 0, 0, 1), inp["octree_res"], inp["dt"],
                        max_samples=len(inp["vx"]), max_points=P, max_segment=S, max_obstacles=len(inp["points"]),
                        acc_limits=inp["acc_limits"])
    for k, o in old.items():
        if o is None:
            del os.environ[k]
        else:
            os.environ[k] = o
    ctx.set_weights(kh.make_weights(*inp["weights"]))
    ctx.set_points(inp["state"], inp["points"], inp["max_range"])
    ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
    ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
    ctxs.append(ctx)
pose = lambda i: (0.0, 0.0, 1e-3 * ((i % 7) - 3), 0.0)
for ctx in ctxs:
    for i in range(60):
        ctx.cycle(pose(i), P)
blocks = {v: [] for v in variants}
for rep in range(9):
    for v, ctx in zip(variants, ctxs):
        t0 = time.perf_counter()
        for i in range(150):
            r = ctx.cycle(pose(i), P)
        blocks[v].append((time.perf_counter() - t0) / 150 * 1e6)
for v in variants:
    b = np.array(blocks[v])
    print(f"{cfg} {scene} {v:40s} median {np.median(b):7.1f} us/cycle  (min {b.min():.1f}, max {b.max():.1f})  admissible {r.n_admissible}")
