"""Same-box A/B of a per-context option on the three cfg2 scenes (resident inputs, kc_dwa_cycle):
python tools/opt_ab.py cycle_block=512 [cfg]   -> us per cycle (mean of 3 alternating rounds) without / with"""
import os, sys, time
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path[:0] = [ROOT, os.path.join(ROOT, "kompass-core_amd")]
import numpy as np
import kompass_hip as kh, synthetic as syn

opts = dict(kv.split("=") for kv in sys.argv[1:] if "=" in kv)
cfg = next((a for a in sys.argv[1:] if a.startswith("cfg")), "cfg2")
for scene in ("survey", "mid", "open"):
    inp = syn.make_controller_inputs(cfg, seed=0, scene=scene)
    P, S = inp["P"], len(inp["seg_xyz"])
    res = {}
    ctxs = {}
    for tag in ("base", "opt"):
        ctx = kh.DwaContext(inp["robot"]["shape"], inp["robot"]["dims"], (0, 0, 0), (0, 0, 0, 1), inp["octree_res"], inp["dt"],
                            max_samples=len(inp["vx"]), max_points=P, max_segment=S, max_obstacles=len(inp["points"]),
                            acc_limits=inp["acc_limits"])
        if tag == "opt":
            for k, v in opts.items():
                ctx.set_option(k, float(v))
        ctx.set_weights(kh.make_weights(*inp["weights"]))
        ctx.set_points(inp["state"], inp["points"], inp["max_range"])
        ctx.set_tracked_segment(inp["seg_xyz"], inp["acc_at_seg"], inp["ref_len"])
        ctx.set_samples(inp["vx"], inp["vy"], inp["omega"])
        ctxs[tag] = ctx
        res[tag] = []
    pose = lambda i: (0.0, 0.0, 1e-3 * ((i % 7) - 3), 0.0)
    last = {}
    for rnd in range(4):
        for tag in ("base", "opt"):
            ctx = ctxs[tag]
            for i in range(100):
                ctx.cycle(pose(i), P)
            t0 = time.perf_counter()
            for i in range(1000):
                r = ctx.cycle(pose(i), P)
            if rnd:
                res[tag].append((time.perf_counter() - t0) / 1000 * 1e6)
            last[tag] = (r.found, r.raw_index, r.cost, r.n_admissible, ctx.get_option("last_cycle_single_launch"), ctx.get_option("last_cycle_samples"))
    same = last["base"][:4] == last["opt"][:4]
    print(f"{cfg} {scene:7s} base {np.mean(res['base']):6.1f} us   {opts} {np.mean(res['opt']):6.1f} us   same result {same}  "
          f"(single launch {last['base'][4]:.0f}/{last['opt'][4]:.0f}, samples per wg {last['base'][5]:.0f}/{last['opt'][5]:.0f})", flush=True)
    for c in ctxs.values():
        c.close()
