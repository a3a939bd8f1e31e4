#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_extra.sh TAG -- the bench lines beside the
# headline set of tools/profile_round.sh: other configs and scenes, --split, the reference's workloads,
# the multi-GPU code path with a world of one, mapper / Bayes / point cloud modes, class-level cycle.
set -e
TAG=${1:-r04_a}
O=gpurun_out/$TAG
mkdir -p $O
for c in cfg1 cfg3 cfg5; do python bench.py --config $c > $O/${c}_bench.json 2>> $O/extra.err; done
for c in cfg3 cfg5; do for s in mid open; do
  python bench.py --config $c --scene $s --only-headline --no-cpu --steps 300 --warmup 30 > $O/${c}_${s}_bench.json 2>> $O/extra.err
done; done
python bench.py --split --only-headline > $O/cfg2_split_bench.json 2>> $O/extra.err
KC_BENCH_FORCE_DIST=1 python bench.py --only-headline --no-cpu > $O/cfg2_world1_rccl_bench.json 2>> $O/extra.err
python bench.py --ref cost5k > $O/cost5k_bench.json 2>> $O/extra.err
python bench.py --ref mapper400 > $O/mapper400_bench.json 2>> $O/extra.err
python bench.py --mapper > $O/cfg4_mapper_bench.json 2>> $O/extra.err
python bench.py --mapper --bayes > $O/bayes_mapper_bench.json 2>> $O/extra.err
python bench.py --pointcloud > $O/pointcloud_bench.json 2>> $O/extra.err
python tools/class_cycle.py > $O/class_cycle.txt 2>> $O/extra.err
python tools/class_cycle_scan.py > $O/class_cycle_scan.txt 2>> $O/extra.err
python - "$O" <<'PY'
import glob, json, sys, os
for f in sorted(glob.glob(os.path.join(sys.argv[1], "*_bench.json"))):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(os.path.basename(f), d.get("ms_per_step"), {k: round(v * 1e3, 1) for k, v in d.get("kernels_ms", {}).items()})
    except Exception as e:
        print(os.path.basename(f), "unreadable", e)
PY
# The reference's own published workloads against the newest COMMITTED profile of each (a round must not
# silently lose time on them): more than 10 % slower fails this script loudly.
python - "$O" <<'PY'
import glob, json, os, sys
out, bad = sys.argv[1], []
for name in ("cost5k", "mapper400"):
    new = json.loads(open(os.path.join(out, f"{name}_bench.json")).read().strip().splitlines()[-1])["ms_per_step"]
    old_files = sorted(f for f in glob.glob(f"profiles/*_{name}_bench.json"))
    if not old_files:
        print(f"[regression check] {name}: {new:.4f} ms, no committed profile to compare with")
        continue
    old = json.loads(open(old_files[-1]).read().strip().splitlines()[-1])["ms_per_step"]
    verdict = "REGRESSION" if new > 1.10 * old else "ok"
    print(f"[regression check] {name}: {new:.4f} ms against {old:.4f} ms in {os.path.basename(old_files[-1])}: {verdict}")
    if verdict != "ok":
        bad.append(name)
if bad:
    sys.exit(f"REFERENCE WORKLOAD SLOWER THAN THE COMMITTED PROFILE BY MORE THAN 10 %: {', '.join(bad)}")
PY
