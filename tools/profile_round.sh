#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh TAG [quick]
# The default bench line, then rocprofv3 evidence (tools/pmc_collect.py: kernel stats + separate counter
# passes) for every workload the bench JSON prices: cfg2 on the three scenes, the fresh-input reference cycle
# (sensor build pair + window + segment + cycle), cfg4 mapper, cfg3 / cfg5 mid (three-kernel cycle:
# sample_cost_kernel), the reference's cost workload (velocity_sums_kernel); phase clocks of the cycle kernel.
set -e
TAG=${1:-r04_a}
O=gpurun_out/$TAG
mkdir -p $O
python bench.py > $O/cfg2_bench.json 2> $O/cfg2_bench.err || { tail -5 $O/cfg2_bench.err; exit 1; }
tail -c 400 $O/cfg2_bench.json; echo
C="--only-headline --no-cpu --steps 200 --warmup 20"
for scene in survey mid open; do
  python3 tools/pmc_collect.py $O cfg2_$scene -- $C --scene $scene
done
python3 tools/pmc_collect.py $O cfg2_fresh -- --fresh --no-cpu --steps 200 --warmup 20
python3 tools/pmc_collect.py $O cfg2_scan -- --scan --no-cpu --steps 200 --warmup 20
if [ "$2" != "quick" ]; then
  python3 tools/pmc_collect.py $O cfg4_mapper -- --mapper --no-cpu --steps 200 --warmup 20
  python3 tools/pmc_collect.py $O cfg3_mid -- $C --config cfg3 --scene mid
  python3 tools/pmc_collect.py $O cfg5_mid -- $C --config cfg5 --scene mid
  python3 tools/pmc_collect.py $O cost5k -- --ref cost5k --no-cpu --steps 50 --warmup 5
fi
for scene in survey mid open; do
  python3 tools/stamps_json.py cfg2 $scene $O/cfg2_${scene}_phase_stamps.json || true
done
ls $O
