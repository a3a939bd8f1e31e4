#!/bin/bash
# usage (here, after gpurun merged gpurun_out/TAG back): tools/collect_profiles.sh TAG
# Summarises the counter CSVs of tools/profile_round2.sh and copies the files that are judged into profiles/.
set -e
TAG=${1:-r02_i}
O=gpurun_out/$TAG
for scene in survey open; do
  python tools/pmc_summary.py $O/cfg2_${scene}_pmc_FETCH_SIZE.csv $O/cfg2_${scene}_pmc_WRITE_SIZE.csv profiles/${TAG}_cfg2_${scene}_pmc_hbm.json > /dev/null
done
python tools/pmc_sq_summary.py $O profiles/${TAG}_cfg2
for f in $O/*_bench.json $O/*_kernel_stats.csv $O/class_cycle.txt; do
  [ -s "$f" ] && cp "$f" profiles/${TAG}_$(basename $f)
done
ls profiles/${TAG}_*
