#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh TAG
# bench (with cpu_baseline), rocprofv3 kernel stats of the same command, PMC HBM passes -> gpurun_out/TAG_*
set -e
TAG=${1:-r01_b}
export TMPDIR=/tmp
O=gpurun_out
python bench.py > $O/${TAG}_cfg2_bench.json 2> $O/${TAG}_cfg2_bench.err
tail -c 600 $O/${TAG}_cfg2_bench.json; echo
python bench.py --mapper --no-cpu > $O/${TAG}_cfg4_mapper_bench.json 2> $O/${TAG}_cfg4_mapper_bench.err || true
rocprofv3 --kernel-trace --stats --output-format csv -d $O/${TAG}_stats -o ${TAG} -- python3 bench.py --no-cpu > $O/${TAG}_stats.log 2>&1
find $O/${TAG}_stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_cfg2_kernel_stats.csv
head -8 $O/${TAG}_cfg2_kernel_stats.csv
for ctr in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/${TAG}_pmc_$ctr -o ${TAG}_$ctr -- python3 bench.py --steps 50 --warmup 5 --no-cpu > $O/${TAG}_pmc_$ctr.log 2>&1 || echo "pmc $ctr failed"
  find $O/${TAG}_pmc_$ctr -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/${TAG}_cfg2_pmc_$ctr.csv || true
done
ls -la $O | grep ${TAG}
