#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round2.sh TAG
# Full bench line; then, per scene, rocprofv3 kernel stats and SEPARATE counter passes (never combined
# with a trace domain other than --kernel-trace) of `bench.py --only-headline --scene S`.
# Everything lands in gpurun_out/TAG/; tools/pmc_summary.py + tools/pmc_sq_summary.py turn the CSVs into the
# small JSON files that are committed under profiles/.
set -e
TAG=${1:-r02_a}
export TMPDIR=/tmp
O=gpurun_out/$TAG
mkdir -p $O
python bench.py > $O/cfg2_bench.json 2> $O/cfg2_bench.err
tail -c 300 $O/cfg2_bench.json; echo
B="bench.py --only-headline --no-cpu --steps 200 --warmup 20"
for scene in survey mid open; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$scene -o s -- python3 $B --scene $scene > $O/stats_$scene.log 2>&1
  find $O/stats_$scene -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $O/cfg2_${scene}_kernel_stats.csv
  head -4 $O/cfg2_${scene}_kernel_stats.csv
done
pass() {  # name scene counters...
  local name=$1 scene=$2; shift 2
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc_${name}_$scene -o p -- python3 $B --steps 50 --warmup 5 --scene $scene > $O/pmc_${name}_$scene.log 2>&1 || echo "pmc pass $name/$scene failed"
  find $O/pmc_${name}_$scene -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $O/cfg2_${scene}_pmc_$name.csv || true
}
for scene in survey open; do
  pass FETCH_SIZE $scene FETCH_SIZE
  pass WRITE_SIZE $scene WRITE_SIZE
  pass sq_a $scene SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU
  pass sq_b $scene SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT
  pass sq_c $scene SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR
  pass sq_d $scene SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE
done
rm -rf $O/stats_* $O/pmc_*/  # keep the CSV copies only (the raw trees are large)
ls -la $O
