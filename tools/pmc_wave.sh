# VALU / SALU / LDS instruction counts of the stand-alone wavefront-per-sample cost kernel by cost term
set -e
export TMPDIR=/tmp
O=gpurun_out/r2p
mkdir -p $O
export KC_FUSED_CYCLE=0 KC_COST_KERNEL=wave
for w in 1,1,1,0,0 1,1,0,0,0 0,0,1,0,0 0,1,0,0,0; do
  rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d $O/w -o p -- python3 tools/run_cycles.py cfg2 ${SCENE:-open} $w 30 > $O/w.log 2>&1
  f=$(find $O/w -name "*counter_collection.csv" | head -1)
  python3 - "$f" "$w" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(list)
for row in csv.DictReader(open(sys.argv[1])):
    if "sample_cost_kernel" in row["Kernel_Name"]:
        acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
m = {c: sum(v) / len(v) for c, v in acc.items()}
print(sys.argv[2], {c: round(v / m["SQ_WAVES"]) for c, v in m.items() if c != "SQ_WAVES"}, "per wave (2 samples)")
PY
  rm -rf $O/w
done
