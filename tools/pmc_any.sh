# SQ counters of every kernel of an arbitrary python command: bash tools/pmc_any.sh bench.py --ref cost5k --steps 20 --warmup 5
set -e
export TMPDIR=/tmp
O=gpurun_out/pmc_any
mkdir -p $O
i=0
for ctrs in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY"; do
  i=$((i+1))
  rm -rf $O/w
  rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $O/w -o p -- python3 "$@" > $O/w$i.log 2>&1
  f=$(find $O/w -name "*counter_collection.csv" | head -1)
  python3 - "$f" <<'PY'
import csv, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"].split("(")[0].split("<")[0].replace("void kc::", "")
    acc[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, cs in acc.items():
    print(k, {c: round(sum(v) / len(v)) for c, v in cs.items()}, "launches", len(next(iter(cs.values()))))
PY
done
rm -rf $O/w
