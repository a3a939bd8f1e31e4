#!/bin/bash
# usage: tools/ab.sh "ENV1=.. ENV2=.." "ENV.." ...   -- alternate configurations, 3 rounds each, 1000 steps
for round in 1 2 3; do
  for cfg in "$@"; do
    out=$(env $cfg timeout -k 10 100 python bench.py --steps 1000 --warmup 50 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2), {k:round(v*1e3,1) for k,v in d['kernels_ms'].items()})")
    echo "[$cfg] $out"
  done
done
