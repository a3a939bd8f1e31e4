#!/bin/bash
# alternate the current library and the reference build (lib_ab) on the large lattices
cd /root/repo
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), {k:round(v*1e3,1) for k,v in d['kernels_ms'].items()})"; }
for round in 1 2; do
  for which in new old; do
    if [ $which = old ]; then cp kompass-core_amd/lib/libkompass_hip.so /tmp/new.so; cp kompass-core_amd/lib_ab/libkompass_hip.so kompass-core_amd/lib/libkompass_hip.so; fi
    for cs in "cfg3 open" "cfg3 mid" "cfg5 open" "cfg5 mid"; do
      set -- $cs
      out=$(timeout -k 10 100 python bench.py --config $1 --scene $2 --steps 300 --warmup 30 --no-cpu --only-headline 2>/dev/null | line)
      echo "[$which] $cs: $out"
    done
    if [ $which = old ]; then cp /tmp/new.so kompass-core_amd/lib/libkompass_hip.so; fi
  done
done
