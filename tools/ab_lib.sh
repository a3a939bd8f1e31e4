#!/bin/bash
# alternate the current library and the reference build (lib_ab), 3 rounds
cd /root/repo
for round in 1 2 3; do
  for which in new old; do
    if [ $which = old ]; then cp kompass-core_amd/lib/libkompass_hip.so /tmp/new.so; cp kompass-core_amd/lib_ab/libkompass_hip.so kompass-core_amd/lib/libkompass_hip.so; fi
    out=$(timeout -k 10 100 python bench.py --steps 1000 --warmup 100 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,2), {k:round(v*1e3,1) for k,v in d['kernels_ms'].items()}, 'mid', round(d['mid_density']['ms_per_step']*1e3,1), '(kernel', round(max(d['mid_density']['kernels_ms'].values())*1e3,1), ') open', round(d['open_space']['ms_per_step']*1e3,1), '(kernel', round(max(d['open_space']['kernels_ms'].values())*1e3,1), ')', 'upd', round(d['extras']['update_and_cycle_ms']*1e3,1))")
    echo "[$which] $out"
    if [ $which = old ]; then cp /tmp/new.so kompass-core_amd/lib/libkompass_hip.so; fi
  done
done
