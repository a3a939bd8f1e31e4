#!/bin/bash
# alternate the current library and the reference build (lib_ab) on the mapper bench lines, 2 rounds
cd /root/repo
one() { timeout -k 10 100 python bench.py $1 --no-cpu --steps 300 --warmup 30 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1e3,2), {k:round(v*1e3,1) for k,v in d.get('kernels_ms',{}).items()})"; }
for round in 1 2; do
  for which in new old; do
    if [ $which = old ]; then cp kompass-core_amd/lib/libkompass_hip.so /tmp/new.so; cp kompass-core_amd/lib_ab/libkompass_hip.so kompass-core_amd/lib/libkompass_hip.so; fi
    echo "[$which] cfg4: $(one --mapper) | mapper400: $(one '--ref mapper400') | bayes: $(one '--mapper --bayes')"
    if [ $which = old ]; then cp /tmp/new.so kompass-core_amd/lib/libkompass_hip.so; fi
  done
done
