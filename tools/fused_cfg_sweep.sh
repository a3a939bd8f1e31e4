# roll-out tile shapes (KC_FUSED_CFG = samples,threads per workgroup) on the three-kernel cycle
cd /root/repo
line() { python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ms_per_step']*1e3,1), {k:round(v*1e3,1) for k,v in d['kernels_ms'].items()})"; }
for fc in "32,1024" "32,512" "16,512" "16,256"; do
  for cs in "cfg2 survey" "cfg2 open" "cfg1 survey" "cfg5 mid" "cfg3 mid"; do
    set -- $cs
    out=$(KC_FUSED_CFG=$fc timeout -k 10 100 python bench.py --config $1 --scene $2 --split --steps 300 --warmup 30 --no-cpu --only-headline 2>/dev/null | line)
    echo "[$fc] $cs: $out"
  done
done
