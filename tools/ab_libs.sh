#!/bin/bash
# tools/ab_libs.sh ROUNDS "python tools/x.py args" NAME [NAME...] -- the same command on builds of the library
# under kompass-core_amd/lib_ab/NAME/ (tools/build_variant.sh), alternating, ROUNDS times: a same-box A/B.
rounds=$1; cmd=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
for r in $(seq 1 "$rounds"); do
  for name in "$@"; do
    # (a name that starts with "old_" is a build of an older commit: the binding tolerates entries it lacks)
    old=0; case "$name" in old_*) old=1;; esac
    echo "[$name] $(KOMPASS_HIP_LIB_OLD=$old KOMPASS_HIP_LIB=$root/kompass-core_amd/lib_ab/$name/libkompass_hip.so timeout -k 10 300 $cmd 2>&1 | tail -n 1)"
  done
done
