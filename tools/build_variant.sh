#!/bin/bash
# tools/build_variant.sh NAME [HIPFLAGS_EXTRA...]  -- a second build of libkompass_hip.so into
# kompass-core_amd/lib_ab/NAME/ (travels to the GPU box, stays out of git); KC_SRC=/path/to/tree builds
# another checkout's kompass-core_amd (e.g. a `git worktree` of an older commit) for bisecting.
# Run with KOMPASS_HIP_LIB=kompass-core_amd/lib_ab/NAME/libkompass_hip.so (tools/ab_libs.sh).
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
src=${KC_SRC:-$root}/kompass-core_amd
out=$root/kompass-core_amd/lib_ab/$name
mkdir -p "$out"
make -C "$src" -j6 OUT="$out" HIPFLAGS_EXTRA="$*" >/dev/null
ls -la "$out/libkompass_hip.so"
