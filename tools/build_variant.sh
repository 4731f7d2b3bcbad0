#!/bin/bash
# A compile-time variant of the library for A/B timing inside one gpurun call:
#   tools/build_variant.sh <name> <unit.hip> [-DFLAG=1 ...]     ->  topicmodelsvb.jl_amd/libtmvb_hip_<name>.so
# = the shipped objects (topicmodelsvb.jl_amd/build/*.o, build them first) with <unit.hip> recompiled under the extra flags.
# Select it with TMVB_LIB_VARIANT=<name> (topicmodelsvb.jl_amd/_lib.py); the .so is git-ignored and travels with gpurun.
set -e
R=$(cd "$(dirname "$0")/.." && pwd); P=$R/topicmodelsvb.jl_amd
name=$1; unit=$2; shift 2
mkdir -p "$P/build/var_$name"
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -Wno-pass-failed -I "$R/include" -I "$P/csrc" "$@" -c "$P/csrc/$unit" -o "$P/build/var_$name/$unit.o"
objs=""
for o in "$P"/build/*.hip.o; do
  if [ "$(basename "$o")" = "$unit.o" ]; then objs="$objs $P/build/var_$name/$unit.o"; else objs="$objs $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o "$P/libtmvb_hip_$name.so" $objs -ldl -Wl,-rpath,/opt/rocm/lib
# stamp: what the variant was built from (content hash of the sources + the flags), so that a test can tell a current variant from a stale one without
# trusting file times (a git checkout or a snapshot copy resets them)
( cd "$R" && cat topicmodelsvb.jl_amd/csrc/* include/tmvb.h | sha256sum | cut -c1-16; echo "$unit $*" ) > "$P/libtmvb_hip_$name.stamp"
echo "built $P/libtmvb_hip_$name.so"
