#!/bin/bash
# scripts/build_variant_units.sh <name> "<units>" [-DFLAG=..] ... : like build_variant.sh, but only the named translation units
# are recompiled with the extra flags; the others are linked from the base build (eva_amd/lib/obj/).
set -e
R=$(cd $(dirname $0)/.. && pwd)
name=$1; units=$2; shift 2
D=$R/eva_amd/lib/variants/$name; mkdir -p $D/obj
pids=()
for u in $units; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $R/eva_amd/csrc/$u.hip -o $D/obj/$u.o 2>/dev/null &
  pids+=($!)
done
for p in ${pids[@]}; do wait $p; done
objs=""
for u in runtime elementwise ewprogram keyswitch rotate windows shard client scheduler; do
  if [ -f $D/obj/$u.o ]; then objs="$objs $D/obj/$u.o"; else objs="$objs $R/eva_amd/lib/obj/$u.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libeva_hip.so $objs
rm -rf $D/obj
echo built $D/libeva_hip.so
