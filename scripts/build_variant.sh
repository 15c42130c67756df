#!/bin/bash
# scripts/build_variant.sh <name> [-DFLAG=..] ... : libeva_hip.so built with extra compile flags into
# eva_amd/lib/variants/<name>/ (git-ignored; travels with gpurun) for A/B runs (scripts/ab_bench.sh label:ENV@name)
set -e
R=$(cd $(dirname $0)/.. && pwd)
name=$1; shift
D=$R/eva_amd/lib/variants/$name; mkdir -p $D/obj
pids=()
for u in runtime elementwise ewprogram keyswitch rotate windows shard client scheduler; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c $R/eva_amd/csrc/$u.hip -o $D/obj/$u.o &
  pids+=($!)
done
for p in ${pids[@]}; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/libeva_hip.so $D/obj/*.o
rm -rf $D/obj
echo built $D/libeva_hip.so
