#!/bin/bash
# Developer helper (GPU box): A/B of two builds of the library on ONE box, alternating -- vclust_amd/_ab/old.so against
# vclust_amd/_ab/new.so (copy the builds there first); per-scope milliseconds of bench.py.  usage: ab_lib.sh [workload] [rounds]
W=${1:-phage-100k}; R=${2:-3}
cp vclust_amd/libvclust_gpu.so /tmp/keep.so
for i in $(seq $R); do
  for v in old new; do cp vclust_amd/_ab/$v.so vclust_amd/libvclust_gpu.so; echo -n "$v "; bash tools/micro/bench_scopes.sh $W 5; done
done
cp /tmp/keep.so vclust_amd/libvclust_gpu.so
