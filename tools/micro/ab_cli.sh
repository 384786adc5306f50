#!/bin/bash
# Developer helper: A/B of two builds of the library on ONE box (boxes differ by +-0.15 s): alternates ab/old.so and
# ab/new.so under the CLI timing (tools/timing.py cli) and prints the upload tail and the total of each run.
# usage (through gpurun): bash tools/micro/ab_cli.sh [families] [rounds]     -- with ab/old.so and ab/new.so in the tree
NF=${1:-10000}; R=${2:-4}
cp vclust_amd/libvclust_gpu.so /tmp/cur.so
python tools/timing.py cli $NF > /dev/null 2>&1
for i in $(seq $R); do for v in old new; do cp ab/$v.so vclust_amd/libvclust_gpu.so; echo -n "$v "; python tools/timing.py cli $NF 2>&1 | grep -E "^== total|resident" | tr '\n' ' ' | sed 's/\[vg host\] ingest: //g; s/@[0-9.]*//g'; echo; done; done
cp /tmp/cur.so vclust_amd/libvclust_gpu.so
