#!/bin/bash
# A/B timing of library variants on the GPU box (a scratch copy of the tree): for every build/var_<name>.so given, swap it
# in for one run of tools/quick_bench.py, then put the shipped library back.  usage: tools/ab.sh [-w workload] [-s steps] name ...
W=cornellbox; S=4
while getopts "w:s:" o; do case $o in w) W=$OPTARG;; s) S=$OPTARG;; esac; done; shift $((OPTIND-1))
cp raytracing_amd/librt_hip.so /tmp/librt_hip_keep.so || exit 1
# (the first run on a fresh box reads 1 % slow: one discarded warm-up run precedes the series)
timeout -k 10 240 python3 -u tools/quick_bench.py 2 $W > /dev/null 2>&1
for v in base "$@"; do
  if [ "$v" = base ]; then cp /tmp/librt_hip_keep.so raytracing_amd/librt_hip.so; else cp build/var_$v.so raytracing_amd/librt_hip.so || exit 1; fi
  echo "== $v"
  timeout -k 10 240 python3 -u tools/quick_bench.py $S $W 2>&1 | grep -v "^$" || { cp /tmp/librt_hip_keep.so raytracing_amd/librt_hip.so; exit 1; }
done
cp /tmp/librt_hip_keep.so raytracing_amd/librt_hip.so
