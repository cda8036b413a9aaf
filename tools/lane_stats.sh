#!/bin/bash
# swaps the -DRT_STATS build in for one run of tools/lane_stats.py (GPU box only; the tree there is a scratch copy)
set -e
cp raytracing_amd/librt_hip.so /tmp/librt_hip_keep.so
cp build/librt_hip_stats.so raytracing_amd/librt_hip.so
timeout -k 10 200 python3 ${LS_SCRIPT:-tools/lane_stats.py} "$@" || true
cp /tmp/librt_hip_keep.so raytracing_amd/librt_hip.so
