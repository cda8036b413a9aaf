#!/bin/bash
# Runs a command with another build of librt_hip.so swapped in (GPU box only: the tree there is a scratch copy).
# usage: tools/probe_so.sh build/librt_hip_probe.so <command...>
set -e
so=$1; shift
cp raytracing_amd/librt_hip.so /tmp/librt_hip_keep.so
cp "$so" raytracing_amd/librt_hip.so
"$@" || true
cp /tmp/librt_hip_keep.so raytracing_amd/librt_hip.so
