#!/bin/bash
# Host-code sanitizer passes (GPU code cannot be sanitized on this pool).  Run from the repo root on the CPU container:
#   tools/sanitize.sh ubsan | asan      the C++ host mirror (librt_host.so) rebuilt with the sanitizer, CPU suite, library put back
#   tools/sanitize.sh hip-ubsan         host side of librt_hip.so (-Xarch_host -fsanitize=undefined) into build/librt_hip_ubsan.so;
#                                       on the GPU box: swap it in, LD_LIBRARY_PATH=<clang rt dir>, pytest -m gpu -s, grep "runtime error"
#   tools/sanitize.sh fuzz              ASan + UBSan build of librt_host.so under tools/fuzz_gltf.py and tools/fuzz_hdr.py
# Round 2: UBSan found one signed-shift overflow (glTF base64 decoder, fixed); ASan and the hip-ubsan GPU run were clean;
# 1 200 corrupted .glb and 3 000 corrupted .hdr files: clean.
set -e
HOST="raytracing_amd/host/BVH.cpp raytracing_amd/host/Scene.cpp raytracing_amd/host/Gltf.cpp raytracing_amd/host/Image.cpp raytracing_amd/host/capi.cpp"
RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.ubsan_standalone-x86_64.so | head -1)
case "$1" in
ubsan|asan)
  cp raytracing_amd/librt_host.so /tmp/librt_host_keep.so
  trap 'cp /tmp/librt_host_keep.so raytracing_amd/librt_host.so' EXIT
  if [ "$1" = ubsan ]; then FLAG="-fsanitize=undefined -fno-sanitize-recover=undefined"; PRE=""; else FLAG="-fsanitize=address"; PRE=$(g++ -print-file-name=libasan.so); fi
  g++ -std=c++17 -O1 -g -ffp-contract=off -fPIC -shared -pthread $FLAG -o raytracing_amd/librt_host.so $HOST -Lraytracing_amd -lrt_hip -Wl,-rpath,'$ORIGIN'
  ASAN_OPTIONS=verify_asan_link_order=0:detect_leaks=0 LD_PRELOAD="$PRE" python -m pytest tests -m "not gpu" -x -q -s 2>&1 | grep -E "runtime error|AddressSanitizer|passed|failed" ;;
fuzz)
  cp raytracing_amd/librt_host.so /tmp/librt_host_keep.so
  trap 'cp /tmp/librt_host_keep.so raytracing_amd/librt_host.so' EXIT
  g++ -std=c++17 -O1 -g -ffp-contract=off -fPIC -shared -pthread -fsanitize=address,undefined -o raytracing_amd/librt_host.so $HOST -Lraytracing_amd -lrt_hip -Wl,-rpath,'$ORIGIN'
  for t in "fuzz_gltf.py 1 400" "fuzz_hdr.py 1 1500"; do
    ASAN_OPTIONS=verify_asan_link_order=0:detect_leaks=0 LD_PRELOAD=$(g++ -print-file-name=libasan.so) python tools/$t 2>&1 | grep -aE "RESULT|loaded|ERROR|runtime error" | tail -3
  done ;;
hip-ubsan)
  mkdir -p build
  hipcc --offload-arch=gfx950 -O3 -g -ffp-contract=off -fPIC -shared -std=c++17 -Xarch_host -fsanitize=undefined -o build/librt_hip_ubsan.so raytracing_amd/csrc/rt_hip.hip raytracing_amd/csrc/rt_comm.hip -ldl -Wl,"$RT"
  echo "built build/librt_hip_ubsan.so (runtime: $RT)" ;;
*) echo "usage: tools/sanitize.sh ubsan|asan|fuzz|hip-ubsan"; exit 2 ;;
esac
