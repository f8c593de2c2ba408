#!/bin/bash
# Builds exp/sanitize/build/driver_device: the whole library (mirt_api.hip's HOST code, the loader, the PNG writer) and the
# driver with AddressSanitizer + UndefinedBehaviorSanitizer on host code only (each -fsanitize= directly after -Xarch_host:
# device code is not instrumented).  Cross-compiles without a GPU; run the result on a GPU machine with run_device.sh.
set -e
cd "$(dirname "$0")"
mkdir -p build
C=../../raytracer_amd/csrc
CL=/opt/rocm/lib/llvm/bin/clang++
SF="-fsanitize=address -fsanitize=undefined -fno-omit-frame-pointer"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize \
    -Wno-unused-function -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined -Xarch_host -fno-omit-frame-pointer \
    -c -o build/mirt_api_san.o $C/mirt_api.hip
for f in mirt_host mirt_png; do $CL -O1 -g -std=c++14 -ffp-contract=off -pthread $SF -c -o build/${f}_san.o $C/$f.cpp; done
$CL -O1 -g -std=c++17 -DWITH_DEVICE $SF -c -o build/driver_san.o driver.cpp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -Xarch_host -fsanitize=address -Xarch_host -fsanitize=undefined \
    -o build/driver_device build/driver_san.o build/mirt_api_san.o build/mirt_host_san.o build/mirt_png_san.o -lpthread -lz
