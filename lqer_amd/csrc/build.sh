#!/bin/bash
# Build liblqer_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
OUT=../liblqer_hip.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function \
  -o "$OUT" api.hip quantize.hip pack.hip lowrank_xa.hip act_limbs.hip gemm_w4a8.hip gemm_w4a8_m256.hip gemm_smallm.hip "$@"
echo "built $(realpath $OUT)"
