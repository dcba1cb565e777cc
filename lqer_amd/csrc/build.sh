#!/bin/bash
# Build liblqer_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [extra hipcc flags]
set -e
cd "$(dirname "$0")"
make -s -j"$(nproc)" EXTRA="$*"
echo "built $(realpath ../liblqer_hip.so)"
