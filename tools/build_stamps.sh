#!/bin/sh
# Debug build of the library with in-kernel timestamps (see tools/stamps.py).
cd "$(dirname "$0")/.." && hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -DLMUFFT_STAMPS=${1:-1} \
  -Iinclude -Iasr_amd/csrc -Wno-unused-function -fno-gpu-rdc -o tools/liblmufft_stamps.so asr_amd/csrc/*.hip
