#!/bin/sh
# Alternative build of the library with extra -D switches (measured variants that are off by default):
#   tools/build_alt.sh -DLMUFFT_FFT_SHUFFLE    ->  tools/liblmufft_alt.so   (run with LMUFFT_LIBRARY pointing at it)
cd "$(dirname "$0")/.." && hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared "$@" \
  -Iinclude -Iasr_amd/csrc -Wno-unused-function -fno-gpu-rdc -o tools/liblmufft_alt.so asr_amd/csrc/*.hip
