#!/bin/sh
# Builds experiment variants of the weight-stationary Linear kernel (parts of its round removed) to
# see which part sets the time per tile.  Usage: scratch/linear_exp.sh ; then run scratch/linear_exp.py
cd "$(dirname "$0")/.." || exit 1
for e in 1 2 3 4; do
  rm -rf /tmp/csrc_exp && cp -r asr_amd/csrc /tmp/csrc_exp && rm -f /tmp/csrc_exp/*.so
  f=/tmp/csrc_exp/linear_bf16.hip
  case $e in
    1) sed -i 's/if (row < rows \&\& col < N)$/if (row < rows \&\& col < N \&\& K < 0)/' $f ;;
    2) sed -i 's/acc\[g\]\[t\] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf\[t\]\[ks\], xf, acc\[g\]\[t\], 0, 0, 0);/acc[g][t][0] += (float)xf[t] * (float)wf[t][ks][0];/' $f ;;
    3) sed -i 's/if (tile + gridDim.y < ntiles) issue(/if (tile + gridDim.y < ntiles \&\& K < 0) issue(/' $f ;;
    4) sed -i 's/if (row < rows \&\& col < N)$/if (row < rows \&\& col < N \&\& K < 0)/; s/if (tile + gridDim.y < ntiles) issue(/if (tile + gridDim.y < ntiles \&\& K < 0) issue(/' $f ;;
  esac
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -Iinclude -I/tmp/csrc_exp -Wno-unused-function -fno-gpu-rdc \
    -o scratch/liblmufft_exp$e.so /tmp/csrc_exp/*.hip || exit 1
done
