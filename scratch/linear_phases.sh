#!/bin/sh
# Debug build of the weight-stationary Linear kernel that accumulates, for thread 0 of workgroup 0,
# the shader clocks spent in each phase of a round and writes them over out[0][0..15] (as floats in
# the bf16 buffer's first bytes are useless, so they go to the LAST row's tail via a float view):
# see scratch/linear_phases.py.  Output: scratch/liblmufft_phases.so
cd "$(dirname "$0")/.." || exit 1
rm -rf /tmp/csrc_ph && cp -r asr_amd/csrc /tmp/csrc_ph && rm -f /tmp/csrc_ph/*.so
python3 - <<'PY'
p='/tmp/csrc_ph/linear_bf16.hip'; s=open(p).read()
def ins(marker, text, after=True, count=1):
    global s
    i = s.index(marker)
    if after: i += len(marker)
    s = s[:i] + text + s[i:]
# accumulators
ins("  const long ntiles = (rows + kWsRows - 1) / kWsRows;\n  long tile = blockIdx.x;", "\n  long long ph[8] = {0,0,0,0,0,0,0,0}; long long tc = clock64(); const long long tstart = tc;\n#define PH(i) { const long long n_ = clock64(); ph[i] += n_ - tc; tc = n_; }\n")
ins('    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n    __syncthreads();\n    if (tile + gridDim.x < ntiles)', "    PH(0)\n", after=False)
s = s.replace('    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n    __syncthreads();\n    if (tile + gridDim.x < ntiles) issue(stage ^ 1, (tile + gridDim.x) * kWsRows);\n',
              '    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");\n    PH(1)\n    __syncthreads();\n    PH(2)\n    if (tile + gridDim.x < ntiles) issue(stage ^ 1, (tile + gridDim.x) * kWsRows);\n    PH(3)\n')
s = s.replace("      // D[n = 4 q + v][row = l15]: four consecutive columns of one row per lane\n", "      PH(4)\n      // D[n = 4 q + v][row = l15]: four consecutive columns of one row per lane\n")
s = s.replace("    __syncthreads();\n\n    const long row0 = tile * kWsRows;", "    PH(5)\n    __syncthreads();\n    PH(6)\n\n    const long row0 = tile * kWsRows;")
# end of kernel: after the loop
end = s.index("\n  }\n}\n\n}  // namespace lmufft")
s = s[:end] + "\n    PH(7)" + "\n  }\n  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && out) {\n    float* dbg = reinterpret_cast<float*>(out + (tile - gridDim.x) * kWsRows * N);\n    for (int i = 0; i < 8; ++i) dbg[i] = (float)ph[i];\n    dbg[8] = (float)(clock64() - tstart);\n  }\n}\n\n}  // namespace lmufft" + s[end + len("\n  }\n}\n\n}  // namespace lmufft"):]
open(p,'w').write(s)
PY
hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -shared -Iinclude -I/tmp/csrc_ph -Wno-unused-function -fno-gpu-rdc \
  -o scratch/liblmufft_phases.so /tmp/csrc_ph/*.hip
