"""Reads the per-phase shader clocks that the debug build of scratch/linear_phases.sh leaves in the
last output row (run with LMUFFT_LIBRARY=scratch/liblmufft_phases.so)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from asr_amd.linear import linear_bf16

dev = torch.device("cuda:0")
names = ["rest of previous round (epilogue 2nd half excluded)", "wait vmcnt(0)", "barrier", "DMA issue",
         "MFMA phases (both halves)", "epilogue writes", "barrier 2", "drain (copy out)"]
for rows, N in ((102400, 384), (102400, 1152)):
    x = torch.randn(rows, 384, device=dev).bfloat16()
    W = (torch.randn(N, 384, device=dev) / 20).bfloat16()
    b = torch.randn(N, device=dev)
    for _ in range(3):
        y = linear_bf16(x, W, b)
    torch.cuda.synchronize()
    slots = 256 if N == 384 else 80
    ntiles = rows // 64
    tiles = (ntiles + slots - 1) // slots  # rounds of workgroup 0
    last = (tiles - 1) * slots  # its last tile: it left its clocks in that tile's first row
    d = y[last * 64].view(torch.uint8)[:36].view(torch.float32).cpu().tolist()
    print(f"rows={rows} N={N}: total {d[8]:.0f} clocks for about {tiles} tiles")
    for n, v in zip(names, d[:8]):
        print(f"   {n:55s} {v:9.0f}  ({v / tiles:7.0f} per tile)")
