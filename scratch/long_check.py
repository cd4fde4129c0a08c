"""30 s utterances (T = 3000) through the whole inference pipeline: bf16 (both stream settings) against fp32."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from asr_amd import MelFrontend
from asr_amd.harness import LMUASRModel

dev = torch.device("cuda:0")
torch.manual_seed(5)
B, L = 8, 479840
model = LMUASRModel(80, 384, 192, 3, 1000.0, 3000, 34, num_attention_heads=6).to(dev).eval()
fe = MelFrontend()
wave = 0.1 * torch.randn(B, L, device=dev)
samples = torch.tensor([L, L - 16000, L // 2, L, 40000, L, L - 1, 160000], device=dev, dtype=torch.int32)
with torch.no_grad():
    mel, frames = fe(wave, samples)
    print("frames", frames.tolist())
    ref, _ = model.encoder(mel, frames)
    for stream in (False, True):
        model.encoder.bf16_stream = stream
        with torch.autocast("cuda", dtype=torch.bfloat16):
            enc, _ = model.encoder(mel, frames)
        torch.cuda.synchronize()
        valid = (torch.arange(enc.shape[1], device=dev)[None, :] < frames[:, None])
        err = ((enc.float() - ref).abs() * valid[..., None]).max().item()
        print(f"bf16_stream={stream}: max |enc - fp32| over valid frames {err:.4f} (|ref| max {ref.abs().max().item():.2f})")
        t0 = time.time()
        for _ in range(5):
            with torch.autocast("cuda", dtype=torch.bfloat16):
                enc, _ = model.encoder(mel, frames)
        torch.cuda.synchronize()
        print(f"   {(time.time() - t0) / 5 * 1e3:.2f} ms per batch of {B} x 30 s")
