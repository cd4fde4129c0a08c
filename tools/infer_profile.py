"""bf16-stream inference pipeline only, for rocprofv3 --kernel-trace --stats (10 batches)."""
import torch
from asr_amd import MelFrontend
from asr_amd.harness import LMUASRModel
dev = torch.device("cuda:0")
torch.manual_seed(1239)
B, L = 256, 63840
model = LMUASRModel(80, 384, 192, 3, 1000.0, 400, 34, num_attention_heads=6).to(dev).eval()
model.encoder.bf16_stream = True
fe = MelFrontend()
wave = 0.1 * torch.randn(B, L, device=dev)
samples = torch.full((B,), L, device=dev, dtype=torch.int32)
with torch.no_grad():
    for _ in range(10):
        mel, frames = fe(wave, samples)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            enc, _ = model.encoder(mel, frames)
        model.decoder.greedy(enc, frames)
torch.cuda.synchronize()
