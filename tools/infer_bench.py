"""The inference benchmark configuration: raw 16 kHz audio (batch 256 x 4 s) -> mel frontend -> 3-layer encoder
(384/192, 6 heads) -> greedy CTC, everything on the GPU; per-stage and end-to-end times."""
import json, sys, time, torch
from asr_amd import MelFrontend, _lib
from asr_amd.harness import LMUASRModel
dev = torch.device("cuda:0")
torch.manual_seed(1239)
B, L = 256, 63840
model = LMUASRModel(80, 384, 192, 3, 1000.0, 400, 34, num_attention_heads=6).to(dev).eval()
fe = MelFrontend()
wave = 0.1 * torch.randn(B, L, device=dev)
samples = torch.full((B,), L, device=dev, dtype=torch.int32)
out = {}
for dt, stream in ((torch.float32, False), (torch.bfloat16, False), (torch.bfloat16, True)):
    model.encoder.bf16_stream = stream
    def run():
        mel, frames = fe(wave, samples)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=dt == torch.bfloat16):
            enc, _ = model.encoder(mel, frames)
        return model.decoder.greedy(enc.float() if enc.dtype not in (torch.float32, torch.bfloat16) else enc, frames)
    with torch.no_grad():
        for _ in range(3): run()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): run()
        torch.cuda.synchronize(); ms = (time.perf_counter() - t0) / 10 * 1e3
        _lib.profile_enable(True); _lib.profile_reset()
        for _ in range(5): run()
        torch.cuda.synchronize(); prof = _lib.profile_read(); _lib.profile_enable(False)
    kern = {k: round(v[0] / 5 * 1e3, 1) for k, v in prof.items() if v[1]}
    frames = B * 400
    out[str(dt)[6:] + ("_stream" if stream else "")] = {"ms_per_batch": ms, "frames_per_s": frames / ms * 1e3, "audio_seconds_per_s": B * 3.99 / ms * 1e3,
                         "hip_kernel_us_per_batch": kern, "hip_kernels_ms": sum(kern.values()) / 1e3}
    print(str(dt)[6:] + ("_stream" if stream else ""), f"{ms:.2f} ms per batch of {B} x 4 s = {frames / ms / 1e3:.1f} M frames/s; HIP kernels {sum(kern.values()) / 1e3:.2f} ms", kern, flush=True)
json.dump(out, open(sys.argv[1], "w"), indent=1) if len(sys.argv) > 1 else None
