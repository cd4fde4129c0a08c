"""CPU restatement of the residual add + LayerNorm of the reference's encoder -- TEST INFRASTRUCTURE,
NOT A PRODUCT PATH (only tests/ and __graft_entry__.smoke() may import it).

Follows `output = self.layer_norm(output + residual)` (src/models/lmu_encoder.py:97) and
`x = layer_norm(x)` (src/models/lmu_encoder.py:274).  The reference calls torch.nn.LayerNorm; the
arithmetic restated here is its documented definition: mean and BIASED variance over the last
dimension, y = (v - mean) / sqrt(var + eps) * weight + bias.  Pinned by tests/test_layernorm.py
against torch.nn.functional.layer_norm on the CPU (the implementation the reference runs) and,
through the whole encoder, by tests/golden/model_e2e.npz.
"""
import numpy as np


def add_layer_norm(x, residual, weight, bias, eps=1e-5):
    """x, residual: [..., C] (residual may be None); statistics in float64, result float32."""
    v = np.asarray(x, dtype=np.float32)
    if residual is not None:
        v = v + np.asarray(residual, dtype=np.float32)  # the reference adds in fp32 (autocast promotes)
    v = v.astype(np.float64)
    mean = v.mean(axis=-1, keepdims=True)
    var = ((v - mean) ** 2).mean(axis=-1, keepdims=True)
    y = (v - mean) / np.sqrt(var + eps) * np.asarray(weight, np.float64) + np.asarray(bias, np.float64)
    return y.astype(np.float32)
