"""CPU restatement of the encoder's Linear layers under bf16 autocast -- TEST INFRASTRUCTURE,
NOT A PRODUCT PATH (only tests/ and __graft_entry__.smoke() may import it).

Follows torch.nn.Linear as the reference calls it (src/models/lmu_encoder.py:68-70, 94, 244, 278):
y = x W^T + b.  Under autocast the operands are rounded to bf16 and the product is accumulated
in fp32 (here: exactly, in float64, from the bf16-rounded operands); the result is rounded to
bf16 once.  `linear_add_layer_norm` appends `layer_norm(output + residual)`
(src/models/lmu_encoder.py:97) through layernorm_oracle.add_layer_norm.  Pinned by
tests/test_linear.py against torch.nn.functional.linear on the CPU.
"""
import numpy as np

from .layernorm_oracle import add_layer_norm


def round_bf16(a):
    """float32 -> nearest-even bfloat16, returned as float32."""
    a = np.ascontiguousarray(a, dtype=np.float32)
    u = a.view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32).reshape(a.shape)


def linear_bf16(x, weight, bias=None):
    """x [..., K], weight [N, K], bias [N] -> [..., N]: bf16 operands, exact accumulation, bf16 result."""
    y = round_bf16(x).astype(np.float64) @ round_bf16(weight).astype(np.float64).T
    if bias is not None:
        y = y + round_bf16(bias).astype(np.float64)
    return round_bf16(y.astype(np.float32))


def linear_add_layer_norm(x, weight, bias, residual, ln_weight, ln_bias, eps=1e-5):
    return add_layer_norm(linear_bf16(x, weight, bias), residual, ln_weight, ln_bias, eps)


def linear_backward_bf16(x, weight, dy):
    """Gradients of y = x W^T + b as bf16 autocast training computes them here: operands rounded to bf16, exact
    accumulation; dx rounded to bf16 (it is an activation gradient), dW and db kept in fp64 (the kernels return
    fp32).  What autograd derives for the reference's nn.Linear layers (src/models/lmu_encoder.py:40-43, 184, 229)."""
    xb = round_bf16(x).astype(np.float64).reshape(-1, np.shape(x)[-1])
    wb = round_bf16(weight).astype(np.float64)
    gb = round_bf16(dy).astype(np.float64).reshape(-1, np.shape(dy)[-1])
    dx = round_bf16((gb @ wb).astype(np.float32)).reshape(np.shape(x))
    return dx, gb.T @ xb, gb.sum(axis=0)
