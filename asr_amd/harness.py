"""Minimal end-to-end harness around the HIP path: the reference's encoder stack with this
package's LMUFFT layers and CTC head, so that a reference checkpoint can be evaluated on the GPU
and greedy decodes can be compared token by token.

The LMUFFT layers, the attention core, the layer norms, the mel frontend and the CTC head are HIP
kernels of this package, and so are the input / output / attention projections under bf16
autocast (fp32 projections are torch's linear).  In training under bf16 autocast (grad enabled)
the same ops run this package's forward AND backward kernels -- attention with dropout on its
weights, the Linear layers, layer norm with the dropout that follows it, the CTC loss -- behind
autograd functions.  Module and
parameter names follow the reference so that its state_dict loads unchanged:
  encoder.input_projection, encoder.attention_layers.{i}.attention.{w_q,w_k,w_v,w_o,layer_norm},
  encoder.lmu_layers.{i}, encoder.layer_norms.{i}, encoder.output_projection,
  decoder.ctc_projection
(reference: src/models/lmu_encoder.py:23-47, 183-229; src/models/asr_model.py:69-102).
Dropout (on the attention weights and after the LMU blocks' layer norms, as the reference) is active
in train() mode and the identity in eval().
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import torch
from torch import nn

from .attention import (SUPPORTED_HEAD_WIDTHS, masked_attention, masked_attention_packed, masked_attention_train,
                        masked_attention_train_packed)
from .layernorm import MAX_WIDTH as _LN_MAX_WIDTH, add_layer_norm, add_layer_norm_train
from .linear import (fused_linear_train, linear_bf16, linear_train, prepare_weights, supported as linear_supported,
                     train_supported as linear_train_supported)
from .ctc import CTCDecoder
from .lmu import LMUFFT


# Which implementation ran: every op of the harness that has both a HIP kernel of this package and a
# torch formulation counts its calls here ("hip:<op>" / "torch:<op>"), and inside `strict_kernels()`
# a torch formulation raises instead of running -- so a benchmark cannot silently time hipBLASLt or
# torch's attention.  (fp32 Linear layers are torch ops by design and are counted as "torch:linear";
# strict mode therefore only makes sense under bf16 autocast, where every op has a kernel here.)
PATHS: dict = {}
_STRICT = [False]
# True: in training under bf16 autocast the layer norm behind an LMUFFT layer returns that layer's gradient already
# multiplied by (h > 0) and the layer skips its own masking pass (the same gradients bit for bit)
FOLD_RELU_BACKWARD = True


class strict_kernels:
    """Context manager: inside it, any harness op that would run as a torch op raises RuntimeError."""

    def __enter__(self):
        self._old = _STRICT[0]
        _STRICT[0] = True
        return self

    def __exit__(self, *exc):
        _STRICT[0] = self._old
        return False


def _took(kind: str, op: str) -> None:
    key = f"{kind}:{op}"
    PATHS[key] = PATHS.get(key, 0) + 1
    if kind == "torch" and _STRICT[0]:
        raise RuntimeError(f"asr_amd.harness: `{op}` would run as a torch op here (strict_kernels is on); "
                           "the HIP kernels need GPU tensors, bf16 autocast and supported widths / dtypes")


def _bf16_inference() -> bool:
    return (not torch.is_grad_enabled() and torch.is_autocast_enabled()
            and torch.get_autocast_dtype("cuda") == torch.bfloat16)


def _bf16_params(layer: nn.Linear):
    """(weight bf16, bias bf16, bias bf16-rounded as fp32) of a Linear, kept on the layer itself
    (so the copies die with it) and rebuilt when the parameters' version counters move."""
    ver = (layer.weight._version, layer.weight.data_ptr(),
           None if layer.bias is None else (layer.bias._version, layer.bias.data_ptr()))
    hit = layer.__dict__.get("_bf16_copy")
    if hit is None or hit[0] != ver:
        b16 = None if layer.bias is None else layer.bias.detach().to(torch.bfloat16)
        hit = (ver, layer.weight.detach().to(torch.bfloat16), b16, None if b16 is None else b16.float())
        layer.__dict__["_bf16_copy"] = hit
    return hit[1], hit[2], hit[3]


def _linear(layer: nn.Linear, x: torch.Tensor) -> torch.Tensor:
    """layer(x).  For inference under bf16 autocast the bf16 copies of weight and bias are kept
    instead of being re-cast by autocast at every call, and the product runs in this package's
    GEMM kernel (torch's linear when the widths are not multiples of 8)."""
    if x.is_cuda and _bf16_inference() and layer.weight.dtype == torch.float32:
        w16, b16, b32 = _bf16_params(layer)
        xb = x.to(torch.bfloat16)
        if linear_supported(xb, w16):
            _took("hip", "linear")
            return linear_bf16(xb, w16, b32)
        _took("torch", "linear")
        return torch.nn.functional.linear(xb, w16, b16)
    if (torch.is_grad_enabled() and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
            and linear_train_supported(x, layer.weight)):
        # training under bf16 autocast: forward, input gradient and weight gradient on this package's kernels
        _took("hip", "linear")
        return linear_train(x, layer.weight, layer.bias)
    _took("torch", "linear")
    return layer(x)


def _linear_layer_norm(layer: nn.Linear, x: torch.Tensor, ln: nn.LayerNorm, residual: torch.Tensor,
                       out_dtype: Optional[torch.dtype]) -> torch.Tensor:
    """ln(layer(x) + residual).  Two kernels: at the encoder's sizes the GEMM followed by the one-pass
    add + layer-norm kernel is faster than linear_add_layer_norm_bf16's fused epilogue, whose
    residual loads sit inside the persistent GEMM loop."""
    return _layer_norm(ln, _linear(layer, x), residual, out_dtype)


def _qkv_projection(wq: nn.Linear, wk: nn.Linear, wv: nn.Linear, x: torch.Tensor) -> torch.Tensor:
    """[w_q(x) | w_k(x) | w_v(x)] as ONE GEMM over the concatenated weights (inference; the
    concatenation, in bf16 under bf16 autocast, is cached on the parameters' version counters)."""
    bf16 = torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
    dt = torch.bfloat16 if bf16 else x.dtype
    layers = (wq, wk, wv)
    ver = tuple((l.weight._version, l.weight.data_ptr(), l.bias._version, l.bias.data_ptr()) for l in layers) + (dt,)
    hit = wq.__dict__.get("_qkv_copy")  # kept on the first layer itself: it dies with it
    if hit is None or hit[0] != ver:
        bias = torch.cat([l.bias.detach() for l in layers]).to(dt)
        hit = (ver, torch.cat([l.weight.detach() for l in layers]).to(dt), bias, bias.float())
        wq.__dict__["_qkv_copy"] = hit
    xd = x.to(dt)
    if bf16 and linear_supported(xd, hit[1]):
        _took("hip", "qkv_projection")
        return linear_bf16(xd, hit[1], hit[3])
    _took("torch", "qkv_projection")
    with torch.autocast("cuda", enabled=False):
        return torch.nn.functional.linear(xd, hit[1], hit[2])


def _layer_norm_trains_on_hip(ln: nn.LayerNorm, C: int, dtype, residual_dtype=None) -> bool:
    """Whether _layer_norm takes the HIP training kernels for a GPU input of this width and dtype."""
    return (torch.is_grad_enabled() and ln.elementwise_affine and ln.bias is not None and C % 4 == 0
            and C <= _LN_MAX_WIDTH and dtype in (torch.float32, torch.bfloat16) and ln.weight.dtype == torch.float32
            and (residual_dtype is None or residual_dtype in (torch.float32, torch.bfloat16)))


def _layer_norm(ln: nn.LayerNorm, x: torch.Tensor, residual: Optional[torch.Tensor] = None,
                out_dtype: Optional[torch.dtype] = None, dropout: Optional[nn.Dropout] = None, twin: bool = False,
                relu_input: bool = False):
    """dropout(ln(x + residual)): the one-pass HIP kernels on the GPU (the dropout rides on the same pass), torch
    otherwise.  `twin`: returns (float32 result, its bfloat16 rounding) -- from the same pass on the training path."""
    C = x.shape[-1]
    p_drop = dropout.p if (dropout is not None and dropout.training) else 0.0
    if (x.is_cuda and not torch.is_grad_enabled() and p_drop == 0.0 and not twin and ln.elementwise_affine and ln.bias is not None
            and C % 4 == 0 and C <= _LN_MAX_WIDTH and x.dtype in (torch.float32, torch.bfloat16)
            and (residual is None or residual.dtype in (torch.float32, torch.bfloat16))):
        _took("hip", "layer_norm")
        return add_layer_norm(x, residual, ln.weight, ln.bias, ln.eps, out_dtype=out_dtype)
    if x.is_cuda and _layer_norm_trains_on_hip(ln, C, x.dtype, None if residual is None else residual.dtype):
        _took("hip", "layer_norm")  # training: the same forward kernel, and its backward, behind an autograd function
        return add_layer_norm_train(x, residual, ln.weight, ln.bias, ln.eps, out_dtype, p_drop, twin=twin,
                                    relu_input=relu_input)
    if relu_input:  # (the producer of x was told that its gradient arrives masked: never skipped silently)
        raise RuntimeError("_layer_norm: relu_input needs the HIP training kernels")
    _took("torch", "layer_norm")
    y = ln(x if residual is None else x + residual)
    y = y if out_dtype is None else y.to(out_dtype)
    y = y if dropout is None else dropout(y)
    return (y, y.to(torch.bfloat16)) if twin else y


class _MultiHeadAttention(nn.Module):
    """softmax(Q K^T / sqrt(d_k), masked) V, output projection, residual, layer norm
    (reference: src/models/lmu_encoder.py:48-99)."""

    def __init__(self, d_model: int, num_heads: int, dropout: float = 0.0):
        super().__init__()
        if d_model % num_heads:
            raise ValueError("d_model must be a multiple of num_heads")
        self.num_heads, self.d_k = num_heads, d_model // num_heads
        self.dropout = nn.Dropout(dropout)  # on the attention weights (src/models/lmu_encoder.py:45, 86); identity in eval
        self.w_q = nn.Linear(d_model, d_model)
        self.w_k = nn.Linear(d_model, d_model)
        self.w_v = nn.Linear(d_model, d_model)
        self.w_o = nn.Linear(d_model, d_model)
        self.layer_norm = nn.LayerNorm(d_model)

    def forward(self, x: torch.Tensor, lengths: Optional[torch.Tensor], out_dtype: Optional[torch.dtype] = None,
                x_bf16: Optional[torch.Tensor] = None) -> torch.Tensor:
        """`x_bf16`: x rounded to bfloat16 when the caller already has it (the projections read that copy under
        bf16 autocast; the residual connection keeps x)."""
        B, T, C = x.shape
        if (x.is_cuda and self.d_k in SUPPORTED_HEAD_WIDTHS and not torch.is_grad_enabled()
                and not (self.training and self.dropout.p > 0)):
            # HIP kernel: no [B, heads, T, T] score tensor, no per-utterance mask loop
            _took("hip", "attention")
            if all(l.bias is not None for l in (self.w_q, self.w_k, self.w_v)):
                # one GEMM for the three projections; the kernel reads q, k, v as column blocks of its output
                ctx = masked_attention_packed(_qkv_projection(self.w_q, self.w_k, self.w_v, x), self.num_heads, lengths)
            else:
                ctx = masked_attention(_linear(self.w_q, x), _linear(self.w_k, x), _linear(self.w_v, x),
                                       self.num_heads, lengths)
            return _linear_layer_norm(self.w_o, ctx, self.layer_norm, x, out_dtype)
        if x.is_cuda and self.d_k in SUPPORTED_HEAD_WIDTHS:
            # training: HIP forward with dropout and HIP backward behind an autograd function; the projections
            # and the layer norm stay torch ops (autograd)
            _took("hip", "attention")
            p_drop = self.dropout.p if self.training else 0.0
            qkv = (self.w_q, self.w_k, self.w_v)
            if (torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16
                    and all(l.bias is not None for l in qkv) and linear_train_supported(x, self.w_q.weight)):
                # one GEMM for the three projections in both directions: x is cast and read once, the three input
                # gradients arrive summed; the attention kernels read and write column blocks of its output
                _took("hip", "qkv_projection")
                packed = fused_linear_train(x if x_bf16 is None else x_bf16, qkv)
                ctx = masked_attention_train_packed(packed, self.num_heads, lengths, p_drop)
            else:
                q, k, v = _linear(self.w_q, x), _linear(self.w_k, x), _linear(self.w_v, x)
                ctx = masked_attention_train(q, k, v, self.num_heads, lengths, p_drop)
            return _layer_norm(self.layer_norm, _linear(self.w_o, ctx), x, out_dtype)
        _took("torch", "attention")
        split = lambda t: t.view(B, T, self.num_heads, self.d_k).transpose(1, 2)  # noqa: E731
        q, k, v = split(self.w_q(x)), split(self.w_k(x)), split(self.w_v(x))
        scores = q @ k.transpose(-2, -1) / math.sqrt(self.d_k)
        if lengths is not None:
            # mask[b, i, j] = (i < len_b) and (j < len_b); masked scores are SET to -1e9, so a fully
            # masked row (i >= len_b) becomes a uniform distribution, exactly as in the reference
            valid = torch.arange(T, device=x.device)[None, :] < lengths.to(x.device)[:, None]
            mask = valid[:, :, None] & valid[:, None, :]
            scores = scores.masked_fill(~mask[:, None], -1e9)
        ctx = (self.dropout(torch.softmax(scores, dim=-1)) @ v).transpose(1, 2).reshape(B, T, C)
        return _layer_norm(self.layer_norm, self.w_o(ctx), x, out_dtype)


class _SelfAttention(nn.Module):
    def __init__(self, d_model: int, num_heads: int, dropout: float = 0.0):
        super().__init__()
        self.attention = _MultiHeadAttention(d_model, num_heads, dropout)

    def forward(self, x, lengths, out_dtype=None, x_bf16=None):
        return self.attention(x, lengths, out_dtype, x_bf16)


class LMUEncoder(nn.Module):
    """Constructor arguments as the reference's LMUEncoder (src/models/lmu_encoder.py:141-151);
    `use_fft_lmu` must be True."""

    def __init__(self, input_size=80, hidden_size=512, memory_size=256, num_lmu_layers=4, theta=128, dropout=0.1,
                 use_fft_lmu=True, seq_len=None, use_attention=True, num_attention_heads=8):
        super().__init__()
        if not use_fft_lmu:
            raise NotImplementedError("only the FFT-LMU encoder is implemented (use_fft_lmu=True)")
        if seq_len is None:
            raise ValueError("seq_len must be specified when using FFT-based LMU")
        self.hidden_size, self.use_attention = hidden_size, use_attention
        self.input_projection = nn.Linear(input_size, hidden_size)
        self.attention_layers = nn.ModuleList(
            [_SelfAttention(hidden_size, num_attention_heads, dropout) if use_attention else nn.Identity()
             for _ in range(num_lmu_layers)])
        self.lmu_layers = nn.ModuleList(
            [LMUFFT(hidden_size, hidden_size, memory_size, seq_len, theta) for _ in range(num_lmu_layers)])
        self.layer_norms = nn.ModuleList([nn.LayerNorm(hidden_size) for _ in range(num_lmu_layers)])
        self.output_projection = nn.Linear(hidden_size, hidden_size)
        self.dropout_layer = nn.Dropout(dropout)  # after every layer norm (src/models/lmu_encoder.py:232, 277); identity in eval
        # Under autocast torch's layer_norm returns fp32, and every consumer (three projections, the
        # LMU layer) casts that tensor to bf16 again.  With bf16_stream the layer norms write bf16
        # directly: the casts disappear and the residual is the bf16-rounded activation -- a
        # deviation from the reference's autocast arithmetic of one bf16 rounding per block, inside
        # the bf16 tolerance.  Off by default.
        self.bf16_stream = False

    def linear_groups(self, x: torch.Tensor) -> list:
        """The (weights, biases) groups of this encoder's training projections, as asr_amd.linear.prepare_weights
        takes them: the layers that _linear / fused_linear_train will run on the HIP kernels for an input like x."""
        def one(layer):
            return ([layer.weight], None if layer.bias is None else [layer.bias])

        groups = []
        if linear_train_supported(x, self.input_projection.weight):
            groups.append(one(self.input_projection))
        h = torch.empty((1, self.hidden_size), device=x.device, dtype=x.dtype)  # what the inner layers see
        if self.use_attention:
            for blk in self.attention_layers:
                a = blk.attention
                qkv = (a.w_q, a.w_k, a.w_v)
                if a.d_k not in SUPPORTED_HEAD_WIDTHS:
                    continue
                if all(l.bias is not None for l in qkv) and linear_train_supported(h, a.w_q.weight):
                    groups.append(([l.weight for l in qkv], [l.bias for l in qkv]))
                else:
                    groups += [one(l) for l in qkv if linear_train_supported(h, l.weight)]
                if linear_train_supported(h, a.w_o.weight):
                    groups.append(one(a.w_o))
        if linear_train_supported(h, self.output_projection.weight):
            groups.append(one(self.output_projection))
        return groups

    def forward(self, x: torch.Tensor, lengths: Optional[torch.Tensor] = None,
                prepared: bool = False) -> Tuple[torch.Tensor, list]:
        """`prepared`: the caller has run prepare_weights over linear_groups itself (with layers of its own)."""
        stream = torch.bfloat16 if (self.bf16_stream and x.is_cuda and torch.is_autocast_enabled()
                                    and torch.get_autocast_dtype("cuda") == torch.bfloat16) else None
        # The attention block's output has one consumer, the LMU layer, which under bf16 autocast rounds it to bf16
        # first: its layer norm writes that rounding directly -- the same numbers without the fp32 tensor and the cast
        # (and, in training, without the widening of the layer's input gradient).
        att_out = stream
        if att_out is None and x.is_cuda and torch.is_autocast_enabled() and torch.get_autocast_dtype("cuda") == torch.bfloat16:
            att_out = torch.bfloat16
        # In training the float32 output of a block has two readers in the next one: the residual connection, and
        # the q/k/v projections, which round it to bf16 first.  The layer norm writes both forms in its pass and
        # takes both gradients back in its backward; the last block's output has one reader (the output
        # projection) and is written in bf16 only.
        autocast16 = att_out == torch.bfloat16 and stream is None and torch.is_grad_enabled()
        if lengths is not None and x.is_cuda and self.use_attention:
            # the attention kernels read int32 lengths on the device: converted once, not once per block
            lengths = lengths.to(device=x.device, dtype=torch.int32).contiguous()
        if autocast16 and not prepared:
            prepare_weights(self.linear_groups(x))  # every projection's bf16 operands in one launch
        x = _linear(self.input_projection, x)
        states = []
        x16 = None
        n = len(self.lmu_layers)
        for i, (att, lmu, ln) in enumerate(zip(self.attention_layers, self.lmu_layers, self.layer_norms)):
            if self.use_attention:
                x = att(x, lengths, att_out, x16)
            x16 = None
            drop = self.training and self.dropout_layer.p > 0
            if not drop and ln.elementwise_affine and ln.bias is not None and lmu.supports_fused_layer_norm(x):
                # the layer norm rides on the projection kernel's epilogue: h never goes to HBM
                x, h_n = lmu.forward_layer_norm(
                    x, ln, stream or (torch.float32 if torch.is_autocast_enabled() else x.dtype))
            else:
                # training under bf16 autocast: the layer norm's backward reads h anyway and hands the LMUFFT layer
                # its gradient already multiplied by (h > 0); the layer's own masking pass is not run
                fold = (FOLD_RELU_BACKWARD and autocast16 and x.is_cuda
                        and _layer_norm_trains_on_hip(ln, self.hidden_size, torch.bfloat16))
                lmu.grad_is_masked = bool(fold)
                x, h_n = lmu(x)
                fold = fold and x.dtype == torch.bfloat16
                if autocast16 and i == n - 1:
                    x = _layer_norm(ln, x, None, torch.bfloat16, self.dropout_layer, relu_input=fold)
                elif autocast16 and self.use_attention:
                    x, x16 = _layer_norm(ln, x, None, None, self.dropout_layer, twin=True, relu_input=fold)
                else:
                    x = _layer_norm(ln, x, None, stream, self.dropout_layer, relu_input=fold)
            states.append(h_n)
        return _linear(self.output_projection, x), states

    def get_output_dim(self) -> int:
        return self.hidden_size


class LMUASRModel(nn.Module):
    """Encoder + CTC head with the reference model's state_dict layout (src/models/asr_model.py:53-128)."""

    def __init__(self, input_size, hidden_size, memory_size, num_lmu_layers, theta, seq_len, vocab_size,
                 num_attention_heads=8, use_attention=True, dropout=0.0):
        super().__init__()
        self.encoder = LMUEncoder(input_size, hidden_size, memory_size, num_lmu_layers, theta, dropout, True, seq_len,
                                  use_attention, num_attention_heads)
        self.decoder = CTCDecoder(hidden_size, vocab_size)
        self.vocab_size = vocab_size

    @torch.no_grad()
    def forward(self, spectrograms: torch.Tensor, input_lengths: Optional[torch.Tensor] = None):
        encoded, states = self.encoder(spectrograms, input_lengths)
        return self.decoder(encoded), states

    @torch.no_grad()
    def transcribe(self, spectrograms: torch.Tensor, input_lengths: Optional[torch.Tensor] = None) -> List[List[int]]:
        """Greedy CTC token lists (the reference's decode(log_probs, input_lengths, beam_width=1))."""
        encoded, _ = self.encoder(spectrograms, input_lengths)
        return self.decoder.decode(encoded, input_lengths)

    def bucket_groups(self) -> list:
        """Parameter groups for asr_amd.parallel.GradBuckets in the order the backward completes them (head first,
        input projection last).  Inside an attention block the q, k, v weights come first and back to back, then their
        biases: the fused projection's weight gradient is then one kernel writing straight into the bucket."""
        enc = self.encoder
        groups = [[*self.decoder.parameters(), *enc.output_projection.parameters()]]
        for i in reversed(range(len(enc.lmu_layers))):
            groups.append(list(enc.layer_norms[i].parameters()))
            groups.append(enc.lmu_layers[i])
            if enc.use_attention:
                a = enc.attention_layers[i].attention
                qkv = (a.w_q, a.w_k, a.w_v)
                groups.append([*(l.weight for l in qkv), *(l.bias for l in qkv if l.bias is not None),
                               *a.w_o.parameters(), *a.layer_norm.parameters()])
        groups.append(list(enc.input_projection.parameters()))
        return groups

    def log_probs_for_training(self, spectrograms: torch.Tensor, input_lengths: Optional[torch.Tensor] = None):
        """Log-probabilities with an autograd graph (the reference's LMUASRModel.forward in training,
        src/models/asr_model.py:110-128).  Under bf16 autocast on the GPU the LMUFFT layers, attention, layer norms
        and every Linear (the vocabulary projection included) run this package's forward and backward kernels;
        dropout after the layer norms, residual adds and the log-softmax are torch ops."""
        enc, proj = self.encoder, self.decoder.ctc_projection
        prepared = False
        if (spectrograms.is_cuda and torch.is_grad_enabled() and not enc.bf16_stream and torch.is_autocast_enabled()
                and torch.get_autocast_dtype("cuda") == torch.bfloat16):
            # the bf16 operands of every projection of the step, the vocabulary projection included, in one launch
            groups = enc.linear_groups(spectrograms)
            if linear_train_supported(torch.empty((1, enc.hidden_size), device=spectrograms.device), proj.weight):
                groups.append(([proj.weight], None if proj.bias is None else [proj.bias]))
            prepare_weights(groups)
            prepared = True
        encoded, _ = enc(spectrograms, input_lengths, prepared=prepared)
        if prepared and proj.weight.shape[0] <= 64 and linear_train_supported(encoded, proj.weight):
            # the projection's padded bf16 logits go straight into the log-softmax kernel (and its gradient straight
            # back): no slice, widening cast, pad or narrowing cast in between
            from .ctc import log_softmax_rows
            _took("hip", "linear")
            _took("hip", "log_softmax")
            return log_softmax_rows(linear_train(encoded, proj.weight, proj.bias, keep_padding=True), proj.weight.shape[0])
        logits = _linear(proj, encoded)
        return torch.log_softmax(logits.float(), dim=-1)

    def compute_loss(self, log_probs: torch.Tensor, targets: torch.Tensor, input_lengths: torch.Tensor,
                     target_lengths: torch.Tensor, total_targets: Optional[int] = None) -> torch.Tensor:
        """CTC loss as the reference builds it (src/models/asr_model.py:130-156; nn.CTCLoss(blank=vocab_size-1,
        reduction='mean', zero_infinity=True), :105): this package's CTC kernel on the GPU, otherwise torch's loss on
        labels packed on the device."""
        from .ctc import ctc_loss, pack_targets

        if log_probs.is_cuda and self.vocab_size <= 64 and log_probs.dtype == torch.float32:
            _took("hip", "ctc_loss")  # loss and gradient in one kernel; the padded labels are read in place
            return ctc_loss(log_probs, targets, input_lengths, target_lengths, self.vocab_size - 1, True)
        _took("torch", "ctc_loss")
        packed, _ = pack_targets(targets, target_lengths, total_targets)
        return torch.nn.functional.ctc_loss(log_probs.transpose(0, 1), packed, input_lengths, target_lengths,
                                            blank=self.vocab_size - 1, reduction="mean", zero_infinity=True)
