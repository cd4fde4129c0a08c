"""Reading checkpoints written by the reference's trainer into the modules of this package.

Format (reference: src/training/utils.py:313-345): torch.save of
{'epoch', 'model_state_dict', 'optimizer_state_dict', 'loss', 'metrics'}; the model is the
reference's LMUASRModel, whose LMU layers live under `encoder.lmu_layers.{i}.` and whose CTC
projection under `decoder.ctc_projection.` (src/models/asr_model.py:69-102,
src/models/lmu_encoder.py:206-212).  The file is read with `weights_only=True`: nothing in it is
executed (the reference's own loader unpickles arbitrary objects, src/training/utils.py:361).
"""
from __future__ import annotations

import re
from typing import Dict, List, Optional

import torch

from .ctc import CTCDecoder
from .lmu import LMUFFT

LMU_PREFIX = "encoder.lmu_layers."
CTC_PREFIX = "decoder."


def read_checkpoint(path: str) -> Dict:
    """The checkpoint dictionary; a bare state_dict is wrapped so that callers see one shape."""
    ckpt = torch.load(path, map_location="cpu", weights_only=True)
    if not isinstance(ckpt, dict):
        raise RuntimeError(f"{path}: expected a dictionary, found {type(ckpt).__name__}")
    if "model_state_dict" not in ckpt:
        ckpt = {"model_state_dict": ckpt}
    # the reference's single-GPU trainer saves the model as it holds it (src/training/trainer.py:294),
    # possibly wrapped in DataParallel / DistributedDataParallel: every key then starts with "module."
    sd = ckpt["model_state_dict"]
    if sd and all(k.startswith("module.") for k in sd):
        ckpt = dict(ckpt)
        ckpt["model_state_dict"] = {k[len("module."):]: v for k, v in sd.items()}
    return ckpt


def lmu_layers_from_checkpoint(ckpt: Dict, conv_algo: str = "auto", theta: Optional[float] = None) -> List[LMUFFT]:
    """One LMUFFT per `encoder.lmu_layers.{i}` of the checkpoint, sizes read from the tensors.

    theta is not stored in a state_dict; it only matters for rebuilding A, B, H, which are loaded
    from the checkpoint anyway (pass it to have the attribute set)."""
    sd = ckpt["model_state_dict"]
    idx = sorted({int(m.group(1)) for k in sd for m in [re.match(re.escape(LMU_PREFIX) + r"(\d+)\.", k)] if m})
    if not idx:
        raise RuntimeError("no `encoder.lmu_layers.*` entries: not a checkpoint of the FFT-LMU model")
    layers = []
    for i in idx:
        pre = f"{LMU_PREFIX}{i}."
        if pre + "fft_H" not in sd:
            raise RuntimeError(f"layer {i} has no `fft_H`: the checkpoint was trained with the sequential LMU "
                               "(use_fft_lmu=False), which this package does not implement")
        W_h, H = sd[pre + "W_h.weight"], sd[pre + "H"]
        hidden, mem_plus_in = W_h.shape
        memory, seq_len = H.shape
        layer = LMUFFT(mem_plus_in - memory, hidden, memory, seq_len, theta if theta is not None else float(seq_len),
                       conv_algo=conv_algo)
        layer.load_state_dict({k[len(pre):]: v for k, v in sd.items() if k.startswith(pre)})
        if theta is not None:
            layer.theta = theta
        layers.append(layer)
    return layers


def ctc_decoder_from_checkpoint(ckpt: Dict) -> CTCDecoder:
    sd = ckpt["model_state_dict"]
    W = sd[CTC_PREFIX + "ctc_projection.weight"]
    dec = CTCDecoder(W.shape[1], W.shape[0])
    dec.load_state_dict({k[len(CTC_PREFIX):]: v for k, v in sd.items() if k.startswith(CTC_PREFIX)})
    return dec
