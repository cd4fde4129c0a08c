"""MI355X-native implementation of the FFT-LMU hot path (LMUFFT layer)."""
from .lmu import LMU, LMUFFT  # noqa: F401
from .functional import (causal_conv, causal_conv_adjoint, lmufft_layer,  # noqa: F401
                         prepare_impulse_bf16, prepare_impulse_fft)

from .ctc import CTCDecoder, pack_targets  # noqa: F401
from .optim import FlatAdamW, LearningRateScheduler  # noqa: F401
from .frontend import MelFrontend  # noqa: F401

__all__ = ["MelFrontend", "CTCDecoder", "pack_targets", "FlatAdamW", "LearningRateScheduler", "LMUFFT", "LMU", "lmufft_layer", "causal_conv", "causal_conv_adjoint", "prepare_impulse_bf16",
           "prepare_impulse_fft"]
