"""CPU-side checks of the drop-in boundary: the shared library loads, exports
every symbol include/lmufft.h declares, and the Python module keeps the
reference's surface (names, buffers, state_dict layout).  No kernel runs here."""
import ctypes
import json
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "lmufft.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lmufft_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from asr_amd import _lib

    declared = _declared_symbols()
    assert declared, "no declarations parsed from include/lmufft.h"
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(raw, name), f"{name} is declared in lmufft.h but not exported"
    assert sorted(_lib.EXPORTS) == declared
    assert _lib.lib.lmufft_abi_version() == _lib.ABI_VERSION


def test_workspace_queries_and_shape_errors():
    from asr_amd import _lib

    s = _lib.Shape(64, 400, 384, 192, 384, 400)
    assert _lib.lib.lmufft_backward_workspace_bytes(ctypes.byref(s), _lib.F32) > 0
    assert _lib.lib.lmufft_forward_workspace_bytes(ctypes.byref(s), _lib.F32) > 0
    bad = _lib.Shape(2, 500, 8, 6, 8, 400)  # T > seq_len
    rc = _lib.lib.lmufft_forward_f32(ctypes.byref(bad), *([None] * 11), 0, None, 0, None)
    assert rc != 0
    assert b"exceeds" in _lib.lib.lmufft_last_error()
    with pytest.raises(RuntimeError):
        _lib.check(rc, "forward")


def test_batched_entry_points_validate_their_tables():
    """The table-driven entry points refuse bad tables before anything is launched (so this runs without a GPU)."""
    from asr_amd import _lib

    lib = _lib.lib
    s = _lib.Shape(64, 400, 384, 192, 384, 400)
    assert lib.lmufft_qt_bytes(ctypes.byref(s)) >= 448 * 384 * 2
    assert lib.lmufft_qt_bytes(ctypes.byref(_lib.Shape(2, 600, 384, 192, 384, 600))) == 0   # beyond 512 frames
    assert lib.lmufft_qt_bytes(ctypes.byref(_lib.Shape(2, 400, 384, 20, 384, 400))) == 0    # memory size not 8 k
    assert lib.lmufft_qprep_many_bf16(None, 0, None) == 0
    assert lib.lmufft_qprep_many_bf16(None, 2, None) != 0
    items = (_lib.QItem * 1)(_lib.QItem(s, None, None, None))
    assert lib.lmufft_qprep_many_bf16(items, 1, None) != 0 and b"hprep" in lib.lmufft_last_error()
    items = (_lib.QItem * 1)(_lib.QItem(_lib.Shape(2, 600, 384, 192, 384, 600), 16, 16, 16))
    assert lib.lmufft_qprep_many_bf16(items, 1, None) != 0 and b"route" in lib.lmufft_last_error()
    assert lib.lmufft_backward_weights_sum_many_bf16(None, 0, None) == 0
    assert lib.lmufft_backward_weights_sum_many_bf16(None, -1, None) != 0
    sums = (_lib.LmuSumItem * 1)(_lib.LmuSumItem(s, 16, 16, 16, 16, 16, 8))
    assert lib.lmufft_backward_weights_sum_many_bf16(sums, 1, None) != 0 and b"workspace" in lib.lmufft_last_error()
    rc = lib.lmufft_backward_flags_bf16(ctypes.byref(s), *([None] * 17), 0, None, 0, None, 0x80, None)
    assert rc != 0 and b"unknown flags" in lib.lmufft_last_error()


def test_module_surface_matches_reference(golden_dir):
    from asr_amd import LMUFFT

    layer = LMUFFT(80, 384, 192, 400, 1000)
    man = json.load(open(os.path.join(golden_dir, "state_dict_manifest_cfg1a.json")))
    sd = layer.state_dict()
    assert list(sd.keys()) == ["A", "B", "H", "fft_H", "W_u.weight", "W_u.bias", "W_h.weight", "W_h.bias"]
    for k, v in sd.items():
        assert list(v.shape) == man[k]["shape"], k
        assert str(v.dtype) == man[k]["dtype"], k
    assert (layer.hidden_size, layer.memory_size, layer.seq_len, layer.theta) == (384, 192, 400, 1000)
    assert isinstance(layer.W_u, torch.nn.Linear) and isinstance(layer.W_h, torch.nn.Linear)
    assert isinstance(layer.f_u, torch.nn.ReLU) and isinstance(layer.f_h, torch.nn.ReLU)
    # constructor buffers agree with the reference's
    b = np.load(os.path.join(golden_dir, "buffers_theta1000_M192.npz"))
    np.testing.assert_allclose(sd["A"].numpy(), b["A"], rtol=1e-6, atol=1e-8)
    np.testing.assert_allclose(sd["B"].numpy(), b["B"], rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(sd["H"].numpy(), b["H400"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(sd["fft_H"].numpy(), b["fft_H400"], rtol=1e-3, atol=1e-5)


def test_no_cpu_fallback():
    from asr_amd import LMUFFT, LMU

    layer = LMUFFT(12, 10, 6, 16, 8)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        layer(torch.randn(2, 16, 12))
    with pytest.raises(NotImplementedError):
        LMU(12, 10, 6, 8)


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "asr_amd")):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src, f"{f} refers to the oracle"
