"""CPU oracle for the LMUFFT layer -- TEST INFRASTRUCTURE, NOT A PRODUCT PATH.

This file restates, in plain numpy, the arithmetic of the reference's parallel
Legendre Memory Unit layer so that the HIP kernels can be checked against it.
Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg of
`bench.py` may import it.  The product (`asr_amd`) never does: it fails loudly
when its HIP library is missing.

Pinning: every function here is checked (tests/test_oracle_golden.py) against
golden vectors that were produced by importing the reference's own
`src/lmu.py::LMUFFT` in the build container (tests/golden/make_golden.py).

Reference citations are relative to the reference checkout:
  * state-space construction ........ src/lmu.py:248-272
  * impulse response H, fft_H ....... src/lmu.py:275-287
  * forward (eq. 18, 26, 20) ........ src/lmu.py:297-320
  * backward ........................ no reference code; it is what autograd
                                      derives from src/lmu.py:300-318
"""
from __future__ import annotations

import numpy as np

__all__ = [
    "state_space_matrices",
    "impulse_response",
    "causal_conv_direct",
    "causal_conv_fft",
    "anticausal_corr_direct",
    "lmufft_forward",
    "lmufft_backward",
    "round_bf16",
]


# --------------------------------------------------------------------------
# construction
# --------------------------------------------------------------------------
def _expm(Mx: np.ndarray) -> np.ndarray:
    """Matrix exponential in fp64 (scaling-and-squaring with a Taylor core).

    scipy.signal.cont2discrete(method="zoh") -- a third-party routine the
    reference calls at src/lmu.py:262-266 (scipy pinned 1.10.* in
    pytorch_lmu_environment.yml:14) -- is published as: build the block matrix
    [[A, B], [0, 0]] * dt, exponentiate it, read A_d from the top-left block
    and B_d from the top-right block.  scipy itself is used when importable
    (it is in the build container and on the GPU box); the Taylor fallback
    keeps the oracle self-contained.
    """
    try:
        from scipy.linalg import expm  # same routine cont2discrete calls

        return expm(Mx)
    except Exception:  # pragma: no cover - scipy is present in this image
        n = Mx.shape[0]
        norm = np.linalg.norm(Mx, 1)
        s = max(0, int(np.ceil(np.log2(max(norm, 1e-300)))) + 4)
        Xs = Mx / (2.0 ** s)
        E = np.eye(n)
        term = np.eye(n)
        for k in range(1, 40):
            term = term @ Xs / k
            E = E + term
        for _ in range(s):
            E = E @ E
        return E


def state_space_matrices(memory_size: int, theta: float):
    """Discrete (ZOH, dt=1) Legendre delay system, fp64 -> fp32.

    Follows src/lmu.py:251-270: Q = 0..M-1, R = (2Q+1)/theta,
    A_c[i,j] = R_i * (-1 if i<j else (-1)^(i-j+1)), B_c = R * (-1)^Q,
    zero-order hold with dt = 1, then cast to float32.
    """
    M = int(memory_size)
    q = np.arange(M, dtype=np.float64)
    r = (2.0 * q + 1.0) / float(theta)
    i = q[:, None]
    j = q[None, :]
    sign = np.where(i < j, -1.0, (-1.0) ** (i - j + 1.0))
    A_c = r[:, None] * sign
    B_c = (r * (-1.0) ** q)[:, None]
    blk = np.zeros((M + 1, M + 1), dtype=np.float64)
    blk[:M, :M] = A_c
    blk[:M, M:] = B_c
    E = _expm(blk)  # dt = 1
    A_d = E[:M, :M]
    B_d = E[:M, M:]
    return A_d.astype(np.float32), B_d.astype(np.float32)


def impulse_response(A: np.ndarray, B: np.ndarray, seq_len: int) -> np.ndarray:
    """H[:, t] = A^t B by float32 repeated multiplication (src/lmu.py:278-284).

    The reference iterates `A_i <- A @ A_i` in float32 starting from the
    identity and stores `A_i @ B` per step; the same order is kept here
    because an fp64-exact A^t B differs from it by ~8e-6 relative.
    """
    A = np.asarray(A, dtype=np.float32)
    B = np.asarray(B, dtype=np.float32)
    M = A.shape[0]
    H = np.empty((M, seq_len), dtype=np.float32)
    A_i = np.eye(M, dtype=np.float32)
    for t in range(seq_len):
        H[:, t] = (A_i @ B)[:, 0]
        A_i = A @ A_i
    return H


def fft_of_impulse(H: np.ndarray) -> np.ndarray:
    """fft_H = rfft(H, n = 2*seq_len) as complex64 (src/lmu.py:285)."""
    T = H.shape[1]
    return np.fft.rfft(H.astype(np.float32), n=2 * T, axis=-1).astype(np.complex64)


# --------------------------------------------------------------------------
# the causal convolution (eq. 26), two independent routes
# --------------------------------------------------------------------------
def causal_conv_direct(u: np.ndarray, H: np.ndarray, dtype=np.float64) -> np.ndarray:
    """m[b,t,k] = sum_{s<=t} H[k,t-s] * u[b,s], evaluated as a Toeplitz product.

    This is what src/lmu.py:303-312 computes (linear convolution truncated to
    the first T samples); written as the time-domain sum it has no FFT
    round-off, so in fp64 it is the ground truth for both kernel designs.
    """
    u = np.asarray(u, dtype=dtype)
    Hd = np.asarray(H, dtype=dtype)
    B, T = u.shape
    M = Hd.shape[0]
    assert Hd.shape[1] >= T
    # Toeplitz index matrix: lag[t, s] = t - s (masked where negative)
    t = np.arange(T)
    lag = t[:, None] - t[None, :]
    mask = lag >= 0
    lag = np.where(mask, lag, 0)
    m = np.empty((B, T, M), dtype=dtype)
    for k in range(M):
        Tk = np.where(mask, Hd[k, :T][lag], 0.0)  # [T(t), T(s)]
        m[:, :, k] = u @ Tk.T
    return m


def causal_conv_fft(u: np.ndarray, H: np.ndarray) -> np.ndarray:
    """Same result through rfft(n=2T) * rfft(H) -> irfft, as src/lmu.py:303-312."""
    u = np.asarray(u)
    B, T = u.shape
    fH = np.fft.rfft(np.asarray(H)[:, :T], n=2 * T, axis=-1)  # [M, T+1]
    fu = np.fft.rfft(u, n=2 * T, axis=-1)  # [B, T+1]
    m = np.fft.irfft(fu[:, None, :] * fH[None, :, :], n=2 * T, axis=-1)[:, :, :T]
    return np.ascontiguousarray(np.transpose(m, (0, 2, 1))).astype(u.dtype, copy=False)


def anticausal_corr_direct(gm: np.ndarray, H: np.ndarray, dtype=np.float64) -> np.ndarray:
    """gu[b,s] = sum_{t>=s} sum_k H[k,t-s] * gm[b,t,k] (adjoint of the conv)."""
    gm = np.asarray(gm, dtype=dtype)
    Hd = np.asarray(H, dtype=dtype)
    B, T, M = gm.shape
    t = np.arange(T)
    lag = t[:, None] - t[None, :]  # [t, s]
    mask = lag >= 0
    lag = np.where(mask, lag, 0)
    gu = np.zeros((B, T), dtype=dtype)
    for k in range(M):
        Tk = np.where(mask, Hd[k, :T][lag], 0.0)  # [t, s]
        gu += gm[:, :, k] @ Tk
    return gu


def anticausal_corr_fft(gm: np.ndarray, H: np.ndarray) -> np.ndarray:
    """The adjoint through FFTs (correlation = convolution with the time-reversed filter); in fp64
    it agrees with anticausal_corr_direct to ~1e-13 of the terms' norm and is what the tests use
    for sequences too long for the O(T^2 M) direct sum."""
    gm = np.asarray(gm, dtype=np.float64)
    B, T, M = gm.shape
    fH = np.fft.rfft(np.asarray(H, dtype=np.float64)[:, :T], n=2 * T, axis=-1)  # [M, T+1]
    fg = np.fft.rfft(np.transpose(gm, (0, 2, 1)), n=2 * T, axis=-1)  # [B, M, T+1]
    full = np.fft.irfft((fg * np.conj(fH)[None]).sum(axis=1), n=2 * T, axis=-1)  # [B, 2T]
    return np.ascontiguousarray(full[:, :T])


# --------------------------------------------------------------------------
# full layer
# --------------------------------------------------------------------------
def lmufft_forward(x, w_u, b_u, W_h, b_h, H, dtype=np.float64, conv="direct"):
    """Forward of the layer (src/lmu.py:297-320).

    x [B,T,D]; w_u [1,D]; b_u [1]; W_h [Hd, M+D] (memory columns first, then
    input columns -- torch.cat((m, x)) at src/lmu.py:315); b_h [Hd]; H [M,>=T].
    Returns dict(u, m, h, h_n).
    """
    x = np.asarray(x, dtype=dtype)
    w_u = np.asarray(w_u, dtype=dtype).reshape(-1)
    b_u = float(np.asarray(b_u, dtype=dtype).reshape(-1)[0])
    W_h = np.asarray(W_h, dtype=dtype)
    b_h = np.asarray(b_h, dtype=dtype)
    B, T, D = x.shape
    u = np.maximum(x @ w_u + b_u, 0.0)  # [B,T]           eq. 18
    if conv == "direct":
        m = causal_conv_direct(u, H, dtype=dtype)  # [B,T,M]  eq. 26
    else:
        m = causal_conv_fft(u, np.asarray(H, dtype=dtype))
    M = m.shape[2]
    z = m @ W_h[:, :M].T + x @ W_h[:, M:].T + b_h  # eq. 20
    h = np.maximum(z, 0.0)
    return {"u": u, "m": m, "h": h, "h_n": h[:, -1, :].copy()}


def lmufft_backward(x, w_u, b_u, W_h, b_h, H, gh, gh_n=None, dtype=np.float64, h_mask=None, u_mask=None,
                    conv="direct"):
    """Hand-derived backward of the layer.

    `conv` = "direct" (time-domain sums, the definition) or "fft" (fp64 FFTs, for long sequences).
    `h_mask` / `u_mask` (boolean, optional) replace the activation patterns
    (z > 0) and (a > 0).  ReLU's derivative jumps at 0, so a reduced-precision
    forward whose pre-activation differs from the fp64 one by rounding noise
    takes the other branch for the few elements with |z| below that noise; a
    backward kernel is then checked against the oracle evaluated with the
    kernel's OWN forward pattern.

    g = gh (+ gh_n on the last frame); gz = g*(z>0); gW_h = gz^T [m;x];
    gb_h = sum gz; [gm; gx1] = gz W_h; gu = adjoint conv of gm;
    ga = gu*(a>0); gw_u = sum ga*x; gb_u = sum ga; gx = gx1 + ga*w_u.
    """
    f = lmufft_forward(x, w_u, b_u, W_h, b_h, H, dtype=dtype, conv=conv)
    x = np.asarray(x, dtype=dtype)
    w_u1 = np.asarray(w_u, dtype=dtype).reshape(-1)
    W_h = np.asarray(W_h, dtype=dtype)
    B, T, D = x.shape
    M = f["m"].shape[2]
    g = np.array(gh, dtype=dtype, copy=True)
    if gh_n is not None:
        g[:, -1, :] += np.asarray(gh_n, dtype=dtype)
    gz = g * ((f["h"] > 0) if h_mask is None else np.asarray(h_mask, dtype=bool))
    mx = np.concatenate([f["m"], x], axis=-1)  # [B,T,M+D]
    gW_h = np.einsum("btn,btj->nj", gz, mx)
    gb_h = gz.sum(axis=(0, 1))
    gmx = gz @ W_h  # [B,T,M+D]
    gm = gmx[:, :, :M]
    gx = gmx[:, :, M:].copy()
    gu = anticausal_corr_direct(gm, H, dtype=dtype) if conv == "direct" else anticausal_corr_fft(gm, H)
    ga = gu * ((f["u"] > 0) if u_mask is None else np.asarray(u_mask, dtype=bool))
    gw_u = np.einsum("bt,btd->d", ga, x)[None, :]
    gb_u = np.array([ga.sum()], dtype=dtype)
    gx += ga[:, :, None] * w_u1[None, None, :]
    return {
        "gx": gx,
        "gw_u": gw_u,
        "gb_u": gb_u,
        "gW_h": gW_h,
        "gb_h": gb_h,
        "gm": gm,
        "gu": gu,
        "gz": gz,
        "ga": ga,
        **{k: f[k] for k in ("u", "m", "h", "h_n")},
    }


def round_bf16(a: np.ndarray) -> np.ndarray:
    """Round-to-nearest-even fp32 -> bf16 -> fp32 (finite inputs)."""
    a32 = np.ascontiguousarray(a, dtype=np.float32)
    bits = a32.view(np.uint32)
    rounded = (bits + 0x7FFF + ((bits >> 16) & 1)) & 0xFFFF0000
    return rounded.astype(np.uint32).view(np.float32).reshape(a32.shape)
