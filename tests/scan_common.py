"""Helpers of the state-space ("scan") convolution tests: the layout of the constants block that
lmufft_scanprep_build_host writes (include/lmufft.h) and a numpy evaluation of the route from that
block, step for step as the kernels do it."""
import numpy as np
import torch


def _al256(n):
    return (n + 255) & ~255


def scan_layout(M, L):
    Kp = (M + L + 63) // 64 * 64
    N = L * M
    off, out = 0, {}
    for name, nbytes in (("WcT", 2 * N * Kp), ("Wc", 2 * Kp * N), ("AL", 4 * M * M), ("ALT", 4 * M * M),
                         ("Hs", 4 * L * M), ("HsT", 4 * M * L)):
        out[name] = off
        off += _al256(nbytes)
    out["total"], out["Kp"], out["N"] = off, Kp, N
    return out


def _bf16_to_f64(raw_u16):
    return (raw_u16.astype(np.uint32) << 16).view(np.float32).astype(np.float64)


def scan_sections(block_u8, M, L):
    """The six matrices of a constants block (numpy uint8 array) as float64 arrays."""
    lay = scan_layout(M, L)
    Kp, N = lay["Kp"], lay["N"]
    assert block_u8.size == lay["total"]
    b = block_u8
    sec = {
        "WcT": _bf16_to_f64(b[lay["WcT"]:lay["WcT"] + 2 * N * Kp].view(np.uint16)).reshape(N, Kp),
        "Wc": _bf16_to_f64(b[lay["Wc"]:lay["Wc"] + 2 * Kp * N].view(np.uint16)).reshape(Kp, N),
        "AL": b[lay["AL"]:lay["AL"] + 4 * M * M].view(np.float32).astype(np.float64).reshape(M, M),
        "ALT": b[lay["ALT"]:lay["ALT"] + 4 * M * M].view(np.float32).astype(np.float64).reshape(M, M),
        "Hs": b[lay["Hs"]:lay["Hs"] + 4 * L * M].view(np.float32).astype(np.float64).reshape(L, M),
        "HsT": b[lay["HsT"]:lay["HsT"] + 4 * M * L].view(np.float32).astype(np.float64).reshape(M, L),
    }
    sec["Kp"], sec["N"] = Kp, N
    return sec


def round_bf16(a):
    return torch.from_numpy(np.asarray(a, np.float32)).to(torch.bfloat16).to(torch.float64).numpy()


def scan_forward_numpy(u, sec, M, L):
    """m [B, T, M] from u [B, T]: chunk states in float64, GEMM operands rounded to bf16 as the kernels round them."""
    B, T = u.shape
    nc = (T + L - 1) // L
    Kp = sec["Kp"]
    m = np.zeros((B, nc * L, M))
    for b in range(B):
        X = np.zeros(M)
        rows = np.zeros((nc, Kp))
        for c in range(nc):
            n = min(L, T - c * L)
            rows[c, :M] = X
            rows[c, M:M + n] = u[b, c * L:c * L + n]
            if n == L:
                X = sec["AL"] @ X + sec["Hs"].T @ u[b, c * L:(c + 1) * L]
        m[b] = (round_bf16(rows) @ sec["Wc"]).reshape(nc * L, M)
    return m[:, :T]


def scan_adjoint_numpy(gm, sec, M, L):
    """gu [B, T] from gm [B, T, M] (already bf16-representable values)."""
    B, T, _ = gm.shape
    nc = (T + L - 1) // L
    gu = np.zeros((B, nc * L))
    for b in range(B):
        pad = np.zeros((nc * L, M))
        pad[:T] = gm[b]
        R = pad.reshape(nc, L * M) @ sec["Wc"].T
        lam = np.zeros(M)
        for c in range(nc - 1, -1, -1):
            gu[b, c * L:(c + 1) * L] = R[c, M:M + L] + sec["HsT"].T @ lam
            lam = R[c, :M] + sec["ALT"] @ lam
    return gu[:, :T]
