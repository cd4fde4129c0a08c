"""Error of the fp32 layer against the reference's goldens (cfg1a, cfg1c): max |got - ref| / max |ref| per tensor.
Run once per library build (LMUFFT_LIBRARY) to compare the fp32 GEMM forms."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import numpy as np
import torch
from asr_amd.functional import lmufft_layer

dev = torch.device("cuda:0")
for name in ("cfg1a_f32", "cfg1c_f32"):
    g = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", name + ".npz"))
    t = lambda k: torch.from_numpy(g[k]).to(dev)
    x = t("x").requires_grad_(True)
    ps = [t(k).requires_grad_(True) for k in ("w_u", "b_u", "W_h", "b_h")]
    h, h_n = lmufft_layer(x, *ps, t("H"), conv_algo="toeplitz")
    torch.autograd.backward([h, h_n], [t("gh"), t("gh_n")])
    got = {"h": h, "h_n": h_n, "gx": x.grad, "gw_u": ps[0].grad, "gb_u": ps[1].grad, "gW_h": ps[2].grad, "gb_h": ps[3].grad}
    errs = {k: float((v.detach().double().cpu() - torch.from_numpy(g[k]).double()).abs().max() / max(np.abs(g[k]).max(), 1e-30))
            for k, v in got.items()}
    print(name, {k: f"{e:.2e}" for k, e in errs.items()})
