import sys, numpy as np, torch
sys.path.insert(0, '.')
from oracle import lmufft_oracle as orc
import asr_amd.functional as F
from asr_amd import lmufft_layer
F._DEBUG = {}
dev = torch.device('cuda:0')
def bf(a): return orc.round_bf16(np.asarray(a, dtype=np.float32))
B,T,D,M,Hd,ldH = 2,100,80,128,256,100
rng = np.random.default_rng(1)
x = bf(rng.standard_normal((B,T,D))); w_u = bf(rng.standard_normal((1,D))/np.sqrt(D)); b_u = bf(np.array([0.05]))
W_h = bf(rng.standard_normal((Hd,M+D))/np.sqrt(M+D)); b_h = bf(0.1*rng.standard_normal(Hd))
t = np.arange(ldH); H = bf((rng.standard_normal((M,ldH))*np.exp(-t/(ldH/3))[None,:]))
gh = bf(rng.standard_normal((B,T,Hd)))
ref = orc.lmufft_backward(x,w_u,b_u,W_h,b_h,H[:,:T],gh,None)
t32 = lambda a: torch.from_numpy(a).to(dev)
tx = t32(x).to(torch.bfloat16).requires_grad_(True)
params = [t32(a).requires_grad_(True) for a in (w_u,b_u,W_h,b_h)]
h,h_n = lmufft_layer(tx,*params,t32(H))
h.backward(t32(gh).to(torch.bfloat16))
def rel(got, want, name):
    got = got.detach().float().cpu().numpy().astype(np.float64); want=np.asarray(want)
    e = got-want
    print(f"{name:6s} rms_rel={np.sqrt((e**2).mean())/np.sqrt((want**2).mean()):.3e} maxabs={np.abs(e).max():.3e} refmax={np.abs(want).max():.3e}")
    return e
rel(h, ref['h'], 'h'); rel(F._DEBUG['m'], ref['m'], 'm'); rel(F._DEBUG['u'], ref['u'], 'u')
e = rel(F._DEBUG['gm'], ref['gm'], 'gm')
print(' gm err per col block', [float(np.sqrt((e[:,:,i:i+32]**2).mean())) for i in range(0,M,32)])
print(' gm err per row block', [float(np.sqrt((e[:,i:i+16,:]**2).mean())) for i in range(0,T,16)])
e = rel(tx.grad, ref['gx'], 'gx')
print(' gx err per col block', [float(np.sqrt((e[:,:,i:i+16]**2).mean())) for i in range(0,D,16)])
print(' gx err per row block', [float(np.sqrt((e[:,i:i+10,:]**2).mean())) for i in range(0,T,10)])
# gx without the ga term
gx1 = ref['gz'] @ W_h[:, M:].astype(np.float64)
rel(tx.grad, gx1, 'gx1?')
for p_, k in zip(params, ('gw_u','gb_u','gW_h','gb_h')): rel(p_.grad, ref[k], k)
# ga implied: (gx - gx1) . w_u / |w_u|^2
got = tx.grad.detach().float().cpu().numpy().astype(np.float64)
wu = w_u.reshape(-1).astype(np.float64)
ga_got = ((got-gx1) @ wu)/ (wu@wu)
print('ga got', ga_got[0,:8]); print('ga ref', ref['ga'][0,:8])
print('ga got', ga_got[0,-8:]); print('ga ref', ref['ga'][0,-8:])
