import sys, numpy as np, torch
sys.path.insert(0,'/root/reference'); sys.path.insert(0,'/root/repo')
from utils.iwe import get_event_flow
from oracle import oracle
rng=np.random.default_rng(5)
H=W=128; N=200000
fx=rng.standard_normal((1,1,H,W)).astype(np.float32)*2; fy=rng.standard_normal((1,1,H,W)).astype(np.float32)*2
loc=(rng.random((1,N,2))*(H-1)).astype(np.float32)
for nt in (1,8):
    torch.set_num_threads(nt)
    r=get_event_flow(torch.tensor(fx[:,0]),torch.tensor(fy[:,0]),torch.tensor(loc)).numpy()
    o=oracle.get_event_flow(fx[:,0],fy[:,0],loc)
    o=o[0] if isinstance(o,tuple) else o
    d=(r!=o)
    print(nt,"mismatch frac",d.mean(), "max ulp-ish", np.abs(r-o).max())
    k=np.argmax(np.abs(r-o).sum(-1)[0]); print(loc[0,k], r[0,k], o[0,k])
