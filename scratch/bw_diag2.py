import numpy as np
for seed in (0,1):
    r=np.load(f"scratch/ref_full_{seed}.npy").astype(np.float64); o=np.load(f"scratch/bench_window_{seed}_oracle_d.npy").astype(np.float64)
    d=np.abs(r-o); gm=np.abs(r).max()
    print(seed,"full max rel",d.max()/gm, "count >1e-4:",(d>1e-4*gm).sum(),"count>1e-5",(d>1e-5*gm).sum(), "of", d.size)
    idx=np.argsort(d.ravel())[::-1][:12]
    for i in idx:
        u=np.unravel_index(i,d.shape); print(u, d[u]/gm, o[u], r[u])
