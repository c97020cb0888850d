import sys, os, numpy as np
sys.argv=[sys.argv[0]]
sys.path.insert(0,'/root/repo/tests/golden')
import make_golden as mg, torch
torch.set_num_threads(8)
from taming_event_flow_amd import synth
from oracle import oracle
seed, i, b = 1, 3, 5
rng=np.random.default_rng(seed)
win=synth.make_window(rng,8,128,128,10,4,10000,0,sigma=2.0,kind="smooth")
P=10
sub={"flows":[[win["flows"][t][i][b:b+1]] for t in range(P)],"ev":[win["ev"][t][b:b+1] for t in range(P)],"pm":[win["pm"][t][b:b+1] for t in range(P)],
     "dev":[win["dev"][t][b:b+1] for t in range(P)],"dpm":[win["dpm"][t][b:b+1] for t in range(P)]}
np.savez("scratch/sub.npz", flows=np.array([s[0] for s in sub["flows"]]), ev=np.array(sub["ev"]), pm=np.array(sub["pm"]))
cfg=mg.make_config(128,128,1,10,1,"two")
l64,l32,g=mg.run_loss("Iterative",cfg,sub)
w=oracle.Window(sub["flows"],sub["ev"],sub["pm"],sub["dev"],sub["dpm"],S=1,mode="two")
l,d=w.loss("Iterative",None,None)
print(l64,l)
diff=np.abs(g.astype(np.float64)-d); gm=np.abs(g).max()
idx=np.argsort(diff.ravel())[::-1][:8]
for k in idx:
    u=np.unravel_index(k,diff.shape); print(u,diff[u]/gm,d[u],g[u])
np.save("scratch/sub_ref_g.npy",g); np.save("scratch/sub_or_g.npy",d)
