import sys, os, numpy as np
sys.argv=[sys.argv[0]]
sys.path.insert(0,'/root/repo/tests/golden')
import make_golden as mg, torch
torch.set_num_threads(8)
from oracle import oracle
oracle.build()
z=np.load("scratch/sub.npz"); P=10
sub={"flows":[[z["flows"][t]] for t in range(P)],"ev":[z["ev"][t] for t in range(P)],"pm":[z["pm"][t] for t in range(P)],
     "dev":[np.zeros((1,0,4),np.float32)]*P,"dpm":[np.zeros((1,0,2),np.float32)]*P}
os.environ["TEF_DUMP_G"]="scratch/dump.bin"
w=oracle.Window(sub["flows"],sub["ev"],sub["pm"],sub["dev"],sub["dpm"],S=1,mode="two")
l,d=w.loss("Iterative",None,None)
raw=np.fromfile("scratch/dump.bin",dtype=np.float32); hdr=raw[:4].view(np.int32); Pp,M,Mt=hdr[:3]
o=4; n=(P+1)*(M+1); gy=raw[o:o+n][: (P+1)*M].reshape(P+1,M); o+=n; gx=raw[o:o+n][:(P+1)*M].reshape(P+1,M); o+=n
n2=(P+1)*(Mt+1); ty=raw[o:o+n2][:(P+1)*Mt].reshape(P+1,Mt); o+=n2; tx=raw[o:o+n2][:(P+1)*Mt].reshape(P+1,Mt); o+=n2
kb=raw[o:o+Mt].view(np.int32); o+=Mt+1; kf=raw[o:o+Mt].view(np.int32)
# reference with hooks
from loss.flow import Iterative
rec=[]
orig=Iterative.iwe_formatting
def patched(self, warped_events, pol_mask, ts_list, tref, ts_scaling, **kw):
    if warped_events.requires_grad:
        warped_events.retain_grad(); rec.append((tref, warped_events))
    return orig(self, warped_events, pol_mask, ts_list, tref, ts_scaling, **kw)
Iterative.iwe_formatting=patched
cfg=mg.make_config(128,128,1,10,1,"two")
l64,l32,g=mg.run_loss("Iterative",cfg,sub)
print(l64,l, "M",M)
N=10000
delta=5
worst=[]
for tref,we in rec:
    lo=max(0,tref-delta); hi=min(P,tref+delta)
    gr=we.grad.numpy()[0]    # [(hi-lo)*N, 2] (y,x)
    pos=we.detach().numpy()[0]
    for t in range(lo,hi):
        sl=slice(t*N,(t+1)*N); rs=slice((t-lo)*N,(t-lo+1)*N)
        dy=np.abs(gr[rs,0]-gy[tref,sl]); dx=np.abs(gr[rs,1]-gx[tref,sl])
        dp=np.abs(pos[rs,0]-ty[tref,sl]*( (kb[sl]<tref)&(tref<kf[sl]) ))
        k=np.argmax(dy+dx)
        worst.append((float(dy[k]+dx[k]),tref,t,int(k),gr[rs][k],gy[tref,sl][k],gx[tref,sl][k],pos[rs][k],ty[tref,sl][k],tx[tref,sl][k], float(dp.max())))
worst.sort(key=lambda r:-r[0])
for r in worst[:8]: print(r)
