import sys, numpy as np, torch
sys.path.insert(0,'/root/reference')
from utils.iwe import get_event_flow
rng=np.random.default_rng(5)
H=W=128; N=200000
fx=rng.standard_normal((1,H,W)).astype(np.float32)*2; fy=rng.standard_normal((1,H,W)).astype(np.float32)*2
loc=(rng.random((1,N,2))*(H-1)).astype(np.float32)
r=get_event_flow(torch.tensor(fx),torch.tensor(fy),torch.tensor(loc)).numpy()[0]
f32=np.float32
def fma(a,b,c): return (a.astype(np.float64)*b.astype(np.float64)+c.astype(np.float64)).astype(f32)
y=loc[0,:,0]; x=loc[0,:,1]
gy=(f32(2)*y/f32(H-1)-f32(1)).astype(f32); gx=(f32(2)*x/f32(W-1)-f32(1)).astype(f32)
def variants(g, size):
    out={}
    out["mulhalf"]=((g+f32(1))*f32((size-1)/2)).astype(f32)
    out["div2"]=(((g+f32(1))/f32(2))*f32(size-1)).astype(f32)
    return out
for un in ("mulhalf","div2"):
    iy=variants(gy,H)[un]; ix=variants(gx,W)[un]
    x0=np.floor(ix); y0=np.floor(iy)
    w=(ix-x0).astype(f32); e=(f32(1)-w).astype(f32); n=(iy-y0).astype(f32); s=(f32(1)-n).astype(f32)
    nw=(s*e).astype(f32); ne=(s*w).astype(f32); sw=(n*e).astype(f32); se=(n*w).astype(f32)
    xi=x0.astype(int); yi=y0.astype(int)
    def tap(m,yy,xx):
        ok=(yy>=0)&(yy<H)&(xx>=0)&(xx<W)
        return np.where(ok,m[0,np.clip(yy,0,H-1),np.clip(xx,0,W-1)],f32(0)).astype(f32)
    for name,m,col in (("fy",fy,0),("fx",fx,1)):
        a,b,c,d=tap(m,yi,xi),tap(m,yi,xi+1),tap(m,yi+1,xi),tap(m,yi+1,xi+1)
        plain=((((a*nw).astype(f32)+(b*ne).astype(f32)).astype(f32)+(c*sw).astype(f32)).astype(f32)+(d*se).astype(f32)).astype(f32)
        f1=fma(d,se,fma(c,sw,fma(b,ne,(a*nw).astype(f32))))
        f2=fma(d,se,fma(c,sw,fma(a,nw,(b*ne).astype(f32))))
        print(un,name,"plain",(plain!=r[:,col]).mean(),"fma-chain",(f1!=r[:,col]).mean(),"fma-chain2",(f2!=r[:,col]).mean())
