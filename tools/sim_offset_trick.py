"""CPU simulation behind gemv_core.h: f32 rounding noise of the packed-dot int4 GEMV when the codes enter as (offset + q)\nfor offset 0 (plain q*x chain), 16 and 128 -- 16 keeps the noise at the plain chain's level, 128 is ~7x worse."""
import numpy as np, sys
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
from helpers import bf16, f32, quant_matrix, dequantize, ulp_diff_bf16
rng=np.random.default_rng(1)
n,k,g=2048,1024,128
q=quant_matrix(rng,n,k,4,g,0)
x=f32(bf16(rng.normal(0,1,size=k))).astype(np.float32)
W=dequantize(q); ref=(W@x.astype(np.float64))
w=q['weights']; codes=np.empty((n,k),np.float32); codes[:,0::2]=w&15; codes[:,1::2]=w>>4
sc=f32(q['scales']).astype(np.float32); bi=f32(q['biases']).astype(np.float32)
def run(off):
    # per 32-step: 4 chains, each sequential f32 adds of 2-products (dot2: assume (a*b+c*d) exact then +acc rounded)
    acc=np.zeros(n,np.float32)
    # lane mapping: lpr=32 lanes each one step; lane partials then butterfly; emulate per step partial then sum in f64->f32 (approx)
    parts=[]
    for c in range(k//32):
        xs=x[c*32:(c+1)*32]; cs=codes[:,c*32:(c+1)*32]+np.float32(off)
        d=[np.zeros(n,np.float32) for _ in range(4)]
        for i in range(4):
            for s in range(4):
                p=(cs[:,8*i+s].astype(np.float64)*xs[8*i+s]+cs[:,8*i+4+s].astype(np.float64)*xs[8*i+4+s])
                d[s]=(d[s].astype(np.float64)+p).astype(np.float32)
        D=((d[0]+d[1])+(d[2]+d[3])).astype(np.float32)
        S=np.float32(xs.astype(np.float32).sum(dtype=np.float32))
        grp=c*32//g
        ofp=(bi[:,grp]-np.float32(off)*sc[:,grp]).astype(np.float32)
        t=(ofp.astype(np.float64)*S).astype(np.float32)  # fma(of,S,0)
        parts.append(((sc[:,grp].astype(np.float64)*D)+t).astype(np.float32))
    tot=np.zeros(n,np.float32)
    for p in parts: tot=(tot+p).astype(np.float32)
    return tot
for off in (0,16,128):
    r=run(off)
    ub=ulp_diff_bf16(bf16(ref.astype(np.float32)),bf16(r))
    print(off,'rel f32 err rms',np.sqrt(np.mean((r-ref)**2))/np.sqrt(np.mean(ref**2)),'bf16 identical',(ub==0).mean(),'max ulp',ub.max())
