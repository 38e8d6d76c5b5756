import sys, numpy as np, torch
sys.path.insert(0,'tools')
import quantize_weights as Q
n,k=2048,4096
w=Q.synthetic_weights(n,k,seed=1)
q,s,ws2=Q.quantize_nvfp4(w)
codes=np.empty((n,k),np.uint8); codes[:,0::2]=q&15; codes[:,1::2]=q>>4
mag=Q.FP4_VALUES[codes&7]*np.where(codes&8,-1.0,1.0)
sf=Q._e4m3_to_f32(s).astype(np.float64)
wd=mag*np.repeat(sf,16,axis=1)          # in units of ws2
def e4m3(x): return torch.from_numpy(x.astype(np.float32)).to(torch.float8_e4m3fn).float().numpy().astype(np.float64)
def e2m3(x):
    ax=np.abs(x); out=np.zeros_like(ax)
    # subnormal step .125 below 1; normals e=0..2 (1..7.5)
    e=np.floor(np.log2(np.maximum(ax,1e-30))); e=np.clip(e,0,2)
    step=np.where(ax<1,0.125,2.0**e/8)
    r=np.round(ax/step)  # numpy round = half to even
    out=np.minimum(r*step,7.5)
    return np.sign(x)*out
blk=wd.reshape(n,k//32,32)
amax=np.abs(blk).max(-1,keepdims=True)
# FP8: E so that amax/2^E <= 448 : choose exponent of amax - 8 (amax in [256,512) -> too big) -> use floor(log2(amax)) - 8 => amax/2^E in [256,512) >448 possible; use -7: [128,256)
def blockenc(fn, emax):
    E=np.floor(np.log2(np.maximum(amax,2.0**-126)))-emax
    return fn(blk/2.0**E)*2.0**E
for name,fn,emax in (("fp8 e4m3 (E=floor(log2 amax)-7)",e4m3,7),("fp8 e4m3 (-8, sat 448)",lambda x:e4m3(np.clip(x,-448,448)),8),("fp6 e2m3 (-2)",e2m3,2)):
    r=blockenc(fn,emax).reshape(n,k)
    err=r-wd
    print(name,"rel rms err vs nvfp4 value: %.4f"%(np.sqrt((err**2).mean())/np.sqrt((wd**2).mean())), " changed frac %.3f"%((err!=0).mean()), " max rel %.3f"%(np.abs(err)/np.maximum(np.abs(wd),1e-30)).max())
    e0=wd*ws2-w; e1=r*ws2-w
    print("   weight error rms rel to w rms: nvfp4 %.4f  re-encoded %.4f"%(np.sqrt((e0**2).mean())/np.sqrt((w.astype(np.float64)**2).mean()), np.sqrt((e1**2).mean())/np.sqrt((w.astype(np.float64)**2).mean())))
# exponent differences between the two groups of a block
ex=np.floor(np.log2(np.maximum(sf,2.0**-10))).reshape(n,k//32,2)
d=np.abs(ex[:,:,0]-ex[:,:,1]); print("group exponent diff hist", np.bincount(d.astype(int).ravel())[:6]/d.size)
