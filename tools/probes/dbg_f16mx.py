import sys, numpy as np, torch
sys.path.insert(0,'petit-kernel_amd'); sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import petit_kernel as pk
from oracle import oracle as O
DEV='cuda'
np.set_printoptions(linewidth=250)
def run(m,n,k,sid,const_scale=None,a_one=False,seed=1):
    rng=np.random.default_rng(seed)
    a=rng.standard_normal((m,k),dtype=np.float32).astype(np.float16)
    if a_one: a[:]=1
    q=rng.integers(0,256,(n,k//2),dtype=np.uint8)
    if a_one: q[:]=0x22  # code 2 = 1.0
    s=rng.integers(119,136,(n,k//32),dtype=np.uint8)
    if const_scale: s[:]=const_scale
    dq=O.dequant_mxfp4(q,s)
    _,ref=O.gemm_ref(a.view(np.uint16),False,dq,1.0)
    b=pk.repack_mxfp4(torch.from_numpy(q).to(DEV).view(torch.int32),n,k)
    sp=pk.process_mxfp4_scales(torch.from_numpy(s).to(DEV),n,k)
    ad=torch.from_numpy(a).to(DEV); gsd=torch.tensor([1.0],dtype=torch.float32,device=DEV)
    c=pk.mul_mxfp4_a16(ad,b,sp,gsd,m,n,k,sid).float().cpu().numpy()
    err=np.abs(c-ref); bound=np.maximum(1e-2,1e-2*np.abs(ref))
    badm=~(err<=bound)
    print(f"sid {sid:#x} m={m} n={n} k={k} const={const_scale} a_one={a_one}: bad {badm.sum()}")
    print(" bad per column n:", badm.sum(0))
    print(" bad per row m:", badm.sum(1))
    if a_one:
        print(" c row0:", c[0,:32]); print(" ref row0:", ref[0,:32])
sid=0x181b811023100101
run(16,128,2048,sid)
run(16,128,2048,sid,const_scale=127,a_one=True)
run(16,128,2048,sid,const_scale=127)
run(16,128,1024,sid)
run(16,128,8192,sid)
run(16,128,16384,sid)
