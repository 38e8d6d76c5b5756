import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/petit-kernel_amd'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
import petit_kernel as pk
from oracle import oracle as O
import test_gpu_parity as T
n,k=128,256
t=np.load('/root/repo/tests/golden/dequant_tables.npz')['nv']
code=np.arange(n)[:,None]%16*np.ones((1,k),dtype=np.int64); code=(code+np.arange(k)[None,:])%16
q=(code[:,0::2]|(code[:,1::2]<<4)).astype(np.uint8)
sidx=(np.arange(n)[:,None]*16+np.arange(k//16)[None,:])%126
s=(1+sidx).astype(np.uint8)
want=t[code,np.repeat(sidx,16,axis=1)]
a_bits=O.f32_to_bf16_bits(np.eye(k,dtype=np.float32))
c=T.run_case(pk,"nv",a_bits,True,q,s,1.0,k,n,k)
got=T.to_f32(c,True).T
bad=np.argwhere(got!=want)
print(len(bad), bad[:10])
for (i,j) in bad[:10]: print(i,j,'code',code[i,j],'scale byte',s[i,j//16], 'got',got[i,j],'want',want[i,j])
