import sys; sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/petit-kernel_amd'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
import petit_kernel as pk
from oracle import oracle as O
import test_gpu_parity as T
n,k=256,256
t=np.load('/root/repo/tests/golden/dequant_tables.npz')['mx']
code=(np.arange(n)[:,None]+np.arange(k)[None,:])%16
q=(code[:,0::2]|(code[:,1::2]<<4)).astype(np.uint8)
sidx=(np.arange(k//32)[None,:]+29*np.arange(n)[:,None])%237
s=(1+sidx).astype(np.uint8)
want=t[code,np.repeat(sidx,32,axis=1)]
eye=np.eye(k,dtype=np.float32)
a_bits=eye.astype(np.float16).view(np.uint16)
c=T.run_case(pk,"mx",a_bits,False,q,s,1.0,k,n,k)
with np.errstate(over='ignore'): want16=want.astype(np.float16)
got=c.view(np.float16).T
bad=np.argwhere(got.view(np.uint16)!=want16.view(np.uint16))
print(len(bad))
for (i,j) in bad[:12]: print(i,j,'code',code[i,j],'e',s[i,j//32],'got',got[i,j], hex(got.view(np.uint16)[i,j]),'want',want16[i,j], hex(want16.view(np.uint16)[i,j]), want[i,j])
g16=got.view(np.uint16); w16=want16.view(np.uint16)
bad=np.argwhere((g16!=w16) & ~(((g16&0x7fff)==0)&((w16&0x7fff)==0)))
print('non-signzero mismatches', len(bad))
for (i,j) in bad[:16]: print(i,j,'code',code[i,j],'e',s[i,j//32],'got',got[i,j], hex(g16[i,j]),'want',want16[i,j], hex(w16[i,j]), want[i,j])
