import sys, numpy as np, torch
sys.path.insert(0,'petit-kernel_amd'); sys.path.insert(0,'.')
import petit_kernel as pk
from oracle import oracle as O
n,k=32,256
rng=np.random.default_rng(5)
q=rng.integers(0,256,(n,k//2),dtype=np.uint8)
s=rng.integers(0x30,0x50,(n,k//16),dtype=np.uint8)
dev=torch.device('cuda')
b=pk.repack_nvfp4(torch.from_numpy(q).to(dev).view(torch.int32),n,k)
sp=pk.process_nvfp4_scales(torch.from_numpy(s).to(dev).view(torch.float8_e4m3fn),n,k)
img=pk.nvfp4_native_image(b,sp,n,k).cpu()
host=pk.offline.nvfp4_native_image_cpu(b.cpu(),sp.cpu(),n,k)
d=pk.offline.nvfp4_native_image_dequant_cpu(img,n,k).numpy(); h=pk.offline.nvfp4_native_image_dequant_cpu(host,n,k).numpy()
bad=np.argwhere(d!=h)
print("mismatch count",len(bad),"of",n*k)
import collections
print("by element index in block:",sorted(collections.Counter((bad[:,1]%32).tolist()).items()))
print("by row:",sorted(collections.Counter((bad[:,0]).tolist()).items())[:10])
print("scale bytes equal:", bool((img[-n*k//32:]==host[-n*k//32:]).all()))
for r,c in bad[:10]:
    print(r,c,d[r,c],h[r,c])
