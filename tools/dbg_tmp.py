import sys, ctypes as C
sys.path.insert(0,'/root/repo/petit-kernel_amd'); sys.path.insert(0,'/root/repo')
import torch, petit_kernel
from petit_kernel import _lib
dev=torch.device('cuda',0)
M,N,K=1,8192,8192
def timeit(fn, launches=100, reps=5):
    stream=torch.cuda.Stream(dev)
    with torch.cuda.stream(stream):
        fn(); stream.synchronize()
        g=torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for i in range(launches): fn()
        g.replay(); stream.synchronize()
        ts=[]
        for _ in range(reps):
            e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
            e0.record(stream); g.replay(); e1.record(stream); stream.synchronize()
            ts.append(e0.elapsed_time(e1)*1e3/launches)
    return sorted(ts)[len(ts)//2]
a=torch.randn((M,K),device=dev).bfloat16(); gs=torch.ones(1,device=dev)
g=torch.Generator().manual_seed(1)
packed=[]
for i in range(10):
    q=torch.randint(0,256,(N,K//2),generator=g,dtype=torch.uint8)
    s=(torch.rand((N,K//16),generator=g)*3.5+0.25).to(torch.float8_e4m3fn)
    b=petit_kernel.repack_nvfp4(q.to(dev).view(torch.int32),N,K)
    sp=petit_kernel.process_nvfp4_scales(s.to(dev),N,K)
    packed.append((b,sp))
torch.cuda.synchronize()
# flush MALL between measurements by touching a big buffer
big=torch.empty(512<<20,dtype=torch.uint8,device=dev)
def one(b,sp):
    def f():
        big.add_(1) if False else None
        petit_kernel.mul_nvfp4_a16(a,b,sp,gs,M,N,K,-1)
    return f
for i,(b,sp) in enumerate(packed):
    print(i, 'b ptr %x (mod 2M %x) sp ptr %x'%(b.data_ptr(), b.data_ptr()%(2<<20), sp.data_ptr()))
# time each copy alone is cache-resident (MALL) -> instead time pairs rotating among 10 but report per-copy via rocprof... simpler: time all-rotating
def rot():
    rot.i=(rot.i+1)%10
    b,sp=packed[rot.i]; petit_kernel.mul_nvfp4_a16(a,b,sp,gs,M,N,K,-1)
rot.i=0
print('rotating bench-style buffers', timeit(rot))
fresh=[(b.clone(),sp.clone()) for b,sp in packed]
def rot2():
    rot2.i=(rot2.i+1)%10
    b,sp=fresh[rot2.i]; petit_kernel.mul_nvfp4_a16(a,b,sp,gs,M,N,K,-1)
rot2.i=0
print('rotating cloned buffers', timeit(rot2))
for i,(b,sp) in enumerate(fresh[:3]):
    print(i, 'b ptr %x sp ptr %x'%(b.data_ptr(), sp.data_ptr()))
# one big arena
arena_w=torch.empty((10,N//16,2*K),dtype=torch.int32,device=dev); arena_s=torch.empty((10,N,K//16),dtype=torch.float8_e4m3fn,device=dev)
for i,(b,sp) in enumerate(packed): arena_w[i].copy_(b); arena_s[i].copy_(sp)
def rot3():
    rot3.i=(rot3.i+1)%10
    petit_kernel.mul_nvfp4_a16(a,arena_w[rot3.i],arena_s[rot3.i],gs,M,N,K,-1)
rot3.i=0
print('rotating arena buffers', timeit(rot3))
