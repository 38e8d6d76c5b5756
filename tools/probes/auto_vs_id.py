import sys
sys.path.insert(0,'/root/repo/petit-kernel_amd'); sys.path.insert(0,'/root/repo/tools')
import torch, benchlib as BL
from petit_kernel import _lib
dev=torch.device('cuda',0); stream=torch.cuda.Stream(dev)
n,k=BL.LLAMA70B['down']
w=BL.Weights('nv',n,k,1280,dev)
for m in (1,8,16):
    g=BL.Gemm(w,m,torch.bfloat16,dev)
    sid=g.default_solution()
    for rep in range(2):
        ra=g.time(_lib.PETIT_SOLUTION_AUTO,stream,reps=7)
        ri=g.time(sid,stream,reps=7)
        print(m, hex(sid), 'auto %.2f'%ra['us'], 'id %.2f'%ri['us'], 'launches', ra['launches'], ri['launches'], flush=True)
