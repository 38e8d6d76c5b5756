import torch, time, inspect
print(torch.__version__)
print('external' in inspect.signature(torch.cuda.Event.__new__).parameters or 'external' in (torch.cuda.Event.__doc__ or ''))
dev=torch.device('cuda',0)
s=torch.cuda.Stream(dev)
x=torch.randn(4096,4096,device=dev)
try:
    e0=torch.cuda.Event(enable_timing=True, external=True); e1=torch.cuda.Event(enable_timing=True, external=True)
except TypeError as ex:
    print('no external kw', ex); raise SystemExit
with torch.cuda.stream(s):
    y=x@x; s.synchronize()
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        e0.record(s)
        for i in range(20): y=x@x
        e1.record(s)
    for _ in range(3):
        g.replay(); s.synchronize()
        print('in-graph events ms', e0.elapsed_time(e1))
    a=torch.cuda.Event(enable_timing=True); b=torch.cuda.Event(enable_timing=True)
    a.record(s); g.replay(); b.record(s); s.synchronize(); print('outer ms', a.elapsed_time(b), 'inner', e0.elapsed_time(e1))
