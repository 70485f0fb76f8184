import time, torch
dev="cuda"
a=torch.randn(8192,8192,device=dev,dtype=torch.bfloat16); b=torch.randn(8192,8192,device=dev,dtype=torch.bfloat16)
st=[torch.zeros(1024,dtype=torch.int32).pin_memory() for _ in range(5)]
ds=[torch.zeros(1024,dtype=torch.int32,device=dev) for _ in range(5)]
def run(busy, label, fn):
    torch.cuda.synchronize()
    if busy:
        for _ in range(20): a@b
    t0=time.perf_counter()
    fn()
    t1=time.perf_counter()
    torch.cuda.synchronize()
    print(f"{label:40s} busy={busy}: cpu issue {1e3*(t1-t0):7.3f} ms")
def copies():
    for s,d in zip(st,ds): d.copy_(s,non_blocking=True)
def copies_np():
    import numpy as np
    for s,d in zip(st,ds):
        s[:512]=torch.from_numpy(np.arange(512,dtype=np.int32)); d[:512].copy_(s[:512],non_blocking=True)
for busy in (False,True):
    run(busy,"5 pinned->dev copy_ non_blocking",copies)
    run(busy,"5 numpy->pinned + sliced copies",copies_np)
    run(busy,"5 pageable .to(dev)", lambda: [torch.zeros(1024,dtype=torch.int32).to(dev,non_blocking=True) for _ in range(5)])
g=torch.cuda.CUDAGraph()
x=torch.zeros(1024,device=dev)
with torch.cuda.graph(g):
    for _ in range(20): c=a@b
def copies_then_graph():
    copies(); g.replay()
for i in range(3):
    run(False,"copies + graph replay (prev graph running)",copies_then_graph)
