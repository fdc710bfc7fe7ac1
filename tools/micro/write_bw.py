import torch, numpy as np
dev="cuda:0"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); ts=[]
    for _ in range(n):
        s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        s.record(); fn(); e.record(); torch.cuda.synchronize(); ts.append(s.elapsed_time(e))
    return float(np.median(ts))
for mb in (256, 512, 1024, 2048):
    n = mb*1024*1024//4
    a=torch.empty(n,device=dev); b=torch.randn(n,device=dev)
    tf=t(lambda: a.fill_(1.5)); tc=t(lambda: a.copy_(b)); tr=t(lambda: b.sum())
    tm=t(lambda: torch.mul(b, 2.0, out=a))
    print(f"{mb} MB: fill {tf:.3f} ms = {mb/1024/tf*1e3/1024:.2f} TB/s write | copy {tc:.3f} ms = {2*mb/1024/tc*1e3/1024:.2f} TB/s (r+w) | mul {tm:.3f} ms | sum(read) {tr:.3f} ms = {mb/1024/tr*1e3/1024:.2f} TB/s")
