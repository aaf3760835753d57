import json, torch
dev="cuda"
res={}
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s,e=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e)/n*1e-3
for mb in (32, 256, 2048):
    x=torch.empty(mb<<20, dtype=torch.uint8, device=dev)
    y=torch.empty_like(x)
    res[f"fill_{mb}MB_TBps"]=round((mb<<20)/t(lambda: x.fill_(1))/1e12,2)
    res[f"copy_{mb}MB_TBps(read+write)"]=round(2*(mb<<20)/t(lambda: y.copy_(x))/1e12,2)
    xb=x.view(torch.bfloat16)
    res[f"read_sum_{mb}MB_TBps"]=round((mb<<20)/t(lambda: xb.sum())/1e12,2)
print(json.dumps(res))
