import torch
a = torch.empty(20,256,256,64, device='cuda'); b = torch.randn_like(a)
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record(); [f() for _ in range(n)]; e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
by = a.numel()*4
ms=t(lambda: a.zero_()); print('zero_ ', ms*1e3,'us', by/ms/1e6,'GB/s write')
ms=t(lambda: a.copy_(b)); print('copy_ ', ms*1e3,'us', 2*by/ms/1e6,'GB/s r+w')
ms=t(lambda: torch.add(b,1.0,out=a)); print('add  ', ms*1e3,'us', 2*by/ms/1e6,'GB/s r+w')
