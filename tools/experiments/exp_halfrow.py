import sys, torch
sys.path.insert(0, '/root/repo')
from cartnet_amd import ops
dev='cuda:0'
R,S,C=4*177140,4*12416,256
g=torch.Generator().manual_seed(0)
big=torch.randn(R,2*C,generator=g).to(dev)
cont=big[:,:C].contiguous()
q=torch.randn(S,3*C,generator=g).to(dev)[:,:C]
idx=torch.sort(torch.randint(0,S,(R,),generator=g)).values.int().to(dev)
s1=torch.empty(C,device=dev); s2=torch.empty(C,device=dev)
def t(fn,n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n*1e3
print('half rows (ld 2C):', t(lambda: ops.rowmul_stats(big[:,:C],q,idx,0.0625,s1,s2)),'us (includes the finaliser)')
print('contiguous (ld C):', t(lambda: ops.rowmul_stats(cont,q,idx,0.0625,s1,s2)),'us')
