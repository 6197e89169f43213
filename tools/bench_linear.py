import torch, sys
sys.path.insert(0,'.')
from kgwas_amd import ops
def bench(f, n=200):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(10): f()
    g.replay(); torch.cuda.synchronize()
    e0.record()
    for _ in range(n//10): g.replay()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/(n//10*10)*1e3
for rows,K,N in [(125000,128,128),(137000,128,128),(125000,20,128),(20032,128,128),(13700,128,128)]:
    X=torch.randn(rows,K,device='cuda'); W=torch.randn(N,K,device='cuda'); b=torch.randn(N,device='cuda'); M=torch.randn(rows,N,device='cuda')
    t1=bench(lambda: ops.linear(X,W,b,relu=True)); t2=bench(lambda: torch.relu_(torch.addmm(b,X,W.t())))
    t3=bench(lambda: ops.linear(X,W,None,mask=M,w_kn=True)) if K==N else 0
    fl=2*rows*K*N/1e12
    print(f'{rows}x{K}x{N}: kgw_linear {t1:7.1f} us ({fl/t1*1e6:5.1f} TF)  kn+mask {t3:7.1f}  torch addmm+relu {t2:7.1f} us ({fl/t2*1e6:5.1f} TF)')
for rows,K,N in [(1171,2176,128),(512,768,128)]:
    X=torch.randn(rows,K,device='cuda'); W=torch.randn(K,N,device='cuda')
    t1=bench(lambda: ops.linear(X,W,None,w_kn=True)); t2=bench(lambda: X@W)
    print(f'{rows}x{K}x{N} kn: linear {t1:7.1f}  torch {t2:7.1f}')
for rows,K,N in [(1171,128,2176),(1171,128,1536),(512,128,768)]:
    X=torch.randn(rows,K,device='cuda'); W=torch.randn(N,K,device='cuda')
    t1=bench(lambda: ops.linear(X,W,None)); t2=bench(lambda: X@W.t())
    print(f'{rows}x{K}x{N} nk (layer-transform dX): linear {t1:7.1f}  torch {t2:7.1f}')
