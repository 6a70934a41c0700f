import torch, time
torch.cuda.init()
N=8192
a=torch.randn(N,N,dtype=torch.float64,device='cuda'); A=a@a.T/N+torch.eye(N,dtype=torch.float64,device='cuda')*2
for name,fn in (("torch.linalg.cholesky (vendor potrf) N=8192", lambda: torch.linalg.cholesky(A)),):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(5): fn()
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/5
    print(name,"%.2f ms  %.1f TF/s"%(dt*1e3, N**3/3/dt/1e12),flush=True)
B=4
Ab=A.unsqueeze(0).repeat(B,1,1).contiguous()
for _ in range(1): torch.linalg.cholesky(Ab)
torch.cuda.synchronize(); t=time.perf_counter()
for _ in range(2): torch.linalg.cholesky(Ab)
torch.cuda.synchronize(); dt=(time.perf_counter()-t)/2
print("batched x%d: %.2f ms per matrix"%(B,dt/B*1e3))
