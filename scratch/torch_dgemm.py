import torch, time
torch.cuda.init()
def bench(m,n,k,reps=10):
    a=torch.randn(m,k,dtype=torch.float64,device='cuda'); b=torch.randn(k,n,dtype=torch.float64,device='cuda'); c=torch.randn(m,n,dtype=torch.float64,device='cuda')
    for _ in range(3): torch.addmm(c,a,b,out=c)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(reps): torch.addmm(c,a,b,out=c)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/reps
    print("rocBLAS/hipBLASLt dgemm m=%d n=%d k=%d: %.3f ms  %.1f TF/s"%(m,n,k,dt*1e3,2*m*n*k/dt/1e12),flush=True)
bench(8192,8192,8192,5)
bench(7744,7680,2048)
bench(7744,7680,512,20)
bench(7744,7680,256,20)
bench(4096,4096,512,20)
