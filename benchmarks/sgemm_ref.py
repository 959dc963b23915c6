import torch,time
for n in (2048,4096,8192):
    a=torch.randn(n,n,device='cuda');b=torch.randn(n,n,device='cuda')
    for _ in range(3): a@b
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): a@b
    e1.record();torch.cuda.synchronize()
    ms=e0.elapsed_time(e1)/10
    print("rocBLAS sgemm %d: %.3f ms %.1f TF"%(n,ms,2*n**3/ms/1e9))
