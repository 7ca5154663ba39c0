import torch, time
dev=torch.device('cuda:0')
g=torch.Generator(device=dev); g.manual_seed(1)
def timed(fn,n=100):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return 1e6*(time.perf_counter()-t0)/n
shape=(8,1,4096,15,1)
print('randn f64 us', timed(lambda: torch.randn(shape,dtype=torch.float64,device=dev,generator=g)))
print('randn f32 us', timed(lambda: torch.randn(shape,dtype=torch.float32,device=dev,generator=g)))
print('randn f32->f64 us', timed(lambda: torch.randn(shape,dtype=torch.float32,device=dev,generator=g).double()))
buf=torch.empty(shape,dtype=torch.float64,device=dev)
print('normal_ f64 inplace us', timed(lambda: buf.normal_(generator=g)))
print('empty kernel-ish (zeros 8 doubles) us', timed(lambda: torch.zeros(8,dtype=torch.float64,device=dev)))
