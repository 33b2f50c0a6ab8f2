import torch, time
from concurrent.futures import ThreadPoolExecutor
torch.cuda.init()
def t(f, rep=1):
    torch.cuda.synchronize(); t0=time.perf_counter(); 
    for _ in range(rep): f()
    torch.cuda.synchronize(); return (time.perf_counter()-t0)/rep
N = 2<<30  # bytes
a = torch.empty(N//4, dtype=torch.float32); a.fill_(1.0)
b = torch.empty(N//4, dtype=torch.float32); b.fill_(2.0)
print("cpu copy_ (1 call) GB/s:", N/1e9/t(lambda: b.copy_(a), 3), "threads:", torch.get_num_threads())
pool = ThreadPoolExecutor(16)
def par(dst, src, k=16):
    n = dst.numel(); step=(n+k-1)//k
    list(pool.map(lambda i: dst[i*step:(i+1)*step].copy_(src[i*step:(i+1)*step]), range(k)))
print("cpu copy 16 threads GB/s:", N/1e9/t(lambda: par(b,a), 3))
t0=time.perf_counter(); p = torch.empty(N//4, dtype=torch.float32, pin_memory=True); dt=time.perf_counter()-t0
print("pinned alloc 2 GiB: %.3f s (%.1f GB/s)"%(dt, N/1e9/dt))
del p; t0=time.perf_counter(); p = torch.empty(N//4, dtype=torch.float32, pin_memory=True); dt=time.perf_counter()-t0
print("pinned alloc again (cached): %.4f s"%dt)
print("pageable->pinned copy_ GB/s:", N/1e9/t(lambda: p.copy_(a), 3))
print("pageable->pinned 16 thr GB/s:", N/1e9/t(lambda: par(p,a), 3))
d = torch.empty(N//4, dtype=torch.float32, device="cuda")
print("H2D pinned GB/s:", N/1e9/t(lambda: d.copy_(p, non_blocking=True), 3))
print("D2H pinned GB/s:", N/1e9/t(lambda: p.copy_(d, non_blocking=True), 3))
print("H2D pageable GB/s:", N/1e9/t(lambda: d.copy_(a), 2))
print("D2H pageable GB/s:", N/1e9/t(lambda: a.copy_(d), 2))
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
p2 = torch.empty(N//4, dtype=torch.float32, pin_memory=True); d2 = torch.empty_like(d)
def both():
    with torch.cuda.stream(s1): d.copy_(p, non_blocking=True)
    with torch.cuda.stream(s2): p2.copy_(d2, non_blocking=True)
print("H2D+D2H concurrently, GB/s total:", 2*N/1e9/t(both, 3))
t0=time.perf_counter(); r = torch.cuda.cudart().cudaHostRegister(a.data_ptr(), N, 0); dt=time.perf_counter()-t0
print("hostRegister 2 GiB: rc", r, "%.3f s"%dt)
print("H2D registered GB/s:", N/1e9/t(lambda: d.copy_(a, non_blocking=True), 3))
