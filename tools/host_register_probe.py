import torch, time, threading
from concurrent.futures import ThreadPoolExecutor
torch.cuda.init()
rt = torch.cuda.cudart()
G = 1<<30
def reg_test(nthreads, per=2*G):
    ts = [torch.empty(per//4, dtype=torch.float32) for _ in range(nthreads)]
    t0=time.perf_counter()
    with ThreadPoolExecutor(nthreads) as ex:
        rcs = list(ex.map(lambda t: rt.cudaHostRegister(t.data_ptr(), per, 0), ts))
    dt=time.perf_counter()-t0
    t1=time.perf_counter()
    for t in ts: rt.cudaHostUnregister(t.data_ptr())
    du=time.perf_counter()-t1
    print(f"{nthreads} thread(s) x {per/G:.0f} GiB: register {dt:.3f} s = {nthreads*per/1e9/dt:.1f} GB/s, unregister {du:.3f} s, rc {set(map(int,rcs))}")
for n in (1,2,4,8): reg_test(n)
# fault-in first (touch pages in parallel with torch), then register
t = torch.empty(2*G//4, dtype=torch.float32)
t0=time.perf_counter(); t.zero_(); dz=time.perf_counter()-t0
t0=time.perf_counter(); rt.cudaHostRegister(t.data_ptr(), 2*G, 0); dr=time.perf_counter()-t0
print(f"zero_ (parallel first touch) 2 GiB: {dz:.3f} s = {2*G/1e9/dz:.1f} GB/s; register after touch: {dr:.3f} s = {2*G/1e9/dr:.1f} GB/s")
d = torch.empty(2*G//4, dtype=torch.float32, device='cuda')
torch.cuda.synchronize(); t0=time.perf_counter(); t.copy_(d, non_blocking=True); torch.cuda.synchronize(); print("D2H into registered: %.1f GB/s"%(2*G/1e9/(time.perf_counter()-t0)))
rt.cudaHostUnregister(t.data_ptr())
