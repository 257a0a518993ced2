"""Developer probe: achievable copy / read / write bandwidth with torch elementwise kernels."""
import torch, time
def t(fn, n=20):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
for mb in (66, 200, 1000):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, device="cuda", dtype=torch.float32).normal_(); b = torch.empty_like(a)
    dt = t(lambda: b.copy_(a)); print(f"copy {mb} MB: {2*mb/1e3/dt/1e3*1e3:.0f} GB/s (r+w)")
    dt = t(lambda: a.sum()); print(f"read {mb} MB: {mb/1e3/dt:.0f} GB/s")
    dt = t(lambda: b.fill_(1.0)); print(f"write {mb} MB: {mb/1e3/dt:.0f} GB/s")
    dt = t(lambda: a.add_(1.0)); print(f"rmw {mb} MB: {2*mb/1e3/dt:.0f} GB/s (r+w)")
