import sys, time, torch
sys.path.insert(0, '.')
from torch.utils.data import DataLoader
from cruse_amd.data import HostPoolPairs, DevicePairs
import os
print("cpus", os.cpu_count(), "torch threads", torch.get_num_threads())
B, L = 64, 64000
for nw in (0, 4, 8, 16):
    ds = HostPoolPairs(num=40 * B, length=L, seed=1, pool=64)
    if nw == 0: ds._ensure()
    ld = DataLoader(ds, batch_size=B, shuffle=False, drop_last=True, num_workers=nw, persistent_workers=nw > 0)
    for ep in range(2):
        t0 = time.perf_counter(); n = 0
        for a, b in ld:
            n += 1
        dt = time.perf_counter() - t0
        print(f"workers {nw} epoch {ep}: {dt / n * 1e3:.2f} ms per batch", flush=True)
    del ld
x = torch.randn(B, L); p = torch.empty(B, L).pin_memory()
t0 = time.perf_counter()
for _ in range(10): p.copy_(x)
print(f"pinned staging copy of one [64,64000] f32: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms")
dev = torch.device("cuda")
d = x.cuda(); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): d.copy_(p, non_blocking=True)
torch.cuda.synchronize()
print(f"H2D of one pinned [64,64000]: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms")
dp = DevicePairs(num=40 * B, length=L, seed=1, pool=64)
idx = torch.arange(B)
dp.device_batch(idx, dev); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20): dp.device_batch(idx, dev)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"device_batch: host {(t1 - t0) / 20 * 1e3:.2f} ms per call, total {(t2 - t0) / 20 * 1e3:.2f} ms")
import gc
t0 = time.perf_counter(); gc.collect(); print(f"gc.collect {(time.perf_counter() - t0) * 1e3:.1f} ms")
