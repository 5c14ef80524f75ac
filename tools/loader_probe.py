import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from torch.utils.data import DataLoader
from cruse_amd.data import HostPoolPairs
B, L = 64, 64000
ds = HostPoolPairs(num=30 * B, length=L, seed=1, pool=64)
pn = torch.empty(B, L).pin_memory(); pc = torch.empty(B, L).pin_memory()
dev = torch.device("cuda"); dn = torch.empty(B, L, device=dev)
for mode in ("iterate_only", "torch_copy", "numpy_copy", "direct_h2d", "threads1_copy"):
    ld = DataLoader(ds, batch_size=B, shuffle=True, drop_last=True, num_workers=4, persistent_workers=True)
    if mode == "threads1_copy": torch.set_num_threads(1)
    for ep in range(2):
        t0 = time.perf_counter(); n = 0; tc = 0.0
        for a, b in ld:
            t1 = time.perf_counter()
            if mode in ("torch_copy", "threads1_copy"):
                pn.copy_(a); pc.copy_(b)
            elif mode == "numpy_copy":
                np.copyto(pn.numpy(), a.numpy()); np.copyto(pc.numpy(), b.numpy())
            elif mode == "direct_h2d":
                dn.copy_(a); dn.copy_(b)
            tc += time.perf_counter() - t1
            n += 1
        dt = time.perf_counter() - t0
        print(f"{mode} epoch {ep}: {dt / n * 1e3:.2f} ms per batch (copy part {tc / n * 1e3:.2f} ms)", flush=True)
    del ld
