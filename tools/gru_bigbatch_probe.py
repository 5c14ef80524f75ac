"""Batches beyond the CU count (B > 96 at Hg = 640): one launch of the wide-chain kernels (16 clips per chain) against several
launches of the lean / reduce-scatter kernels on chains of 8 (library option gru_w16 = 0)."""
import sys, torch
sys.path.insert(0, '.')
from cruse_amd import ops
T, H, G = 401, 640, 1
Hg = H // G
torch.manual_seed(0)
ws = [(torch.randn(3 * Hg, Hg) / 25).cuda()]; bs = [(0.1 * torch.randn(3 * Hg)).cuda()]


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B in (104, 128, 136, 192, 256):
    gi = (0.5 * torch.randn(B, T, 3 * H)).cuda()
    dout = (0.1 * torch.randn(B, T, H)).cuda()
    for w16 in (1, 0):
        ops.set_option("gru_w16", w16)
        f = ops.gru_seq_fwd(gi, ws, bs, B, T, G, Hg, "bf16")
        dh, dgi = ops.gru_seq_bwd(dout, ws, f[1], f[3], B, T, G, Hg, "bf16", an=f[2], want_dgi=True)
        tf = timeit(lambda: ops.gru_seq_fwd(gi, ws, bs, B, T, G, Hg, "bf16", out=f))
        tb = timeit(lambda: ops.gru_seq_bwd(dout, ws, f[1], f[3], B, T, G, Hg, "bf16", an=f[2], want_dgi=True, out=(dh, dgi)))
        print(f"B={B} {'wide chains' if w16 else 'chains of 8 '}: fwd {tf:.0f} us, bwd {tb:.0f} us; status {ops.gru_status()}", flush=True)
