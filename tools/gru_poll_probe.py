"""Sweep of the first-poll delay of the tag-free recurrence kernels (library options gru_poll_fwd / gru_poll_bwd, in s_sleep(1)
periods): time per dependent step at the bench shape."""
import sys, torch
sys.path.insert(0, '.')
from cruse_amd import ops
B, T, H = 64, 401, 640
G = int(sys.argv[1]) if len(sys.argv) > 1 else 1
Hg = H // G
torch.manual_seed(0)
gi = (0.5 * torch.randn(B, T, 3 * H)).cuda()
ws = [(torch.randn(3 * Hg, Hg) / Hg ** 0.5).cuda() for _ in range(G)]; bs = [torch.zeros(3 * Hg).cuda() for _ in range(G)]
dout = (0.1 * torch.randn(B, T, H)).cuda()
def timeit(fn, n=8):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
h, coef, an, z = ops.gru_seq_fwd(gi, ws, bs, B, T, G, Hg, "bf16")
with ops.options(gru_tf=0):
    print(f"G={G} tagged kernels: fwd {timeit(lambda: ops.gru_seq_fwd(gi, ws, bs, B, T, G, Hg, 'bf16')) * 1e3 / T:.3f}  bwd {timeit(lambda: ops.gru_seq_bwd(dout, ws, coef, z, B, T, G, Hg, 'bf16')) * 1e3 / T:.3f} us/step")
for tf in (0, 1, 2, 0, 1, 2):
    with ops.options(gru_tf=tf, gru_poll_bwd=10, gru_poll_fwd=8 if tf == 2 else 0):
        a = ops.gru_seq_fwd(gi, ws, bs, B, T, G, Hg, "bf16")
        tfw = timeit(lambda: ops.gru_seq_fwd(gi, ws, bs, B, T, G, Hg, "bf16"))
        tbw = timeit(lambda: ops.gru_seq_bwd(dout, ws, coef, z, B, T, G, Hg, "bf16"))
    err = float((a[0] - h).norm() / h.norm())
    print(f"  gru_tf={tf}: fwd {tfw * 1e3 / T:.3f}  bwd {tbw * 1e3 / T:.3f} us/step   (h vs the first run {err:.2e})")
for rnd in range(1):
    for d in (0, 2, 4, 6, 8, 10):
        with ops.options(gru_poll_fwd=d, gru_poll_bwd=d):
            tf = timeit(lambda: ops.gru_seq_fwd(gi, ws, bs, B, T, G, Hg, "bf16"))
            tb = timeit(lambda: ops.gru_seq_bwd(dout, ws, coef, z, B, T, G, Hg, "bf16"))
        print(f"  delay {d}: fwd {tf * 1e3 / T:.3f}  bwd {tb * 1e3 / T:.3f} us/step")
print("status", ops.gru_status())
