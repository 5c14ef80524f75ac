import sys, torch
sys.path.insert(0, '.')
from cruse_amd import ops
B, T, H = 64, 401, 640
torch.manual_seed(0)
gi = (0.5 * torch.randn(B, T, 3 * H)).cuda()
w = [(torch.randn(3 * H, H) / H ** 0.5).cuda()]; b = [torch.zeros(3 * H).cuda()]
dout = (0.1 * torch.randn(B, T, H)).cuda()
f = ops.gru_seq_fwd(gi, w, b, B, T, 1, H, "bf16")
def timeit(fn, n=8):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for ag in (1, 0):
    for dbg in (0, 7, 1):
        with ops.options(gru_bwd_ag=ag, gru_dbg=dbg, gru_poll_bwd=3 if ag else 10):
            t = timeit(lambda: ops.gru_seq_bwd(dout, w, f[1], f[3], B, T, 1, H, "bf16"))
        print(f"ag={ag} dbg={dbg} ({'normal' if dbg == 0 else 'no operand streams' if dbg == 7 else 'no tag wait'}): {t * 1e3 / T:.3f} us/step")
print(ops.gru_status())
