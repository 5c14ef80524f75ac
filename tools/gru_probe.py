import os, sys, time, torch
sys.path.insert(0, '.')
from cruse_amd import ops
B, T, H = 64, 401, 640
torch.manual_seed(0)
gi = (0.5 * torch.randn(B, T, 3 * H)).cuda()
w = (torch.randn(3 * H, H) / 25).cuda(); b = torch.zeros(3 * H).cuda()
dout = (0.1 * torch.randn(B, T, H)).cuda()
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
h, coef, an, z = ops.gru_seq_fwd(gi, [w], [b], B, T, 1, H, "bf16")
tf = timeit(lambda: ops.gru_seq_fwd(gi, [w], [b], B, T, 1, H, "bf16"))
tb = timeit(lambda: ops.gru_seq_bwd(dout, [w], coef, z, B, T, 1, H, "bf16"))
print(f"dbg={os.environ.get('CRUSE_GRU_DBG','0')} B={B}: fwd {tf*1e3/T:.2f} us/step  bwd {tb*1e3/T:.2f} us/step  status {ops.gru_status()}")
for Bx in (8, 16, 32):
    g2 = gi[:Bx].contiguous()
    tf = timeit(lambda: ops.gru_seq_fwd(g2, [w], [b], Bx, T, 1, H, "bf16"))
    print(f"   B={Bx}: fwd {tf*1e3/T:.2f} us/step")
