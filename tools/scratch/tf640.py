import sys, torch
sys.path.insert(0, '.')
from cruse_amd import ops
B, T, H = 64, 401, 640
torch.manual_seed(0)
gi = (0.5 * torch.randn(B, T, 3 * H)).cuda()
w = [(torch.randn(3 * H, H) / H ** 0.5).cuda()]; b = [torch.zeros(3 * H).cuda()]
def timeit(fn, n=8):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ref = ops.gru_seq_fwd(gi, w, b, B, T, 1, H, "f32")
for tf in (1, 2):
    for rd in (0, 1):
        row = []
        for d in (0, 2, 4, 8):
            with ops.options(gru_tf=tf, gru_fwd_rd=rd, gru_poll_fwd=d):
                o = ops.gru_seq_fwd(gi, w, b, B, T, 1, H, "bf16")
                t = timeit(lambda: ops.gru_seq_fwd(gi, w, b, B, T, 1, H, "bf16"))
            row.append(f"delay {d}: {t * 1e3 / T:.3f}")
        err = float((o[0].double() - ref[0].double()).norm() / ref[0].double().norm())
        print(f"gru_tf={tf} rd={rd}: " + " | ".join(row) + f" us/step   h vs f32 {err:.2e} status {ops.gru_status()}")
