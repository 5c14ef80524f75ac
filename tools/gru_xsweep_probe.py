import sys, torch
sys.path.insert(0, '.')
from cruse_amd import ops
B, T, H = 64, 401, 640
torch.manual_seed(0)
gi = (0.5 * torch.randn(B, T, 3 * H)).cuda()
ws = [(torch.randn(3 * H, H) / H ** 0.5).cuda()]; bs = [torch.zeros(3 * H).cuda()]
def timeit(fn, n=8):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for rnd in range(2):
    for xs in (0, 1, 2, 3):
        with ops.options(gru_xsweep=xs):
            t = timeit(lambda: ops.gru_seq_fwd(gi, ws, bs, B, T, 1, H, "bf16"))
        with ops.options(gru_xsweep=xs, gru_dbg=32):
            tw = timeit(lambda: ops.gru_seq_fwd(gi, ws, bs, B, T, 1, H, "bf16")); torch.cuda.synchronize()
            for buf in ops._gru_hdr.values():
                st = buf[64:112].view(torch.int64).tolist(); n = max(st[5], 1)
                ph = f"sweep {st[0] / n:.0f} | LDS image + barrier {st[1] / n:.0f} | MFMA phase {st[2] / n:.0f} | gates + publish {st[3] / n:.0f} | re-polls {st[4] / n:.2f}"
        print(f"xsweep {xs} ({10 * (1 + xs)} KB swept, {1 + xs} publishes): {t * 1e3 / T:.3f} us/step | stamped {tw * 1e3 / T:.3f}: {ph}")
print("status", ops.gru_status())
