"""Register-direct sweep in the forward lean kernel (library option gru_fwd_rd) against the LDS-image form: bit-equality, time per step
over first-poll delays, phase stamps.  usage: python tools/gru_rd_probe.py"""
import sys, torch
sys.path.insert(0, '.')
from cruse_amd import ops


def timeit(fn, n=8):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


H = 640
for (B, T) in ((3, 7), (20, 50), (64, 401)):
    torch.manual_seed(B + T)
    gi = (0.5 * torch.randn(B, T, 3 * H)).cuda()
    w = [(torch.randn(3 * H, H) / H ** 0.5).cuda()]; b = [(0.1 * torch.randn(3 * H)).cuda()]
    out = {}
    for rd in (0, 1):
        with ops.options(gru_fwd_rd=rd):
            out[rd] = ops.gru_seq_fwd(gi, w, b, B, T, 1, H, "bf16"); torch.cuda.synchronize()
    print(f"B={B} T={T}: register-direct == image form: " + " ".join(str(bool((x.float() == y.float()).all())) for x, y in zip(out[0], out[1])) + f" status {ops.gru_status()}")
B, T = 64, 401
for rnd in range(2):
    for rd in (0, 1):
        row = []
        for d in (0, 2, 4, 6, 8):
            with ops.options(gru_fwd_rd=rd, gru_poll_fwd=d):
                t = timeit(lambda: ops.gru_seq_fwd(gi, w, b, B, T, 1, H, "bf16"))
            row.append(f"delay {d}: {t * 1e3 / T:.3f}")
        print(f"  rd={rd}: " + " | ".join(row) + " us/step")
for rd in (0, 1):
    for d in (0, 4):
        with ops.options(gru_fwd_rd=rd, gru_dbg=32, gru_poll_fwd=d):
            tw = timeit(lambda: ops.gru_seq_fwd(gi, w, b, B, T, 1, H, "bf16")); torch.cuda.synchronize()
            for buf in ops._gru_hdr.values():
                st = buf[64:112].view(torch.int64).tolist(); n = max(st[5], 1)
                print(f"  rd={rd} delay {d} stamped {tw * 1e3 / T:.3f} us/step: sweep {st[0] / n:.0f} | image / barrier {st[1] / n:.0f} | MFMA phase {st[2] / n:.0f} | gates + publish {st[3] / n:.0f} | re-polls {st[4] / n:.2f}")
print("status", ops.gru_status())
