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
ref = ops.gru_seq_fwd(gi, w, b, B, T, 1, H, "bf16")
for rnd in range(2):
    for stag in (0, 1, 2, 3, 4):
        row = []
        for d in (0, 1, 2, 3):
            with ops.options(gru_stag_fwd=stag, gru_poll_fwd=d):
                t = timeit(lambda: ops.gru_seq_fwd(gi, w, b, B, T, 1, H, "bf16"))
            row.append(f"delay {d}: {t * 1e3 / T:.3f}")
        with ops.options(gru_stag_fwd=stag):
            o = ops.gru_seq_fwd(gi, w, b, B, T, 1, H, "bf16"); torch.cuda.synchronize()
        print(f"stagger {stag}: " + " | ".join(row) + f" us/step  same bits {all(bool((x.float() == y.float()).all()) for x, y in zip(o, ref))}")
for stag in (0, 2):
    with ops.options(gru_stag_fwd=stag, gru_dbg=32):
        tw = timeit(lambda: ops.gru_seq_fwd(gi, w, b, B, T, 1, H, "bf16")); torch.cuda.synchronize()
        for buf in ops._gru_hdr.values():
            st = buf[64:112].view(torch.int64).tolist(); n = max(st[5], 1)
            print(f"  stagger {stag} stamped {tw * 1e3 / T:.3f} us/step: sweep {st[0] / n:.0f} | barrier {st[1] / n:.0f} | MFMA phase {st[2] / n:.0f} | gates + publish {st[3] / n:.0f} | re-polls {st[4] / n:.2f}")
print(ops.gru_status())
