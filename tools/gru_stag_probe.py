"""Two polls in flight in the backward tag-free recurrence (library option gru_stag_bwd = s_sleep(1) periods between the two polls,
0 = one poll at a time) against the first-poll delay (gru_poll_bwd): time per dependent step at the bench shape, results compared
with the one-poll kernel's.  usage: python tools/gru_stag_probe.py [G]"""
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
ref = ops.gru_seq_bwd(dout, ws, coef, z, B, T, G, Hg, "bf16", an=an, want_dgi=True)
torch.cuda.synchronize()
for rnd in range(2):
    for stag in (0, 2, 3, 4, 5, 6, 8):
        row = []
        for d in (10, 6, 3, 0):
            with ops.options(gru_stag_bwd=stag, gru_poll_bwd=d):
                tb = timeit(lambda: ops.gru_seq_bwd(dout, ws, coef, z, B, T, G, Hg, "bf16"))
            row.append(f"delay {d}: {tb * 1e3 / T:.3f}")
        with ops.options(gru_stag_bwd=stag):
            o = ops.gru_seq_bwd(dout, ws, coef, z, B, T, G, Hg, "bf16", an=an, want_dgi=True)
        torch.cuda.synchronize()
        same = bool((o[0] == ref[0]).all()) and bool((o[1] == ref[1]).all())
        print(f"G={G} stagger {stag}: " + " | ".join(row) + f" us/step   identical to one-poll: {same}")
if G == 1:
    for stag in (0, 4):
        with ops.options(gru_stag_bwd=stag, gru_dbg=32):
            tb = timeit(lambda: ops.gru_seq_bwd(dout, ws, coef, z, B, T, G, Hg, "bf16")); torch.cuda.synchronize()
            for buf in ops._gru_hdr.values():
                st = buf[128:176].view(torch.int64).tolist(); n = max(st[5], 1)
                print(f"  stagger {stag} (stamped: {tb * 1e3 / T:.3f} us/step) bwd phases (cycles/step): sweep {st[0] / n:.0f} | sums + panel {st[1] / n:.0f}"
                      f" | barrier {st[2] / n:.0f} | MFMA + publish {st[3] / n:.0f} | re-polls {st[4] / n:.2f}")
print("status", ops.gru_status())
