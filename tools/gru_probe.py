"""Within-process A/B of the GRU recurrence kernels (profiling aid).  library option gru_dbg (argv): 0 = shipped,
9 = force write-through publishes, 1 = no tag waits (wrong results), 2 = also no MFMA."""
import os, sys, torch
sys.path.insert(0, '.')
from cruse_amd import ops
B, T, H = int(os.environ.get("PB", 64)), 401, 640
G = int(os.environ.get("PG", 1))
Hg = H // G
torch.manual_seed(0)
gi = (0.5 * torch.randn(B, T, 3 * H)).cuda()
ws = [(torch.randn(3 * Hg, Hg) / 25).cuda() for _ in range(G)]; bs = [torch.zeros(3 * Hg).cuda() for _ in range(G)]
dout = (0.1 * torch.randn(B, T, H)).cuda()
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
h, coef, an, z = ops.gru_seq_fwd(gi, ws, bs, B, T, G, Hg, "bf16")
modes = sys.argv[1:] or ["0", "9"]
for rnd in range(3):
    out = []
    for m in modes:
        ops.set_option("gru_dbg", int(m))
        tf = timeit(lambda: ops.gru_seq_fwd(gi, ws, bs, B, T, G, Hg, "bf16"))
        if int(m) == 32:                       # phase stamps of the lean forward kernel (workgroup 0 of chain 0)
            ops.gru_seq_fwd(gi, ws, bs, B, T, G, Hg, "bf16"); torch.cuda.synchronize()
            for buf in ops._gru_hdr.values():
                if True:
                    st = buf[64:112].view(torch.int64).tolist()
                    n = max(st[5], 1)
                    tot = sum(st[:4])
                    print(f"   lean fwd phases (s_memtime ticks per step; {tf*1e3/T:.2f} us/step measured => {tf*1e3/T/(tot/n)*1e3:.1f} ns per tick): "
                          f"sweep until tags match {st[0]/n:.1f} | LDS write + barrier A {st[1]/n:.1f} | MFMA + red + barrier B {st[2]/n:.1f} | "
                          f"red read + gates + publish {st[3]/n:.1f} | rest (saves, gi) {tf*1e3/T*0 + 0:.0f}; re-polls per step {st[4]/n:.2f}")
        tb = timeit(lambda: ops.gru_seq_bwd(dout, ws, coef, z, B, T, G, Hg, "bf16"))
        if int(m) in (32, 33, 34, 35):
            for buf in ops._gru_hdr.values():
                if True:
                    st = buf[128:176].view(torch.int64).tolist()
                    n = max(st[5], 1)
                    print(f"   rs bwd phases (cycles per step; {tb*1e3/T:.2f} us/step): sweep until tags match {st[0]/n:.0f} | sum + dh + panel write {st[1]/n:.0f} | "
                          f"barrier {st[2]/n:.0f} | fragment reads + MFMA + publishes {st[3]/n:.0f}; re-polls per step {st[4]/n:.2f}")
        out.append(f"dbg={m}: fwd {tf*1e3/T:.2f} bwd {tb*1e3/T:.2f} us/step")
    print(f"B={B} G={G} round {rnd}: " + " | ".join(out), "status", ops.gru_status())
