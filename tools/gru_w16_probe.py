"""Wide-chain (16 clips per chain) recurrence kernels vs the lean ones: results and step time (library option gru_bg)."""
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


for B in (128, 64, 24, 8):
    gi = (0.5 * torch.randn(B, T, 3 * H)).cuda()
    dout = (0.1 * torch.randn(B, T, H)).cuda()
    h0 = (0.3 * torch.randn(B, H)).cuda()
    res = {}
    for bg in (8, 16):
        ops.set_option("gru_bg", bg)
        f = ops.gru_seq_fwd(gi, ws, bs, B, T, G, Hg, "bf16", h0=h0)
        dh, dgi = ops.gru_seq_bwd(dout, ws, f[1], f[3], B, T, G, Hg, "bf16", an=f[2], want_dgi=True)
        # chunks: 3 forward, 3 backward (last to first)
        fc = ops.gru_seq_fwd(gi, ws, bs, B, T, G, Hg, "bf16", h0=h0, chunk=(0, 100))
        ops.gru_seq_fwd(gi, ws, bs, B, T, G, Hg, "bf16", out=fc, chunk=(100, 200))
        ops.gru_seq_fwd(gi, ws, bs, B, T, G, Hg, "bf16", out=fc, chunk=(300, 101))
        torch.cuda.synchronize()
        tf = timeit(lambda: ops.gru_seq_fwd(gi, ws, bs, B, T, G, Hg, "bf16", out=f))
        tb = timeit(lambda: ops.gru_seq_bwd(dout, ws, f[1], f[3], B, T, G, Hg, "bf16", an=f[2], want_dgi=True, out=(dh, dgi)))
        res[bg] = [t.clone() for t in f] + [dh.clone(), dgi.clone()]
        ck = max(float((a - b).abs().max()) for a, b in zip(f, fc))
        print(f"B={B} clips/chain={bg}: fwd {tf:.0f} us ({tf/T:.2f}/step) bwd {tb:.0f} us ({tb/T:.2f}/step); chunked-vs-whole max diff {ck:.1e}; "
              f"status {ops.gru_status()}", flush=True)
    names = ["h", "coef", "an", "z", "dh", "dgi"]
    print("   16 vs 8 max abs diff:", {n: float((a.float() - b.float()).abs().max()) for n, a, b in zip(names, res[8], res[16])}, flush=True)
