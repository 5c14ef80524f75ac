import sys, torch
sys.path.insert(0, '.')
from cruse_amd import ops
def rel(a, b):
    a = a.double().flatten(); b = b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))
def timeit(fn, n=8):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
H = 640
for G in (4, 2):
    Hg = H // G
    for (B, T, scale) in ((3, 7, 1.0), (20, 50, 30.0), (9, 1, 1.0), (5, 2, 1.0), (64, 401, 0.1)):
        torch.manual_seed(B + T + G)
        gi = (0.5 * torch.randn(B, T, 3 * H)).cuda()
        w = [(torch.randn(3 * Hg, Hg) / Hg ** 0.5).cuda() for _ in range(G)]; b = [(0.1 * torch.randn(3 * Hg)).cuda() for _ in range(G)]
        dout = (scale * torch.randn(B, T, H)).cuda()
        f32 = ops.gru_seq_fwd(gi, w, b, B, T, G, Hg, "f32")
        f = ops.gru_seq_fwd(gi, w, b, B, T, G, Hg, "bf16")
        ref = ops.gru_seq_bwd(dout, w, f32[1], f32[3], B, T, G, Hg, "f32")
        out = {}
        for ag in (0, 2):
            with ops.options(gru_bwd_ag=ag):
                out[ag] = ops.gru_seq_bwd(dout, w, f[1], f[3], B, T, G, Hg, "bf16", an=f[2], want_dgi=True)
                plain = ops.gru_seq_bwd(dout, w, f[1], f[3], B, T, G, Hg, "bf16")
                torch.cuda.synchronize()
                assert torch.equal(plain, out[ag][0])
        print(f"G={G} B={B} T={T}: dh ag vs rs {rel(out[2][0], out[0][0]):.2e} | rs vs f32 {rel(out[0][0], ref):.2e} | ag vs f32 {rel(out[2][0], ref):.2e} | dgi {rel(out[2][1].float(), out[0][1].float()):.2e} status {ops.gru_status()}")
    for ag in (0, 2, 0, 2):
        row = []
        for d in (10, 7, 5, 3, 0):
            with ops.options(gru_bwd_ag=ag, gru_poll_bwd=d):
                t = timeit(lambda: ops.gru_seq_bwd(dout, w, f[1], f[3], B, T, G, Hg, "bf16"))
            row.append(f"delay {d}: {t * 1e3 / T:.3f}")
        print(f"  G={G} ag={ag}: " + " | ".join(row) + " us/step")
print(ops.gru_status())
