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
for (H, G, B, T) in ((640, 4, 3, 7), (640, 4, 20, 50), (640, 2, 9, 12), (320, 1, 5, 9), (160, 1, 8, 3), (640, 4, 64, 401), (640, 2, 64, 401)):
    Hg = H // G
    torch.manual_seed(B + T + G)
    gi = (0.5 * torch.randn(B, T, 3 * H)).cuda()
    w = [(torch.randn(3 * Hg, Hg) / Hg ** 0.5).cuda() for _ in range(G)]; b = [(0.1 * torch.randn(3 * Hg)).cuda() for _ in range(G)]
    out = {}
    for rd in (0, 1):
        with ops.options(gru_fwd_rd=rd):
            out[rd] = ops.gru_seq_fwd(gi, w, b, B, T, G, Hg, "bf16"); torch.cuda.synchronize()
    same = " ".join(str(bool((x.float() == y.float()).all())) for x, y in zip(out[0], out[1]))
    line = f"H={H} G={G} B={B} T={T}: register-direct == image form: {same} status {ops.gru_status()}"
    if T == 401:
        for rd in (0, 1):
            for d in (0, 4, 8):
                with ops.options(gru_fwd_rd=rd, gru_poll_fwd=d):
                    t = timeit(lambda: ops.gru_seq_fwd(gi, w, b, B, T, G, Hg, "bf16"))
                line += f" | rd={rd} delay {d}: {t * 1e3 / T:.3f}"
    print(line)
