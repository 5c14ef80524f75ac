import sys, torch
sys.path.insert(0, '.')
from cruse_amd import ops
torch.manual_seed(0)
for (rows, G, Hg) in ((25664, 1, 640), (1000, 2, 128), (64 * 3 + 5, 1, 64)):
    H = G * Hg
    dh = torch.randn(rows, H, device="cuda") * 0.1
    coef = torch.randn(rows, G, 3, Hg, device="cuda").to(torch.bfloat16)
    an = torch.randn(rows, H, device="cuda")
    bi = [torch.zeros(3 * Hg, device="cuda") for _ in range(G)]; bh = [torch.zeros(3 * Hg, device="cuda") for _ in range(G)]
    dgi, dgT, ldT = ops.gru_gate_grads_bf16(dh, coef, an, rows, G, Hg, bi, bh)
    for i in range(G):
        W = torch.randn(3 * Hg, Hg, device="cuda") / Hg ** 0.5
        w_t = ops.transpose_bf16(W, 3 * Hg, Hg)
        for acc in (False, True):
            base = torch.randn(rows, H, device="cuda")
            c0 = base.clone(); c1 = base.clone()
            ops.gemm_bf16_nt(rows, Hg, w_t.shape[0] * 64, dgi, i * 3 * Hg, 3 * H, w_t, 0, 64, c0, i * Hg, H, accumulate=acc, b_kstride=Hg * 64)
            ops.gemm_bf16_nt_atr(rows, Hg, 3 * Hg, dgT, 4 * i * Hg * 64, G * 4 * Hg * 64, ldT // 64, w_t, 0, 64, c1, i * Hg, H, accumulate=acc, b_kstride=Hg * 64)
            torch.cuda.synchronize()
            sl = slice(i * Hg, (i + 1) * Hg)
            print(f"rows {rows} G {G} Hg {Hg} group {i} acc {acc}: equal {torch.equal(c0, c1)} max diff {float((c0 - c1).abs().max()):.3e} ref max {float(c0[:, sl].abs().max()):.3e}", flush=True)
    if rows > 20000:
        for name, fn in (("nt ", lambda: ops.gemm_bf16_nt(rows, Hg, w_t.shape[0] * 64, dgi, 0, 3 * H, w_t, 0, 64, c0, 0, H, b_kstride=Hg * 64)),
                         ("atr", lambda: ops.gemm_bf16_nt_atr(rows, Hg, 3 * Hg, dgT, 0, G * 4 * Hg * 64, ldT // 64, w_t, 0, 64, c1, 0, H, b_kstride=Hg * 64))):
            for _ in range(3): fn()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            print(name, f"{e0.elapsed_time(e1) / 20 * 1e3:.1f} us", flush=True)
