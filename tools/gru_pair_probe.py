"""Can two recurrences share the chip?  (profiling aid for the layer-1 / layer-2 pipeline)
Times one forward / backward recurrence launch at B clips, two launches on two streams (own hand-off panels, XCD rotation
0 / 4), and the pair with a third stream of projection GEMMs (the lean kernels: 8 clips per chain, 20 workgroups per 8 clips;
the wide-chain kernels: 16 clips per chain, library option gru_bg = 16)."""
import os, sys, torch
sys.path.insert(0, '.')
from cruse_amd import ops
T, H, G = 401, 640, 1
Hg = H // G
torch.manual_seed(0)
ws = [(torch.randn(3 * Hg, Hg) / 25).cuda()]; bs = [torch.zeros(3 * Hg).cuda()]
s2, s3 = torch.cuda.Stream(), torch.cuda.Stream()


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def run(B, label, nch=8):
    gi = [(0.5 * torch.randn(B, T, 3 * H)).cuda() for _ in range(2)]
    dout = [(0.1 * torch.randn(B, T, H)).cuda() for _ in range(2)]
    outs = [ops.gru_seq_fwd(gi[i], ws, bs, B, T, G, Hg, "bf16", slot=i, xcd_rot=4 * i) for i in range(2)]
    # the projection GEMM of one time chunk (rows B*T/nch, N = 3*Hg, K = Hg)
    x_bf = (0.5 * torch.randn(B * T * H)).cuda().to(torch.bfloat16)
    w_t = ops.ktile_bf16(ws[0], 3 * Hg, Hg)
    n = T // nch
    torch.cuda.synchronize()

    def gemms():
        for j in range(nch):
            ops.gemm_bf16_nt_seg(B * n, 3 * Hg, Hg, x_bf, None, 0, H, w_t, None, 0, 64, gi[1], 0, 3 * H, (n, T, j * n), b_kstride=3 * Hg * 64)

    def fwd(i): ops.gru_seq_fwd(gi[i], ws, bs, B, T, G, Hg, "bf16", out=outs[i], slot=i, xcd_rot=4 * i)
    def bwd(i): ops.gru_seq_bwd(dout[i], ws, outs[i][1], outs[i][3], B, T, G, Hg, "bf16", slot=i, xcd_rot=4 * i)

    def par(*fs):
        def f():
            cur = torch.cuda.current_stream()
            for st in (s2, s3): st.wait_stream(cur)
            fs[0]()
            for st, fn in zip((s2, s3), fs[1:]):
                with torch.cuda.stream(st):
                    fn()
            for st in (s2, s3): cur.wait_stream(st)
        return f
    r = dict(f1=timeit(lambda: fwd(0)), f2=timeit(par(lambda: fwd(0), lambda: fwd(1))), f2g=timeit(par(lambda: fwd(0), lambda: fwd(1), gemms)),
             g=timeit(gemms), b1=timeit(lambda: bwd(0)), b2=timeit(par(lambda: bwd(0), lambda: bwd(1))),
             b2g=timeit(par(lambda: bwd(0), lambda: bwd(1), gemms)))
    print(f"{label} B={B}: fwd one {r['f1']:.0f} us ({r['f1']/T:.2f}/step), two {r['f2']:.0f}, two + {nch} chunk GEMMs {r['f2g']:.0f} "
          f"(GEMMs alone {r['g']:.0f}) | bwd one {r['b1']:.0f} ({r['b1']/T:.2f}/step), two {r['b2']:.0f}, two + GEMMs {r['b2g']:.0f}; "
          f"status {ops.gru_status()}", flush=True)


run(32, "lean / reduce-scatter, 8 clips per chain")
ops.set_option("gru_bg", 16)
run(64, "wide chains, 16 clips per chain")
