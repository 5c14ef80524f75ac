"""Wide-chain recurrences (gru_w16.hip: 16 clips per chain, 80 workgroups at B = 64 / Hg = 640) against the chains of 8, at the bench
shape: results, time per dependent step alone, two wide launches side by side (the two GGRU layers: slots / XCD halves), the pair
beside chunk projections, phase stamps (gru_dbg = 32) and the first-poll delays.  usage: python tools/gru_wide_probe.py [quick]"""
import sys, torch
sys.path.insert(0, '.')
from cruse_amd import ops
T, H, G, B = 401, 640, 1, 64
Hg = H // G
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
torch.manual_seed(0)
ws = [(torch.randn(3 * Hg, Hg) / 25).cuda()]; bs = [(0.1 * torch.randn(3 * Hg)).cuda()]
s2, s3 = torch.cuda.Stream(), torch.cuda.Stream()


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def rel(a, b):
    return float((a.float() - b.float()).norm() / b.float().norm())


gi = [(0.5 * torch.randn(B, T, 3 * H)).cuda() for _ in range(2)]
dout = [(0.1 * torch.randn(B, T, H)).cuda() for _ in range(2)]
lean = ops.gru_seq_fwd(gi[0], ws, bs, B, T, G, Hg, "bf16")
wide = ops.gru_seq_fwd(gi[0], ws, bs, B, T, G, Hg, "bf16", wide=True)
torch.cuda.synchronize()
print("fwd wide == lean:", {n: bool(torch.equal(a, b)) for n, a, b in zip(("h", "coef", "an", "z"), wide, lean)}, "status", ops.gru_status(), flush=True)
dl = ops.gru_seq_bwd(dout[0], ws, lean[1], lean[3], B, T, G, Hg, "bf16", an=lean[2], want_dgi=True)
dw = ops.gru_seq_bwd(dout[0], ws, lean[1], lean[3], B, T, G, Hg, "bf16", an=lean[2], want_dgi=True, wide=True)
dx = ops.gru_seq_bwd(dout[0], ws, lean[1].float(), lean[3], B, T, G, Hg, "f32")
torch.cuda.synchronize()
print(f"bwd dh: wide vs 8 {rel(dw[0], dl[0]):.2e}, wide vs f32 {rel(dw[0], dx):.2e}, 8 vs f32 {rel(dl[0], dx):.2e}; dgi wide vs 8 {rel(dw[1], dl[1]):.2e}; "
      f"status {ops.gru_status()}", flush=True)
# chunks on one scratch
cuts = [(0, 100), (100, 100), (200, 100), (300, 101)]
out = None
for i, c in enumerate(cuts):
    out = ops.gru_seq_fwd(gi[0], ws, bs, B, T, G, Hg, "bf16", out=out, chunk=c, wide=True)
bo = (torch.zeros_like(dw[0]), torch.zeros_like(dw[1]))
for i, c in enumerate(reversed(cuts)):
    ops.gru_seq_bwd(dout[0], ws, lean[1], lean[3], B, T, G, Hg, "bf16", an=lean[2], want_dgi=True, out=bo, chunk=c, wide=True)
torch.cuda.synchronize()
print("chunked: fwd h equal", bool(torch.equal(out[0], lean[0])), "bwd dh equal", bool(torch.equal(bo[0], dw[0])),
      "dgi equal", bool(torch.equal(bo[1].view(torch.int16), dw[1].view(torch.int16))), "status", ops.gru_status(), flush=True)

outs = [ops.gru_seq_fwd(gi[i], ws, bs, B, T, G, Hg, "bf16", slot=i, xcd_rot=4 * i, wide=True) for i in range(2)]
dhs = [ops.gru_seq_bwd(dout[i], ws, outs[i][1], outs[i][3], B, T, G, Hg, "bf16", slot=i, xcd_rot=4 * i, wide=True) for i in range(2)]
dgs = [ops.gru_seq_bwd(dout[i], ws, outs[i][1], outs[i][3], B, T, G, Hg, "bf16", slot=i, xcd_rot=4 * i, wide=True, an=outs[i][2], want_dgi=True) for i in range(2)]
x_bf = (0.5 * torch.randn(B * T * H)).cuda().to(torch.bfloat16)
w_t = ops.ktile_bf16(ws[0], 3 * Hg, Hg)
gsc = torch.empty(B, T, 3 * H, device="cuda")
torch.cuda.synchronize()


def gemms(nch=8):
    n = T // nch
    for j in range(nch):
        ops.gemm_bf16_nt_seg(B * n, 3 * Hg, Hg, x_bf, None, 0, H, w_t, None, 0, 64, gsc, 0, 3 * H, (n, T, j * n), b_kstride=3 * Hg * 64)


def fwd(i, wide_=True): ops.gru_seq_fwd(gi[i], ws, bs, B, T, G, Hg, "bf16", out=outs[i], slot=i, xcd_rot=4 * i, wide=wide_)
def bwd(i, wide_=True): ops.gru_seq_bwd(dout[i], ws, outs[i][1], outs[i][3], B, T, G, Hg, "bf16", slot=i, xcd_rot=4 * i, wide=wide_, out=dhs[i])
def bwdg(i, wide_=True): ops.gru_seq_bwd(dout[i], ws, outs[i][1], outs[i][3], B, T, G, Hg, "bf16", slot=i, xcd_rot=4 * i, wide=wide_, an=outs[i][2], want_dgi=True, out=dgs[i])


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


r = dict(f8=timeit(lambda: fwd(0, False)), b8=timeit(lambda: bwd(0, False)), f16=timeit(lambda: fwd(0)), b16=timeit(lambda: bwd(0)),
         b16g=timeit(lambda: bwdg(0)),
         f2=timeit(par(lambda: fwd(0), lambda: fwd(1))), b2=timeit(par(lambda: bwd(0), lambda: bwd(1))),
         f2g=timeit(par(lambda: fwd(0), lambda: fwd(1), gemms)), b2g=timeit(par(lambda: bwdg(0), lambda: bwdg(1), gemms)), g=timeit(gemms))
print(f"alone: fwd 8-clip {r['f8']:.0f} us ({r['f8']/T:.3f}/step), wide {r['f16']:.0f} ({r['f16']/T:.3f}/step) | bwd 8-clip {r['b8']:.0f} ({r['b8']/T:.3f}/step), "
      f"wide {r['b16']:.0f} ({r['b16']/T:.3f}/step), wide + dgi {r['b16g']:.0f}", flush=True)
print(f"pairs: two wide fwd {r['f2']:.0f} us, + 8 chunk GEMMs {r['f2g']:.0f} (GEMMs alone {r['g']:.0f}) | two wide bwd {r['b2']:.0f}, two (dgi) + GEMMs {r['b2g']:.0f}; "
      f"status {ops.gru_status()}", flush=True)

# chunked launches of one recurrence back to back: the price of re-launching (prologue: weights to registers)
for nch in (4, 8):
    base = T // nch
    cs = [(j * base, base if j + 1 < nch else T - j * base) for j in range(nch)]

    def fchunks():
        for i, c in enumerate(cs):
            ops.gru_seq_fwd(gi[0], ws, bs, B, T, G, Hg, "bf16", out=outs[0], chunk=c, wide=True)

    def bchunks():
        for i, c in enumerate(reversed(cs)):
            ops.gru_seq_bwd(dout[0], ws, outs[0][1], outs[0][3], B, T, G, Hg, "bf16", out=dhs[0], chunk=c, wide=True)
    print(f"{nch} chunk launches back to back: fwd {timeit(fchunks):.0f} us, bwd {timeit(bchunks):.0f} us; status {ops.gru_status()}", flush=True)

if not quick:
    for name, vals, fn in (("gru_poll_fwd16", (0, 2, 4, 6, 8, 12), lambda: fwd(0)), ("gru_poll_bwd16", (0, 3, 5, 8, 12, 16), lambda: bwd(0))):
        res = []
        for v in vals:
            ops.set_option(name, v)
            res.append(f"{v}: {timeit(fn) / T:.3f}")
        ops.set_option(name, None)
        print(f"{name} (us per step):", ", ".join(res), flush=True)
    # phase stamps
    ops.set_option("gru_dbg", 32)
    hdr = ops.gru_header(torch.device("cuda"))
    for nm, fn, off in (("fwd wide", lambda: fwd(0), 8), ("bwd wide", lambda: bwd(0), 16), ("fwd 8-clip", lambda: fwd(0, False), 8), ("bwd 8-clip", lambda: bwd(0, False), 16)):
        fn(); torch.cuda.synchronize()
        st = hdr.view(torch.int64)[off:off + 6].tolist()
        n = max(st[5], 1)
        print(f"{nm}: cycles per step: sweep {st[0] / n:.0f}, barrier {st[1] / n:.0f}, mfma+reduce {st[2] / n:.0f}, gates+publish {st[3] / n:.0f}, re-polls {st[4] / n:.2f}", flush=True)
    ops.set_option("gru_dbg", None)
print("final status", ops.gru_status())
