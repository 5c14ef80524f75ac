"""Which piece of the backward wavefront breaks hipStreamEndCapture?  Captures pieces into torch.cuda.graph one by one."""
import sys, torch, faulthandler
sys.path.insert(0, '.')
faulthandler.enable()
from cruse_amd import ops
case = sys.argv[1]
T, H, G, B = 201, 640, 1, 16
Hg = H
torch.manual_seed(0)
ws = [(torch.randn(3 * Hg, Hg) / 25).cuda()]; bs = [(0.1 * torch.randn(3 * Hg)).cuda()]
gi = (0.5 * torch.randn(B, T, 3 * H)).cuda(); dout = (0.1 * torch.randn(B, T, H)).cuda()
f = ops.gru_seq_fwd(gi, ws, bs, B, T, G, Hg, "bf16")
dh = torch.empty(B, T, H).cuda(); dgi = ops.dgi_buffer(B * T, G, Hg, "cuda")
dl = torch.empty(B, T, H).cuda(); dl2 = torch.empty(B, T, H).cuda()
w_t = ops.transpose_bf16(ws[0], 3 * Hg, Hg)
gam = torch.ones(H).cuda(); dg = torch.zeros(H).cuda(); db = torch.zeros(H).cuda()
m1 = torch.zeros(B * T).cuda(); s1 = torch.ones(B * T).cuda()
cuts = [(0, 50), (50, 50), (100, 50), (150, 51)]
s2, s3 = torch.cuda.Stream(), torch.cuda.Stream()


def ev(st):
    e = torch.cuda.Event(); e.record(st); return e


def body():
    main = torch.cuda.current_stream()
    if case == "one":
        ops.gru_step_ws_clear(B, G, Hg, "cuda")
        ops.gru_seq_bwd(dout, ws, f[1], f[3], B, T, G, Hg, "bf16", an=f[2], want_dgi=True, out=(dh, dgi), wide=True, slot=ops.STEP_SLOT0, zeroed=True, seq=0)
    elif case == "one8":
        ops.gru_seq_bwd(dout, ws, f[1], f[3], B, T, G, Hg, "bf16", an=f[2], want_dgi=True, out=(dh, dgi))
    elif case == "chunks":
        ops.gru_step_ws_clear(B, G, Hg, "cuda")
        for i, c in enumerate(reversed(cuts)):
            ops.gru_seq_bwd(dout, ws, f[1], f[3], B, T, G, Hg, "bf16", an=f[2], want_dgi=True, out=(dh, dgi), chunk=c, wide=True, slot=ops.STEP_SLOT0, zeroed=True, seq=i)
    elif case in ("seg", "lnb"):
        for c in cuts:
            if case == "seg":
                ops.gemm_bf16_nt_seg(B * c[1], Hg, 3 * Hg, dgi, None, 0, 3 * H, w_t, None, 0, 64, dl, 0, H, (c[1], T, c[0]), b_kstride=Hg * 64)
            else:
                ops.ln_bwd(dout, f[0], m1, s1, gam, B * c[1], H, 1, dg, db, seg=(c[1], T, c[0]), out=dl2)
    elif case == "wave":
        ops.gru_step_ws_clear(B, G, Hg, "cuda")
        st = ev(main); s2.wait_event(st); s3.wait_event(st)
        for i, c in enumerate(reversed(cuts)):
            ops.gru_seq_bwd(dout, ws, f[1], f[3], B, T, G, Hg, "bf16", an=f[2], want_dgi=True, out=(dh, dgi), chunk=c, wide=True, slot=ops.STEP_SLOT0, zeroed=True, seq=i)
            e_r = ev(main)
            with torch.cuda.stream(s3):
                s3.wait_event(e_r)
                ops.gemm_bf16_nt_seg(B * c[1], Hg, 3 * Hg, dgi, None, 0, 3 * H, w_t, None, 0, 64, dl, 0, H, (c[1], T, c[0]), b_kstride=Hg * 64)
                ops.ln_bwd(dl, f[0], m1, s1, gam, B * c[1], H, 1, dg, db, seg=(c[1], T, c[0]), out=dl2)
                e_q = ev(s3)
            with torch.cuda.stream(s2):
                s2.wait_event(e_q)
                ops.gru_seq_bwd(dl2, ws, f[1], f[3], B, T, G, Hg, "bf16", an=f[2], want_dgi=True, out=(dh, dgi), chunk=c, wide=True, slot=ops.STEP_SLOT0 + 1, zeroed=True, seq=i, xcd_rot=4)
        main.wait_stream(s2); main.wait_stream(s3)


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    body()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print(case, "eager ok, status", ops.gru_status(), flush=True)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    body()
print(case, "captured", flush=True)
g.replay(); torch.cuda.synchronize()
print(case, "replayed, status", ops.gru_status(), flush=True)
