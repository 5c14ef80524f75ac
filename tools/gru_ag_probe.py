"""All-gather backward recurrence (gru_bwd_ag_kernel, library option gru_bwd_ag) against the reduce-scatter tag-free kernel: results vs the
exact-f32 kernels, time per dependent step, phase stamps.  usage: python tools/gru_ag_probe.py"""
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


for (B, T, scale) in ((3, 7, 1.0), (8, 33, 1e-3), (20, 50, 30.0), (64, 401, 0.1)):
    H = 640; torch.manual_seed(B + T)
    gi = (0.5 * torch.randn(B, T, 3 * H)).cuda()
    w = [(torch.randn(3 * H, H) / H ** 0.5).cuda()]; b = [(0.1 * torch.randn(3 * H)).cuda()]
    dout = (scale * torch.randn(B, T, H)).cuda()
    f32 = ops.gru_seq_fwd(gi, w, b, B, T, 1, H, "f32")
    ref = ops.gru_seq_bwd(dout, w, f32[1], f32[3], B, T, 1, H, "f32")
    f = ops.gru_seq_fwd(gi, w, b, B, T, 1, H, "bf16")
    out = {}
    for ag in (0, 1, 2):
        with ops.options(gru_bwd_ag=ag):
            out[ag] = ops.gru_seq_bwd(dout, w, f[1], f[3], B, T, 1, H, "bf16", an=f[2], want_dgi=True)
            torch.cuda.synchronize()
    same_saves_ref = ops.gru_seq_bwd(dout, w, f32[1], f32[3], B, T, 1, H, "f32")
    print(f"B={B} T={T} scale={scale}: register-direct == LDS-image form: dh {bool((out[2][0] == out[1][0]).all())} dgi {bool((out[2][1] == out[1][1]).all())}")
    print(f"B={B} T={T} scale={scale}: dh ag vs rs {rel(out[1][0], out[0][0]):.2e} | rs vs f32 {rel(out[0][0], same_saves_ref):.2e} | ag vs f32 {rel(out[1][0], same_saves_ref):.2e}"
          f" | dgi ag vs rs {rel(out[1][1].float(), out[0][1].float()):.2e} | finite {bool(torch.isfinite(out[1][0]).all())} status {ops.gru_status()}")
B, T, H = 64, 401, 640
for rnd in range(2):
    for ag in (0, 1, 2):
        for d in (10, 6, 3, 0):
            with ops.options(gru_bwd_ag=ag, gru_poll_bwd=d):
                t = timeit(lambda: ops.gru_seq_bwd(dout, w, f[1], f[3], B, T, 1, H, "bf16"))
                t2 = timeit(lambda: ops.gru_seq_bwd(dout, w, f[1], f[3], B, T, 1, H, "bf16", an=f[2], want_dgi=True))
            print(f"  ag={ag} poll delay {d}: {t * 1e3 / T:.3f} us/step ({t * 1e3:.0f} us) | with dgi {t2 * 1e3 / T:.3f}")
for ag in (0, 1, 2):
    with ops.options(gru_bwd_ag=ag, gru_dbg=32, gru_poll_bwd=10 if ag == 0 else 3):
        tb = timeit(lambda: ops.gru_seq_bwd(dout, w, f[1], f[3], B, T, 1, H, "bf16")); torch.cuda.synchronize()
        for buf in ops._gru_hdr.values():
            st = buf[128:176].view(torch.int64).tolist(); n = max(st[5], 1)
            print(f"  ag={ag} stamped {tb * 1e3 / T:.3f} us/step: phases (cycles/step) {st[0] / n:.0f} | {st[1] / n:.0f} | {st[2] / n:.0f} | {st[3] / n:.0f} | re-polls {st[4] / n:.2f}")
print("status", ops.gru_status())
