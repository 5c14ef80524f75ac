"""The tag-free hand-off recurrence kernels (csrc/gru_tf.hip, library option gru_tf) against the tagged lean / reduce-scatter
kernels of gru.hip: results (both are bf16-mode kernels with the same rounding points; they differ in the f32 summation order),
distance of each from the exact-f32 kernels, time per dependent step.  usage: python tools/gru_tf_probe.py [quick]"""
import sys
import torch
sys.path.insert(0, '.')
from cruse_amd import ops


def rel(a, b):
    a = a.double().flatten(); b = b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def case(B, T, H, G, time_it=False, scale=1.0):
    Hg = H // G
    torch.manual_seed(B + T + H)
    gi = (0.5 * torch.randn(B, T, 3 * H)).cuda()
    w = [(torch.randn(3 * Hg, Hg) / Hg ** 0.5).cuda() for _ in range(G)]
    b = [(0.1 * torch.randn(3 * Hg)).cuda() for _ in range(G)]
    dout = (scale * torch.randn(B, T, H)).cuda()
    ref32 = ops.gru_seq_fwd(gi, w, b, B, T, G, Hg, "f32")
    ref32_b = ops.gru_seq_bwd(dout, w, ref32[1], ref32[3], B, T, G, Hg, "f32")
    out = {}
    for tf in (0, 1):
        with ops.options(gru_tf=tf):
            f = ops.gru_seq_fwd(gi, w, b, B, T, G, Hg, "bf16")
            bw = ops.gru_seq_bwd(dout, w, f[1], f[3], B, T, G, Hg, "bf16", an=f[2], want_dgi=True)
            torch.cuda.synchronize()
            out[tf] = (f, bw)
    st = ops.gru_status()
    f0, b0 = out[0]; f1, b1 = out[1]
    print(f"B={B} T={T} H={H} G={G} status={st}")
    print(f"  fwd h:  tf vs lean {rel(f1[0], f0[0]):.2e} | lean vs f32 {rel(f0[0], ref32[0]):.2e} | tf vs f32 {rel(f1[0], ref32[0]):.2e}"
          f" | finite {bool(torch.isfinite(f1[0]).all())}")
    for i, nm in ((1, "coef"), (2, "an"), (3, "z")):
        print(f"      {nm}: tf vs lean {rel(f1[i].float(), f0[i].float()):.2e}")
    # backward of the SAME saved tensors (so that only the backward kernels differ)
    with ops.options(gru_tf=1):
        bx = ops.gru_seq_bwd(dout, w, f0[1], f0[3], B, T, G, Hg, "bf16", an=f0[2], want_dgi=True)
    torch.cuda.synchronize()
    ref_b_same = ops.gru_seq_bwd(dout, w, ref32[1], ref32[3], B, T, G, Hg, "f32")
    print(f"  bwd dh (same saves): tf vs rs {rel(bx[0], b0[0]):.2e} | rs vs f32 {rel(b0[0], ref_b_same):.2e} | tf vs f32 {rel(bx[0], ref_b_same):.2e}"
          f" | dgi tf vs rs {rel(bx[1].float(), b0[1].float()):.2e} | finite {bool(torch.isfinite(bx[0]).all())}")
    if time_it:
        for tf in (0, 1, 0, 1):
            with ops.options(gru_tf=tf):
                tfw = timeit(lambda: ops.gru_seq_fwd(gi, w, b, B, T, G, Hg, "bf16"))
                tbw = timeit(lambda: ops.gru_seq_bwd(dout, w, f0[1], f0[3], B, T, G, Hg, "bf16", an=f0[2], want_dgi=False))
                tbw2 = timeit(lambda: ops.gru_seq_bwd(dout, w, f0[1], f0[3], B, T, G, Hg, "bf16", an=f0[2], want_dgi=True))
            print(f"  gru_tf={tf}: fwd {tfw * 1e3:.0f} us ({tfw * 1e3 / T:.3f} us/step) | bwd {tbw * 1e3:.0f} us ({tbw * 1e3 / T:.3f} us/step)"
                  f" | bwd + dgi {tbw2 * 1e3:.0f} us")
    return st


def stamps(B=64, T=401, H=640):
    gi = (0.5 * torch.randn(B, T, 3 * H)).cuda()
    w = [(torch.randn(3 * H, H) / 25).cuda()]; b = [torch.zeros(3 * H).cuda()]
    dout = (0.1 * torch.randn(B, T, H)).cuda()
    h, coef, an, z = ops.gru_seq_fwd(gi, w, b, B, T, 1, H, "bf16")
    for tf in (0, 1):
        with ops.options(gru_tf=tf, gru_dbg=32):
            tw = timeit(lambda: ops.gru_seq_fwd(gi, w, b, B, T, 1, H, "bf16")); torch.cuda.synchronize()
            print(f"  (stamped forward instance: {tw * 1e3 / T:.3f} us/step)")
            for buf in ops._gru_hdr.values():
                if True:
                    st = buf[64:112].view(torch.int64).tolist(); n = max(st[5], 1)
                    print(f"  gru_tf={tf} fwd phases (cycles/step): sweep {st[0] / n:.0f} | LDS image + barrier {st[1] / n:.0f} | MFMA phase {st[2] / n:.0f}"
                          f" | gates + publish {st[3] / n:.0f} | re-polls {st[4] / n:.2f}")
            ops.gru_seq_bwd(dout, w, coef, z, B, T, 1, H, "bf16"); torch.cuda.synchronize()
            for buf in ops._gru_hdr.values():
                if True:
                    st = buf[128:176].view(torch.int64).tolist(); n = max(st[5], 1)
                    print(f"  gru_tf={tf} bwd phases (cycles/step): sweep {st[0] / n:.0f} | sums + dh + panel {st[1] / n:.0f} | barrier {st[2] / n:.0f}"
                          f" | MFMA + publishes {st[3] / n:.0f} | re-polls {st[4] / n:.2f}")


if __name__ == "__main__":
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    bad = 0
    bad += case(5, 9, 640, 1)
    bad += case(11, 37, 640, 4)
    bad += case(16, 21, 640, 2)
    bad += case(64, 401, 640, 1, time_it=True, scale=0.1)
    if not quick:
        bad += case(64, 401, 640, 4, time_it=True, scale=0.1)
        bad += case(64, 401, 640, 1, scale=1e4)        # large cotangents: the 2^-64 exchange scale
        bad += case(64, 401, 640, 1, scale=1e-12)
    stamps()
    print("status sum", bad)
