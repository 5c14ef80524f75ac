"""isolated timings of the NCHW f16 kernels of config 5 on [8,24,161,401]"""
import sys, torch
sys.path.insert(0, '.')
from cruse_amd import ops, nn_generic as G


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


B, C, H, W = 8, 24, 161, 401
x = torch.randn(B, C, H, W, device="cuda").half(); dy = torch.randn(B, C, H, W, device="cuda").half()
w = torch.randn(C, C, 1, 1, device="cuda"); dw = torch.zeros(C, C, 1, 1, device="cuda"); bias = torch.randn(C, device="cuda")
mb = x.numel() * 2 / 1e6
ref = torch.einsum("nahw,nbhw->ab", dy.float(), x.float())
for blocks, run, dbg in ((512, 4096, 0), (256, 4096, 0), (128, 4096, 0), (256, 4096, 1)):
    with ops.options(wgpw_blocks=blocks, wgpw_run=run, wgpw_dbg=dbg):
        dw.zero_()
        G._wgrad_raw(dy, x, dw, 1, 1, (1, 1), (1, 1), 0, 0, 1, 1)
        err = float((dw.view(C, C) - ref).norm() / ref.norm())
        us = timeit(lambda: G._wgrad_raw(dy, x, dw, 1, 1, (1, 1), (1, 1), 0, 0, 1, 1))
    print(f"pw wgrad blocks>={blocks} run<={run} dbg={dbg}: {us:6.1f} us ({2 * mb / us:5.2f} TB/s) err {err:.1e}")
us = timeit(lambda: G._conv_raw(x, w, bias, (H, W), 1, 1, (1, 1), (1, 1), 0, 0, 1, 1, False, C))
print(f"pointwise fwd: {us:6.1f} us ({2 * mb / us:5.2f} TB/s)")
us = timeit(lambda: G._conv_raw(x, w, None, (H, W), 1, 1, (1, 1), (1, 1), 0, 0, 1, 1, True, C))
print(f"pointwise dgrad: {us:6.1f} us ({2 * mb / us:5.2f} TB/s)")
out = torch.zeros(C, device="cuda")
us = timeit(lambda: G._channel_sum(dy, out)); print(f"channel_sum: {us:6.1f} us ({mb / us:5.2f} TB/s)")
