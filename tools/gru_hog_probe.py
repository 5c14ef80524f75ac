"""Profiling aid: the backward / forward recurrence alone and beside an HBM-streaming co-runner on another stream
(a chain of large device copies) -- how much of the in-step slowdown a memory hog reproduces.  CRUSE_GRU_DBG=7 drops the
backward kernel's operand streams (dout, z, coef rows), which separates their latency from the hand-off's."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cruse_amd import ops


def main():
    dev = "cuda"
    B, T, H, g = 64, 401, 640, 1
    Hg = H // g
    torch.manual_seed(0)
    w_hh = [torch.randn(3 * Hg, Hg, device=dev) * 0.05]
    b_hh = [torch.zeros(3 * Hg, device=dev)]
    gi = torch.randn(B, T, 3 * H, device=dev) * 0.3
    h, coef, an, z = ops.gru_seq_fwd(gi, w_hh, b_hh, B, T, g, Hg, "bf16")
    dout = torch.randn(B, T, H, device=dev)
    hog_a = torch.empty(256 << 20, dtype=torch.uint8, device=dev); hog_b = torch.empty_like(hog_a)
    s2 = torch.cuda.Stream()

    def timed(fn, hog, n=6):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            if hog:
                with torch.cuda.stream(s2):
                    for _ in range(hog):
                        hog_b.copy_(hog_a)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        return sorted(ts)[len(ts) // 2]

    fwd = lambda: ops.gru_seq_fwd(gi, w_hh, b_hh, B, T, g, Hg, "bf16")
    bwd = lambda: ops.gru_seq_bwd(dout, w_hh, coef, z, B, T, g, Hg, "bf16")
    for name, fn in (("fwd", fwd), ("bwd", bwd)):
        print(f"{name}: alone {timed(fn, 0):7.1f} us | beside a 512 MB/copy HBM hog {timed(fn, 12):7.1f} us")
    ops.check_gru_status()


if __name__ == "__main__":
    main()
