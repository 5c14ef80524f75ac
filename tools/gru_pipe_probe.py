"""Profiling aid: two half-batch recurrences launched concurrently on two streams (cruse_gru_seq_fwd_on / _bwd_on) --
how long each pair takes against one launch alone, for different XCD rotations of the second launch."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cruse_amd import ops


def main():
    dev = "cuda"
    T, H, g = 401, 640, 1
    Hg = H // g
    torch.manual_seed(0)
    w_hh = [torch.randn(3 * Hg, Hg, device=dev) * 0.05 for _ in range(g)]
    b_hh = [torch.zeros(3 * Hg, device=dev) for _ in range(g)]
    s2 = torch.cuda.Stream()

    def run(B, rots, bwd=False):
        gi = [torch.randn(B, T, 3 * H, device=dev) * 0.3 for _ in rots]
        saved = [ops.gru_seq_fwd(gi[k], w_hh, b_hh, B, T, g, Hg, "bf16", slot=k, xcd_rot=r) for k, r in enumerate(rots)]
        dout = [torch.randn(B, T, H, device=dev) for _ in rots]
        torch.cuda.synchronize()

        def once():
            main = torch.cuda.current_stream()
            ev = torch.cuda.Event(); ev.record(main)
            for k, r in enumerate(rots):
                st = main if k == 0 else s2
                if k:
                    st.wait_event(ev)
                with torch.cuda.stream(st):
                    if bwd:
                        ops.gru_seq_bwd(dout[k], w_hh, saved[k][1], saved[k][3], B, T, g, Hg, "bf16", slot=k, xcd_rot=r)
                    else:
                        ops.gru_seq_fwd(gi[k], w_hh, b_hh, B, T, g, Hg, "bf16", slot=k, xcd_rot=r)
            if len(rots) > 1:
                main.wait_stream(s2)
        for _ in range(3):
            once()
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            once()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / 10 * 1e3

    for bwd in (False, True):
        tag = "bwd" if bwd else "fwd"
        print(f"{tag}: B=64 one launch {run(64, [0], bwd):7.1f} us | B=32 one launch {run(32, [0], bwd):7.1f} us | B=32 x2 rot (0,0) "
              f"{run(32, [0, 0], bwd):7.1f} us | rot (0,4) {run(32, [0, 4], bwd):7.1f} us | rot (0,2) {run(32, [0, 2], bwd):7.1f} us")
    ops.check_gru_status()


if __name__ == "__main__":
    main()
