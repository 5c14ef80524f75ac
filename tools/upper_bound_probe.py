"""Profiling aid: what would the bench step take if a family of in-between kernels around the GRU recurrences were FREE?
The named cruse_amd.ops entry points are replaced by stubs that return the buffers of their first (real) call without
launching anything (results are garbage; the guarded Adam skips) -- an upper bound for any scheme that hides those kernels
behind the recurrences (time-chunk pipelining)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    from cruse_amd import ops
    from cruse_amd.data import synth_batch
    from cruse_amd.engine import TrainEngine
    from cruse_amd.model.cruse_net import unet_2
    torch.manual_seed(0)
    m = unet_2(rnn_groups=1, precision="bf16").cuda()
    eng = TrainEngine(m, use_graph=False)
    pool = [synth_batch(64, 64000, "cuda", 1234 + s) for s in range(4)]
    variants = {"baseline": [], "fwd gate GEMMs free": ["gemm_f16_nt", "gemm_bf16x3_nt"], "fwd gate GEMMs + ln_fwd free": ["gemm_f16_nt", "gemm_bf16x3_nt", "ln_fwd"],
                "gate grads free": ["gru_gate_grads_bf16"], "dX GEMMs free": ["gemm_bf16_nt"], "ln_bwd free": ["ln_bwd"],
                "GRU dW GEMMs free": ["gemm_bf16_nt_cat"], "transposes free": ["transpose_bf16"],
                "bwd gate grads + dX + ln_bwd free": ["gru_gate_grads_bf16", "gemm_bf16_nt", "ln_bwd"],
                # conv stack
                "BatchNorm backward free": ["bn_act_bwd"], "conv weight gradients free": ["conv_wgrad"],
                "skip convs (fwd + dgrad) free": ["conv_gather"],
                "dW + wgrad free": ["gemm_bf16_nt_cat", "conv_wgrad"],
                "recurrences free": ["gru_seq_fwd", "gru_seq_bwd"]}
    if len(sys.argv) > 1:
        variants = {k: v for k, v in variants.items() if k == "baseline" or any(a in k for a in sys.argv[1:])}
    for name, fns in variants.items():
        saved, cache = {}, {}
        for fn in fns:
            orig = getattr(ops, fn)
            saved[fn] = orig

            def stub(*a, _fn=fn, _orig=orig, **k):
                key = (_fn, len(cache.get(_fn, [])) if not cache.get((_fn, "done")) else None)
                lst = cache.setdefault(_fn, [])
                if not cache.get((_fn, "done")):
                    lst.append(_orig(*a, **k))
                    return lst[-1]
                i = cache[(_fn, "i")] = (cache.get((_fn, "i"), -1) + 1) % len(lst)
                return lst[i]
            setattr(ops, fn, stub)
        eng.step(*pool[0]); torch.cuda.synchronize()          # first (real) calls fill the cache
        for fn in fns:
            cache[(fn, "done")] = True
            # the dW GEMMs also go through gemm_bf16_nt: only the dX calls (accumulate / M == rows) are skipped by position
        for _ in range(3):
            eng.step(*pool[1])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 30
        for s in range(n):
            for fn in fns:
                cache[(fn, "i")] = -1
            eng.step(*pool[s % 4])
        torch.cuda.synchronize()
        print(f"{name:40s} {(time.perf_counter() - t0) / n * 1e3:.3f} ms/step")
        for fn, orig in saved.items():
            setattr(ops, fn, orig)


if __name__ == "__main__":
    main()
