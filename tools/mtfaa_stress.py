"""BASELINE config 5 (alt-model stress of the STFT + conv kernels): the blocks model/mtfaa.py actually defines --
STFT.transform -> PhaseEncoder -> 6 x TFCM_Block (dilations 1..32) -- forward + backward on B clips of `seconds` s.
The file has no axial attention and its `Banks` needs the absent `spafe`, so this is all there is to
stress.  Prints frames/s (10 ms hop) of fwd+bwd and the tensor traffic rate.  --dtype f16 (the config's stated dtype): the
TFCM stack keeps its activations in f16 (pointwise convolutions on v_mfma_f32_16x16x32_f16, generic.hip); f32: all f32."""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--dtype", default="f16", choices=["f16", "f32"])
    a = ap.parse_args()
    from model import mtfaa as M
    from cruse_amd.nn_generic import to_f16, to_f32
    torch.manual_seed(0)
    stft = M.STFT(320, 160, 320, "hann")
    pe = M.PhaseEncoder(4, 1).cuda()                 # 1 signal: [B,2,F,T] -> |.|^0.5 [B,2,F,T]
    tfcm = M.TFCM(24, (3, 3), 6).cuda()
    L = int(a.seconds * 16000)
    x = (0.1 * torch.randn(a.batch, L)).cuda()
    params = [p for m in (pe, tfcm) for p in m.parameters()]

    def step():
        c = stft.transform(x)                        # [B,2,161,T]
        amp = pe([c])                                # [B,2,161,T]
        h = torch.cat([amp] * 12, dim=1)             # [B,24,161,T] (channel plumbing)
        if a.dtype == "f16":
            y = to_f32(tfcm(to_f16(h)))
        else:
            y = tfcm(h)
        for p in params:
            p.grad = None
        # a mean over 1.2e7 elements makes gradients of ~1e-7, below f16's normal range: the usual loss scale of f16 training
        (y.square().mean() * (65536.0 if a.dtype == "f16" else 1.0)).backward()
        return y
    y = step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        y = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    T = y.shape[-1]
    # tensor traffic of the TFCM stack, counted per block: forward 5 kernels (conv, BN+PReLU, depthwise, BN+PReLU, conv + add)
    # read + write one [B,24,161,T] tensor each (+ the BN statistics passes: 2 reads); backward ~2.5 x that
    esz = 2 if a.dtype == "f16" else 4
    tensor = a.batch * 24 * 161 * T * esz
    approx = 6 * (5 * 2 + 2 + 2) * tensor * 3.5
    print(f"mtfaa blocks fwd+bwd ({a.dtype} activations): B={a.batch} x {a.seconds:g} s, activations [B,24,161,{T}]: "
          f"{dt * 1e3:.1f} ms/step, {a.batch * T / dt:,.0f} frames/s, ~{approx / dt / 1e9:,.0f} GB/s of tensor traffic "
          f"(one [B,24,161,T] tensor = {tensor / 1e6:.0f} MB); finite={bool(torch.isfinite(y).all())}")


if __name__ == "__main__":
    main()
