"""Run-to-run reproducibility of two optimizer steps: N engines built from one seed, stepped on one batch, compared after every
step (gradients of step 1 and 2, parameters after step 2).  Atomics (f64 BatchNorm sums in the conv epilogues, f32 bias sums)
land in a run-dependent order: gradients differ by ~1e-10 absolute, Adam makes up to ~1e-6 of that on weights whose gradient is
noise-level (update = lr g / (|g| + eps)), and in the next step a ReLU mask (f32 mode, rarely) or a bf16 rounding (bf16 mode,
always) flips on a borderline element -- as in any f32 / bf16 training run; nothing here is specific to this library."""
import sys, torch
sys.path.insert(0, '.')
from cruse_amd.config import EngineConfig
from cruse_amd.data import synth_batch
from cruse_amd.engine import TrainEngine
from cruse_amd.model.cruse_net import unet_2


def rel_l2(a, b):
    return float((a.double() - b.double()).norm() / (b.double().norm() + 1e-30))


def trial(nit, B, L, prec, groups, **kw):
    noisy, clean = synth_batch(B, L, "cuda", 9)
    worst = [0.0, 0.0, 0.0]
    nbad = 0
    for _ in range(nit):
        engs = []
        for _e in range(2):
            torch.manual_seed(2)
            engs.append(TrainEngine(unet_2(rnn_groups=groups, precision=prec).cuda(), use_graph=False, config=EngineConfig(**kw)))
        g = [[], []]
        for step in range(2):
            for e in engs:
                e.step(noisy, clean)
                torch.cuda.synchronize()
                g[step].append(e.flat.grads.clone())
        r = [rel_l2(g[0][0], g[0][1]), rel_l2(g[1][0], g[1][1]), rel_l2(engs[0].flat.params, engs[1].flat.params)]
        worst = [max(a, b) for a, b in zip(worst, r)]
        nbad += r[2] > 1e-5
    print(f"B={B} L={L} {prec} g={groups} {kw}: worst rel-L2 over {nit} pairs: step-1 grads {worst[0]:.1e}, step-2 grads {worst[1]:.1e}, "
          f"parameters after 2 steps {worst[2]:.1e}; {nbad} pairs above 1e-5", flush=True)


if __name__ == "__main__":
    for fuse in (True, False):
        trial(12, 4, 8000, "f32", 2, fuse_bn_bwd_stats=fuse)
        trial(12, 4, 8000, "bf16", 1, fuse_bn_bwd_stats=fuse)
    trial(4, 16, 32000, "bf16", 1)
