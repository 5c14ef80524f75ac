"""Engine steps for tools/engine_guard_run.py (every torch tensor end-aligned in its own hipMalloc block: an access past a tensor faults):
one and three clips of an odd length, g = 1 and g = 4, bf16 and f32 modes, every loss the engine knows -- eager launches (no HIP graphs under
the guard allocator).  Prints "guard steps ok" when no kernel left its tensors.  Run by tests/test_gpu_guard.py in a subprocess."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cruse_amd.data import synth_batch            # noqa: E402
from cruse_amd.engine import TrainEngine          # noqa: E402
from cruse_amd.model.cruse_net import unet_2      # noqa: E402

dev = torch.device("cuda", 0)
n = 0
for g in (1, 4):
    for prec in ("bf16", "f32"):
        for B, L in ((1, 160 * 9 + 37), (3, 160 * 21 - 1)):
            for loss in (("wo_male", "wo_male_df", "si_snr") if prec == "bf16" else ("wo_male",)):
                torch.manual_seed(0)
                m = unet_2(rnn_groups=g, precision=prec).to(dev)
                e = TrainEngine(m, lr=1e-3, use_graph=False, loss=loss)
                noisy, clean = synth_batch(B, L, dev, 5 + n)
                for _ in range(2):
                    v = e.loss_value(e.step(noisy, clean))
                torch.cuda.synchronize()
                assert v == v and abs(v) < 1e6, (g, prec, B, L, loss, v)
                n += 1
print(f"guard steps ok ({n} engine configurations)")
