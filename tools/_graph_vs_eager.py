"""Does a HIP-graph replay compute what the eager launches compute?  lr = 0 (parameters never move): capture + replay on batch A,
then replay on batch B; a kernel that ran before its producer would see batch A's tensors."""
import os, sys, torch
os.environ["CRUSE_DBG_KEEP"] = "1"
sys.path.insert(0, '.')
from cruse_amd.config import EngineConfig
from cruse_amd.data import synth_batch
from cruse_amd.engine import TrainEngine
from cruse_amd.model import cruse_net
from cruse_amd.model.cruse_net import unet_2
A = synth_batch(64, 64000, "cuda", 4)
Bb = synth_batch(64, 64000, "cuda", 11)
res = {}
for graph in (False, True):
    torch.manual_seed(5)
    eng = TrainEngine(unet_2(rnn_groups=1, precision="bf16").cuda(), use_graph=graph, lr=0.0)
    eng.step(*A); eng.step(*A)
    eng.step(*Bb)
    torch.cuda.synchronize()
    res[graph] = ({k: v.clone() for k, v in cruse_net._DBG_KEEP.items() if v is not None}, eng._last_mask.clone(), eng.flat.grads.clone(), eng.flat.params.clone())
a, b = res[False], res[True]
rl = lambda x, y: float((x.float() - y.float()).norm() / y.float().norm())
print("lr = 0, steps A, A, B: graph vs eager rel-L2:", {k: f"{rl(b[0][k], a[0][k]):.1e}" for k in a[0]},
      f"mask {rl(b[1], a[1]):.1e} grads {rl(b[2], a[2]):.1e} params {rl(b[3], a[3]):.1e}")
# per-parameter gradient comparison
eng_names = None
