"""Does the GGRU wavefront capture into a HIP graph?  Runs the engine step eagerly and as a graph with the forward half, then both."""
import sys, torch, faulthandler
sys.path.insert(0, '.')
import ctypes, os
if os.path.exists("tools/scratch/libsegv_bt.so"): ctypes.CDLL("tools/scratch/libsegv_bt.so").segv_bt_install()
from cruse_amd.config import EngineConfig
from cruse_amd.data import synth_batch
from cruse_amd.engine import TrainEngine
from cruse_amd.model.cruse_net import unet_2
from cruse_amd import ops
which = sys.argv[1] if len(sys.argv) > 1 else "fwd"
noisy, clean = synth_batch(16, 32000, "cuda", 4)
cfg = EngineConfig(ggru_wave=0, fuse_dgi=True) if which == "fusedgi" else EngineConfig(ggru_wave=4, ggru_wave_bwd=(which in ("both", "bwd")), ggru_wave_fwd=(which in ("both", "fwd")))
for graph in (False, True):
    torch.manual_seed(5)
    eng = TrainEngine(unet_2(rnn_groups=1, precision="bf16").cuda(), use_graph=graph, config=cfg)
    if os.path.exists("tools/scratch/libsegv_bt.so"): ctypes.CDLL("tools/scratch/libsegv_bt.so").segv_bt_install()
    ls = eng.step(noisy, clean)
    torch.cuda.synchronize()
    print(which, "graph" if graph else "eager", "loss", eng.loss_value(ls), "status", ops.gru_status(), flush=True)
