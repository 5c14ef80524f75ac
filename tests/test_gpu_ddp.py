"""GPU: the data-parallel step with 2 ranks (both on cuda:0, gloo for the test rig; production uses
nccl = RCCL).  Same clips on both ranks => the averaged gradient equals the single-rank gradient, so the
parameters after 2 Adam steps must match a single-process run and be identical across ranks."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))

WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
if world > 1:
    dist.init_process_group("gloo", rank=rank, world_size=world)
from cruse_amd.engine import TrainEngine
from cruse_amd.model.cruse_net import unet_2
from cruse_amd import ops
from oracle import cruse_oracle as O
torch.manual_seed(100 + rank)                    # different init per rank: the engine must broadcast rank 0's
m = unet_2(rnn_groups=4, precision="f32").cuda()
if world == 1:
    torch.manual_seed(100); m = unet_2(rnn_groups=4, precision="f32").cuda()
eng = TrainEngine(m, lr=1e-3, use_graph=True)
for step in range(2):
    noisy, clean = O.synth_pair(2, 3200, seed=50 + step)
    eng.step(noisy.cuda(), clean.cuda())
torch.cuda.synchronize()
assert ops.gru_status() == 0
torch.save(eng.flat.params.cpu(), sys.argv[2] + f"/p_w{world}_r{rank}.pt")
if world > 1:
    dist.barrier(); dist.destroy_process_group()
print("OK")
'''


def _run(world, tmp, port):
    script = os.path.join(tmp, "w.py")
    open(script, "w").write(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(world))
    procs = [subprocess.Popen([sys.executable, script, ROOT, tmp], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(world)]
    for p in procs:
        out = p.communicate(timeout=600)[0].decode()
        assert p.returncode == 0 and "OK" in out, out[-2000:]


def test_two_rank_step_equals_single_rank(tmp_path):
    tmp = str(tmp_path)
    _run(1, tmp, 29541)
    _run(2, tmp, 29542)
    p1 = torch.load(tmp + "/p_w1_r0.pt")
    a = torch.load(tmp + "/p_w2_r0.pt")
    b = torch.load(tmp + "/p_w2_r1.pt")
    assert torch.equal(a, b), "ranks diverged"
    # sum of two identical gradients * 0.5 == the single-rank gradient (bitwise up to atomic-add order)
    assert (a - p1).abs().max() <= 5e-4, float((a - p1).abs().max())
