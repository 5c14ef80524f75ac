"""CPU: C-ABI surface, flat-parameter layout, plug-in loader, world_size-2 gloo all-reduce."""
import ctypes
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_library_exports_every_header_symbol():
    from cruse_amd._abi_check import parse_header
    from cruse_amd._lib import LIB_PATH, SIGNATURES
    hdr = parse_header()
    assert len(hdr) >= 32
    lib = ctypes.CDLL(LIB_PATH)
    for name, sig in hdr.items():
        assert hasattr(lib, name), f"{name} declared in include/cruse_hip.h but not exported"
        assert SIGNATURES[name] == sig, f"{name}: ctypes signature {SIGNATURES[name]} != header {sig}"
    assert set(SIGNATURES) == set(hdr)
    lib.cruse_abi_version.restype = ctypes.c_int
    from cruse_amd._lib import ABI_VERSION
    assert lib.cruse_abi_version() == ABI_VERSION == 13


def test_error_channel_without_gpu():
    from cruse_amd._lib import lib
    # shape validation happens on the host before any HIP call
    rc = lib.cruse_gemm(0, 0, 0, 4, 4, None, 4, None, 4, None, 4, None, 0, 1, 0, 0, None)
    assert rc == -1 and b"gemm" in lib.cruse_last_error()
    rc = lib.cruse_conv_gather(None, None, None, None, 1, 1, 1, 8, 4, 8, 2, 2, 1, 0, 0, 0, -1, 0, 0, None)
    assert rc == -1 and b"Fout" in lib.cruse_last_error()
    rc = lib.cruse_gru_seq_fwd(None, None, None, None, None, None, None, 2, 3, 1, 100, 0, None, None)
    assert rc == -1 and b"multiple of 32" in lib.cruse_last_error()


def test_ops_refuse_cpu_tensors():
    from cruse_amd import ops
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        ops.stft(torch.zeros(1, 3200), 320, 160)


def test_model_surface_and_state_dict_compat():
    from cruse_amd.model.cruse_net import GGRU, unet_2
    from oracle import cruse_oracle as O
    m = unet_2(rnn_groups=2)
    o = O.unet_2(rnn_groups=2)
    assert list(m.state_dict().keys()) == list(o.state_dict().keys())
    m.load_state_dict(o.state_dict(), strict=True)
    g = GGRU()
    assert g.ln1.normalized_shape == (1024,) and len(g.gru_list1) == 2
    with pytest.raises(RuntimeError):
        m(torch.zeros(1, 1, 4, 161))          # 161 bins do not satisfy the hidden-size formula (R8)


def test_same_seed_same_init_as_reference_layout():
    from cruse_amd.model.cruse_net import unet_2
    from oracle import cruse_oracle as O
    torch.manual_seed(7); a = unet_2(rnn_groups=4)
    torch.manual_seed(7); b = O.unet_2(rnn_groups=4)
    for (na, pa), (nb, pb) in zip(a.named_parameters(), b.named_parameters()):
        assert na == nb and torch.equal(pa, pb), na


def test_flat_params_layout():
    from cruse_amd.engine import ALIGN, FlatParams, unused_parameter
    from oracle import cruse_oracle as O
    m = O.unet_2(rnn_groups=4)
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    fp = FlatParams(m)
    assert all(o % ALIGN == 0 for o in fp.offsets.values())
    assert not any(unused_parameter(n) and n in fp.G for n in before)
    assert sum(p.numel() for n, p in before.items() if not unused_parameter(n)) == 1280155 - 2  # bn1_t excluded
    for n, p in m.named_parameters():
        assert torch.equal(p, before[n])
        if n in fp.offsets:
            assert p.data_ptr() == fp.params.data_ptr() + 4 * fp.offsets[n]
    fp.params.mul_(2.0)
    assert torch.equal(m.conv1.weight, before["conv1.weight"] * 2)


def test_initialize_module_and_shims():
    from train_base.utils import initialize_module
    cls = initialize_module("model.cruse_net.unet_2", initialize=False)
    from cruse_amd.model.cruse_net import unet_2
    assert cls is unet_2
    m = initialize_module("model.cruse_net.unet_2", args={"rnn_groups": 1})
    assert m.rnn_groups == 1 and m.hidden_size == 640
    tr = initialize_module("train.trainer_casual.Trainer", initialize=False)
    import inspect
    assert list(inspect.signature(tr.__init__).parameters)[1:] == [
        "dist", "rank", "config", "resume", "only_validation", "model", "loss_function", "optimizer",
        "train_dataloader", "validation_dataloader"]


def test_toml_config_parses(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from tools.train_stand import load_toml
    cfg = load_toml(os.path.join(ROOT, "configs", "cruse_synthetic.toml"))
    for sec in ("meta", "acoustics", "train_dataset", "validation_dataset", "model", "optimizer", "loss_function", "trainer"):
        assert sec in cfg
    assert cfg["acoustics"]["n_fft"] == 320 and cfg["acoustics"]["hop_length"] == 160


WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from cruse_amd.engine import FlatParams
from oracle import cruse_oracle as O
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
torch.manual_seed(100 + rank)                       # different init per rank on purpose
m = O.unet_2(rnn_groups=4)
fp = FlatParams(m)
fp.broadcast(0)
ref = [torch.zeros_like(fp.params) for _ in range(world)]
dist.all_gather(ref, fp.params)
assert all(torch.equal(ref[0], r) for r in ref), "weights differ after broadcast"
# rank-local gradients g_r = (rank+1) * pattern ; the all-reduce must give sum_r g_r
pat = torch.sin(torch.arange(fp.total, dtype=torch.float32))
fp.grads.copy_((rank + 1) * pat)
fp.all_reduce_grads()
want = sum(r + 1 for r in range(world)) * pat
assert torch.allclose(fp.grads, want, atol=1e-6)
assert torch.allclose(fp.G["conv1.weight"].flatten(), want[fp.offsets["conv1.weight"]:fp.offsets["conv1.weight"] + 48])
dist.barrier(); dist.destroy_process_group()
print("OK", rank)
'''


def test_gloo_world2_gradient_allreduce(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    import socket
    with socket.socket() as sk:                      # a free port of this host (a fixed one collides with a parallel run)
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE="2", OMP_NUM_THREADS="1")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=300)[0].decode() for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0 and "OK" in o, o


def test_inferencer_import_path_and_int16_scaling():
    """train_base/inferencer/base_inferencer.py resolves to the HIP inferencer; int16 scaling of :183-185."""
    import numpy as np
    from train_base.inferencer.base_inferencer import Inferencer
    w = Inferencer.to_int16(np.array([0.0, 0.25, -0.5], dtype=np.float32))
    assert w.dtype == np.int16 and w.tolist() == [0, int(0.8 * 32767 * 0.5), int(-0.8 * 32767)]


def test_bucket_layout_follows_backward_order():
    from cruse_amd.engine import FlatParams
    from cruse_amd.model.cruse_net import bucket_of, unet_2
    m = unet_2(rnn_groups=1)
    fp = FlatParams(m)
    (s0, e0), (s1, e1), (s2, e2) = fp.bucket_range
    assert s0 == 0 and e0 == s1 and e1 == s2 and e2 == fp.total
    for n, o in fp.offsets.items():
        b = bucket_of(n)
        assert fp.bucket_range[b][0] <= o < fp.bucket_range[b][1], n
    assert bucket_of("gru.gru_list2.0.weight_hh_l0") == 0 and bucket_of("conv3_t.weight") == 0
    assert bucket_of("gru.gru_list1.0.weight_ih_l0") == 1 and bucket_of("conv2.weight") == 2
    # the two GGRU layers dominate: buckets 0 and 1 each carry about half of the 19.9 MB
    assert abs((e0 - s0) - (e1 - s1)) < 0.05 * fp.total and (e2 - s2) < 0.02 * fp.total


def test_adam_state_dict_is_torch_adam_layout():
    """FlatParams.adam_state_dict / load_adam_state_dict <-> torch.optim.Adam.state_dict() (base_trainer.py:167,199-203)."""
    from cruse_amd.engine import FlatParams
    from oracle import cruse_oracle as O
    m = O.unet_2(rnn_groups=2)
    fp = FlatParams(m)
    fp.exp_avg.copy_(torch.sin(torch.arange(fp.total, dtype=torch.float32)))
    fp.exp_avg_sq.copy_(torch.cos(torch.arange(fp.total, dtype=torch.float32)) ** 2)
    sd = fp.adam_state_dict(7, 1e-3, (0.9, 0.999), 1e-8, 0.0)
    opt = torch.optim.Adam(m.parameters(), lr=5e-4)
    opt.load_state_dict(sd)                                   # the reference's resume path accepts it
    st = opt.state_dict()
    names = [n for n, _ in m.named_parameters()]
    assert set(st["state"]) == {i for i, n in enumerate(names) if n in fp.offsets}
    i = names.index("gru.gru_list1.0.weight_hh_l0")
    assert torch.equal(st["state"][i]["exp_avg"], fp.view(fp.exp_avg, names[i])) and float(st["state"][i]["step"]) == 7.0
    fp2 = FlatParams(O.unet_2(rnn_groups=2))
    assert fp2.load_adam_state_dict(st) == 7
    for n in fp.names:                                        # (the alignment gaps between tensors are not state)
        assert torch.equal(fp2.view(fp2.exp_avg, n), fp.view(fp.exp_avg, n)), n
        assert torch.equal(fp2.view(fp2.exp_avg_sq, n), fp.view(fp.exp_avg_sq, n)), n
    bad = dict(st, param_groups=[dict(st["param_groups"][0], params=st["param_groups"][0]["params"][:-1])])
    with pytest.raises(RuntimeError, match="parameters"):
        fp2.load_adam_state_dict(bad)


def test_loss_factories_carry_engine_tags():
    import train_base.loss as L
    assert L.wo_male_loss(alpha=3.0).cruse_loss == ("wo_male", {"loss_alpha": 3.0, "loss_beta": 1.0})
    assert L.si_snr_loss().cruse_loss == ("si_snr", {})
    assert L.sdnr_loss(snr=5.0).cruse_loss == ("sdnr", {"snr_db": 5.0, "sdnr_beta_db": 20.0})
    assert not hasattr(L.l1_loss(), "cruse_loss")


def test_reference_dotted_paths_resolve():
    """every reference import path this repo stands in for (train_base/utils.py:68-100 loads by dotted path)."""
    from train_base.utils import initialize_module
    for path in ("model.based_model.cust_conv.GroupGRU", "model.based_model.cust_conv.convkxf", "model.mtfaa.TFCM_Block",
                 "train_base.acoustics.conv_stft.STFT", "train_base.acoustics.feature.CustomSTFT",
                 "train_base.acoustics.mask.compress_cIRM", "loss_func.loss.loss_func", "dataset.dataset.SynDataset",
                 "model.deep_filter.DeepFilter"):
        assert initialize_module(path, initialize=False) is not None, path


def test_hostpin_plan_gives_every_rank_its_own_cores():
    """cruse_amd/hostpin.py: ranks whose GPUs share a NUMA node split that node's cores, ranks without locality information split the allowed
    cores evenly; slices are disjoint, inside the allowed set and never empty"""
    from cruse_amd import hostpin
    assert hostpin._parse_cpulist("0-3,8,10-11\n") == [0, 1, 2, 3, 8, 10, 11]
    allowed = list(range(256))
    node0, node1 = list(range(0, 64)) + list(range(128, 192)), list(range(64, 128)) + list(range(192, 256))
    by_rank = {r: (node0 if r < 4 else node1) for r in range(8)}
    slices = [hostpin.plan(r, 8, allowed, by_rank) for r in range(8)]
    assert all(len(s) == 32 for s in slices)
    assert all(set(s) <= set(node0 if r < 4 else node1) for r, s in enumerate(slices))
    assert len(set().union(*map(set, slices))) == 256                      # disjoint and complete
    none = {r: None for r in range(8)}
    slices = [hostpin.plan(r, 8, allowed[:8], none) for r in range(8)]
    assert sorted(c for s in slices for c in s) == list(range(8))          # one core each on an 8-core box
    assert hostpin.plan(5, 8, [3], none) == [3]                            # fewer cores than ranks: shared, not empty
    # sysfs says nothing in this container / for a missing device: no exception
    assert hostpin.gpu_local_cores(0) is None or isinstance(hostpin.gpu_local_cores(0), list)
    info = hostpin.pin_rank(0, 1)
    assert info["pinned"] is False
