"""Entry point with the reference's CLI (tools/train_stand.py:93-155): -C -R -V -N -P and the same
TOML sections, wired for one process per MI355X over RCCL.

Launch either way:
    python tools/train_stand.py -C cfg.toml -N 8            # spawns N ranks (mp.spawn, as the reference)
    torchrun --nproc-per-node 8 tools/train_stand.py -C cfg.toml   # RANK/WORLD_SIZE from the env
"""
import argparse
import os
import random
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch.utils.data import DataLoader, DistributedSampler

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))

import train_base.loss as loss  # noqa: E402
from train_base.utils import initialize_module  # noqa: E402


def load_toml(path):
    try:
        import tomllib as t
    except ImportError:
        import tomli as t
    with open(path, "rb") as f:
        return t.load(f)


def entry(rank, world_size, config, resume, only_validation):
    seed = config["meta"]["seed"]
    torch.manual_seed(seed); np.random.seed(seed); random.seed(seed)          # train_stand.py:24-26
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(config["meta"].get("port", 29511)))
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        torch.cuda.set_device(rank % torch.cuda.device_count())
        # each rank's host threads on its GPU's NUMA-local cores ([meta] pin_cores = false opts out): the reference pins nothing
        # (train_stand.py:151-155); eight unpinned launch loops cost an eager step ~20 % (cruse_amd/hostpin.py)
        from cruse_amd import hostpin
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(min(world_size, torch.cuda.device_count()))))
        info = hostpin.pin_rank(rank % max(local_world, 1), local_world, device_index=rank % torch.cuda.device_count(),
                                enable=bool(config["meta"].get("pin_cores", True)))
        if rank == 0 and info.get("pinned"):
            print(f"[train_stand] rank 0 pinned to {info['cores']} cores ({info['first_core']}..{info['last_core']}, "
                  f"{'NUMA-local' if info['numa_local'] else 'even slice'})")
    dist.init_process_group("nccl" if use_gpu else "gloo", rank=rank, world_size=world_size)
    if rank == 0:
        os.makedirs(config["meta"]["save_dir"], exist_ok=True)
    train_dataset = initialize_module(config["train_dataset"]["path"], args=config["train_dataset"]["args"])
    sampler = DistributedSampler(dataset=train_dataset, num_replicas=world_size, rank=rank, shuffle=True)
    train_dataloader = DataLoader(dataset=train_dataset, sampler=sampler, shuffle=False,
                                  **config["train_dataset"]["dataloader"])
    valid_dataloader = DataLoader(dataset=initialize_module(config["validation_dataset"]["path"],
                                                            args=config["validation_dataset"]["args"]),
                                  num_workers=0, batch_size=1)
    model = initialize_module(config["model"]["path"], args=config["model"]["args"])
    optimizer = torch.optim.Adam(params=model.parameters(), lr=config["optimizer"]["lr"],
                                 betas=(config["optimizer"]["beta1"], config["optimizer"]["beta2"]))
    loss_function = getattr(loss, config["loss_function"]["name"])(**config["loss_function"].get("args", {}))
    trainer_class = initialize_module(config["trainer"]["path"], initialize=False)
    trainer = trainer_class(dist=dist, rank=rank, config=config, resume=resume, only_validation=only_validation,
                            model=model, loss_function=loss_function, optimizer=optimizer,
                            train_dataloader=train_dataloader, validation_dataloader=valid_dataloader)
    trainer.train()
    dist.destroy_process_group()


def main(argv=None):
    parser = argparse.ArgumentParser(description="CRUSE on MI355X")
    parser.add_argument("-C", "--configuration", required=True, type=str, help="Configuration (*.toml).")
    parser.add_argument("-R", "--resume", action="store_true", help="Resume from the latest checkpoint.")
    parser.add_argument("-V", "--only_validation", action="store_true", help="Only run validation.")
    parser.add_argument("-N", "--num_gpus", type=int, default=0, help="Number of GPUs (0 = all visible).")
    parser.add_argument("-P", "--preloaded_model_path", type=str, help="Path of the *.pth file of a model.")
    args = parser.parse_args(argv)
    if args.preloaded_model_path:
        assert not args.resume, "The 'resume' conflicts with the 'preloaded_model_path'."
    configuration = load_toml(args.configuration)
    configuration["meta"]["experiment_name"], _ = os.path.splitext(os.path.basename(args.configuration))
    configuration["meta"]["config_path"] = args.configuration
    configuration["meta"]["preloaded_model_path"] = args.preloaded_model_path
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ:                    # torchrun
        entry(int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), configuration, args.resume, args.only_validation)
        return
    if args.num_gpus == 0:
        args.num_gpus = max(torch.cuda.device_count(), 1)
    mp.spawn(entry, args=(args.num_gpus, configuration, args.resume, args.only_validation),
             nprocs=args.num_gpus, join=True)


if __name__ == "__main__":
    main()
