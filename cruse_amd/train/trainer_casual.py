"""The DDP trainer `train/trainer_casual.py` was meant to hold (the reference file is empty).

Accepts exactly the keyword arguments tools/train_stand.py:79-88 passes and exposes
train() like train_base/trainer/base_trainer.py:378-424.  One process per GPU; gradients are
all-reduced over RCCL (torch.distributed backend "nccl") by cruse_amd.engine.TrainEngine --
no DistributedDataParallel wrapper, BatchNorm statistics stay rank-local as with the
reference's plain DDP (base_trainer.py:31, no SyncBN).
"""
from __future__ import annotations

import os
import time

import torch

from ..engine import TrainEngine


class Trainer:
    def __init__(self, dist, rank, config, resume, only_validation, model, loss_function, optimizer,
                 train_dataloader, validation_dataloader):
        self.dist, self.rank, self.config = dist, rank, config
        self.only_validation = only_validation
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.model = model.to(self.device)
        self.optimizer = optimizer
        self.loss_function = loss_function            # kept for interface parity; the engine fuses WO-MALE
        self.train_dataloader = train_dataloader
        self.validation_dataloader = validation_dataloader
        ac = config["acoustics"]
        tr = config["trainer"]["train"]
        self.epochs = tr["epochs"]
        self.save_checkpoint_interval = tr.get("save_checkpoint_interval", 1)
        self.clip_grad_norm_value = tr.get("clip_grad_norm_value", None)
        assert self.save_checkpoint_interval >= 1                      # base_trainer.py:76
        self.save_dir = os.path.join(config["meta"]["save_dir"], config["meta"].get("experiment_name", "exp"))
        self.checkpoints_dir = os.path.join(self.save_dir, "checkpoints")
        g = optimizer.param_groups[0]
        self.engine = TrainEngine(self.model, lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"],
                                  weight_decay=g.get("weight_decay", 0.0), n_fft=ac["n_fft"], hop=ac["hop_length"],
                                  precision=config["meta"].get("precision", None))
        self.start_epoch = 1
        self.best_score = float("inf")
        if rank == 0:
            os.makedirs(self.checkpoints_dir, exist_ok=True)
        if resume:
            self._resume_checkpoint()

    # -- checkpoint schema of base_trainer.py:186-232 (latest_model.tar) --------------------
    def _save_checkpoint(self, epoch):
        state = {"epoch": epoch, "best_score": self.best_score,
                 "optimizer": {"step": self.engine.step_count, "exp_avg": self.engine.flat.exp_avg.cpu(),
                               "exp_avg_sq": self.engine.flat.exp_avg_sq.cpu()},
                 "scaler": None,
                 "model": {k: v.detach().cpu() for k, v in self.model.state_dict().items()}}
        torch.save(state, os.path.join(self.checkpoints_dir, "latest_model.tar"))
        torch.save(state["model"], os.path.join(self.checkpoints_dir, f"model_{str(epoch).zfill(4)}.pth"))

    def _resume_checkpoint(self):
        path = os.path.join(self.checkpoints_dir, "latest_model.tar")
        assert os.path.exists(path), f"{path} does not exist, can not load latest checkpoint."
        if self.dist is not None and self.dist.is_initialized():
            self.dist.barrier()                                          # base_trainer.py:161
        ck = torch.load(path, map_location="cpu")
        self.start_epoch = ck["epoch"] + 1
        self.best_score = ck["best_score"]
        self.model.load_state_dict(ck["model"])
        self.engine.step_count = ck["optimizer"]["step"]
        self.engine.flat.exp_avg.copy_(ck["optimizer"]["exp_avg"])
        self.engine.flat.exp_avg_sq.copy_(ck["optimizer"]["exp_avg_sq"])

    def _train_epoch(self, epoch):
        total, nb, frames = 0.0, 0, 0
        t0 = time.time()
        for noisy, clean in self.train_dataloader:
            noisy = noisy.to(self.device, non_blocking=True).float().contiguous()
            clean = clean.to(self.device, non_blocking=True).float().contiguous()
            ls = self.engine.step(noisy, clean)
            total += self.engine.loss_value(ls)
            nb += 1
            frames += noisy.shape[0] * (1 + noisy.shape[1] // self.engine.hop)
        dt = time.time() - t0
        if self.rank == 0:
            print(f"[epoch {epoch}] loss {total / max(nb, 1):.6f}  {frames / max(dt, 1e-9):.0f} frames/s/rank")
        return total / max(nb, 1)

    @torch.no_grad()
    def _validation_epoch(self, epoch):
        from ..acoustics.feature import pre_stft
        from ..loss import masked_wo_male
        from .. import ops
        self.model.eval()
        total, nb = 0.0, 0
        for noisy, clean in self.validation_dataloader:
            noisy = noisy.to(self.device).float().contiguous()
            clean = clean.to(self.device).float().contiguous()
            f = pre_stft(noisy, self.engine.n_fft, self.engine.hop, self.engine.n_fft, f_net=self.engine.f_net)
            _, _, cmag = ops.stft(clean, self.engine.n_fft, self.engine.hop, want_ri=False,
                                  mag_bins=self.engine.f_stft)
            mask = self.model(f["mag_net"])
            total += float(masked_wo_male(mask, f["real"], f["imag"], cmag))
            nb += 1
        self.model.train()
        return total / max(nb, 1)

    def train(self):
        for epoch in range(self.start_epoch, self.epochs + 1):
            if self.only_validation and self.rank == 0:
                print(f"validation loss {self._validation_epoch(epoch):.6f}")
                return
            self.model.train()
            self._train_epoch(epoch)
            if self.rank == 0 and epoch % self.save_checkpoint_interval == 0:
                self._save_checkpoint(epoch)
            vi = self.config["trainer"].get("validation", {}).get("validation_interval", 0)
            if self.rank == 0 and vi and epoch % vi == 0 and self.validation_dataloader is not None:
                score = self._validation_epoch(epoch)
                self.best_score = min(self.best_score, score)
                print(f"[epoch {epoch}] validation loss {score:.6f}")
