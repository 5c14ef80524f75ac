"""The DDP trainer `train/trainer_casual.py` was meant to hold (the reference file is empty).

Accepts exactly the keyword arguments tools/train_stand.py:79-88 passes and exposes train() like
train_base/trainer/base_trainer.py:378-424.  One process per GPU; gradients are all-reduced over RCCL
(torch.distributed backend "nccl") by cruse_amd.engine.TrainEngine in buckets overlapped with the backward pass --
no DistributedDataParallel wrapper; BatchNorm statistics stay rank-local as with the reference's plain DDP
(base_trainer.py:31, no SyncBN) and rank 0's are the ones saved (base_trainer.py:206-207).

What the reference's knobs do here:
  loss_function               -> the fused engine loss named by its `.cruse_loss` tag (train_base/loss.py); l1_loss /
                                 mse_loss (torch.nn.L1Loss / MSELoss instances) -> the waveform losses "l1" / "mse"
  meta.use_amp                -> true: precision "bf16", false: "f32" (printed); meta.precision overrides (base_trainer.py:41-42)
  optimizer (torch Adam)      -> lr / betas / eps / weight_decay are read from it; its state_dict() layout is what
                                 latest_model.tar stores and what resume loads (base_trainer.py:167,199-203)
  clip_grad_norm_value        -> torch.nn.utils.clip_grad_norm_ semantics, folded into the fused Adam (base_trainer.py:75)
  meta.preloaded_model_path   -> model weights from a checkpoint, strict=False (base_trainer.py:130-146)
  meta.hip_graph (this repo)  -> true: replay the step from HIP graphs, false: launch eagerly, "auto" (default): time both
                                 on the first eight real steps and keep the faster
  save_checkpoint_interval, validation_interval, save_max_metric_score -> as base_trainer.py:72-86,395-418
The validation score is the configured loss on the validation set (the reference's STOI / PESQ metrics are host-side
libraries outside this path), so `save_max_metric_score = false` is the meaningful setting.
"""
from __future__ import annotations

import os
import time

import torch

from .. import ops
from ..engine import TrainEngine


class _RingCollate:
    """collate_fn wrapper that runs inside the DataLoader's WORKERS (round 6).  torch hands a collated batch to the main process as freshly
    created shared-memory tensors: per batch two file descriptors fetched from the worker's resource-sharer thread (an authenticated socket
    round trip each: `recvmsg` was 0.96 ms per storage, 3 ms per batch of the consumer's 4.2 ms -- cProfile on the bench host, whatever the
    worker count), an mmap / munmap of 32.8 MB and page faults on every first touch.  Here a worker stacks its samples straight into ITS slots
    of a shared ring the trainer allocated before the workers were forked and returns (tag, slot, count) -- a few bytes through the queue, pages
    that stay resident.  Anything that is not a batch of (tensor, tensor) pairs of the ring's shape goes through the original collate_fn."""

    TAG = "__cruse_ring__"

    def __init__(self, inner, ring, per_worker):
        self.inner, self.ring, self.k, self.count = inner, ring, int(per_worker), 0

    def __call__(self, batch):
        from torch.utils.data import get_worker_info
        info, ring = get_worker_info(), self.ring
        try:
            shp, dt = ring.shape[3:], ring.dtype
            ok = (info is not None and 0 < len(batch) <= ring.shape[2] and (info.id + 1) * self.k <= ring.shape[0]
                  and all(isinstance(s_, (tuple, list)) and len(s_) == 2 and isinstance(s_[0], torch.Tensor) and isinstance(s_[1], torch.Tensor)
                          and s_[0].shape == shp and s_[1].shape == shp and s_[0].dtype == dt and s_[1].dtype == dt for s_ in batch))
        except Exception:
            ok = False
        if not ok:
            return self.inner(batch)
        slot = info.id * self.k + self.count % self.k
        self.count += 1
        b = len(batch)
        torch.stack([s_[0] for s_ in batch], 0, out=ring[slot, 0, :b])
        torch.stack([s_[1] for s_ in batch], 0, out=ring[slot, 1, :b])
        return (self.TAG, slot, b)


def _install_ring(loader, max_bytes=2 << 30):
    """-> the shared ring of `loader` (installing it on first use), or None where it does not apply: no workers, workers already running
    (a persistent iterator made before), manual batching, samples that are not (tensor, tensor) pairs of one shape, or a ring beyond max_bytes.
    Slots per worker: prefetch_factor batches the DataLoader keeps outstanding per worker + what the prefetcher holds before its staging copy
    (two queued, one being copied) + one spare."""
    ring = getattr(loader, "_cruse_ring", None)
    if ring is not None:
        return ring
    if getattr(loader, "_cruse_ring_tried", False):
        return None
    try:
        loader._cruse_ring_tried = True
    except Exception:
        return None
    nw, bs = int(getattr(loader, "num_workers", 0) or 0), getattr(loader, "batch_size", None)
    if nw <= 0 or bs is None or getattr(loader, "_iterator", None) is not None or getattr(loader, "collate_fn", None) is None:
        return None
    # one sample decides the ring's geometry; the host RNGs are put back afterwards (a dataset that augments with torch / numpy / random draws
    # would otherwise shift the DataLoader's base seed and with it every shuffle of the run)
    import random as _random
    import numpy as _np
    rng = (torch.random.get_rng_state(), _np.random.get_state(), _random.getstate())
    try:
        s0 = loader.dataset[0]
    except Exception:
        return None
    finally:
        torch.random.set_rng_state(rng[0]); _np.random.set_state(rng[1]); _random.setstate(rng[2])
    if not (isinstance(s0, (tuple, list)) and len(s0) == 2 and all(isinstance(t_, torch.Tensor) for t_ in s0)
            and s0[0].shape == s0[1].shape and s0[0].dtype == s0[1].dtype and s0[0].dim() >= 1):
        return None
    k = int(getattr(loader, "prefetch_factor", None) or 2) + 5          # outstanding per worker + pull thread's hand (1) + queue (2) + staging (1) + spare (1)
    shape = (nw * k, 2, int(bs)) + tuple(s0[0].shape)
    nbytes = s0[0].element_size()
    for d in shape:
        nbytes *= d
    if nbytes > max_bytes:
        return None
    try:                                                     # the ring lives in /dev/shm: a tmpfs that is too small fails at first TOUCH (SIGBUS)
        import os as _os
        st = _os.statvfs("/dev/shm")
        if st.f_bavail * st.f_frsize < 2 * nbytes:
            return None
    except OSError:
        pass
    try:
        ring = torch.empty(shape, dtype=s0[0].dtype).share_memory_()
        loader.collate_fn = _RingCollate(loader.collate_fn, ring, k)
        loader._cruse_ring = ring
    except Exception:
        return None
    return ring


class _Prefetcher:
    """Device batches one step AHEAD of the training loop, off the compute stream (the loop used to copy an un-pinned
    [B, L] pair over PCIe on the compute stream, in series with every step).
      * a HOST dataset (the reference's DataLoader, its kwargs untouched): a background thread pulls batches from the DataLoader and
        stages them in a ring of four PINNED host buffers; the copy stream moves a staged pair to the device with non-blocking copies
        (32.8 MB at B = 64: ~0.5 ms of PCIe beside a ~5 ms step) and records the event the compute stream waits for;
      * a DEVICE-RESIDENT dataset (`dataset.device_resident`, e.g. cruse_amd.data.DevicePairs): the DataLoader is never iterated --
        the indices of its own sampler go to dataset.device_batch() on the data stream (gather + on-GPU snr_mix).
    Batch k + 1 is issued before batch k is handed to the loop.  The device tensors are allocated on the data stream and marked as used
    by the compute stream (record_stream), so their memory is not reused before the step that reads them has run; a pinned buffer is
    re-staged only after its copy's event has completed.  Iterating yields (noisy, clean) f32 device tensors ready on the CURRENT stream."""

    NPIN = 4

    def __init__(self, loader, device, use_ring=True):
        from .. import streams
        self.loader, self.device, self.use_ring = loader, device, bool(use_ring)
        # ONE data stream per compute stream for the life of the process (a stream per epoch strands the blocks the caching allocator
        # holds for the old one), proven to overlap with the compute stream and its leaf stream (cruse_amd/streams.py: a copy stream
        # that shares their hardware queue would put every H2D copy in series with the step)
        with torch.cuda.device(device):
            main = torch.cuda.current_stream(device)
            self.stream = streams.stream_beside(main, avoid=(streams.side_stream_for(main),), tag="data")
        self.resident = bool(getattr(getattr(loader, "dataset", None), "device_resident", False))

    def _index_batches(self):
        """the index lists of the DataLoader's OWN batch sampler (batch_size / drop_last / sampler, or a custom batch_sampler=)"""
        bsamp = getattr(self.loader, "batch_sampler", None)
        if bsamp is None:
            raise RuntimeError("a device-resident dataset needs a DataLoader with automatic batching (batch_size or batch_sampler)")
        for idx in bsamp:
            yield torch.as_tensor(list(idx), dtype=torch.int64)

    def _resident_items(self, main):
        for idx in self._index_batches():
            with torch.cuda.stream(self.stream):
                noisy, clean = self.loader.dataset.device_batch(idx, self.device)
                ev = torch.cuda.Event()
                ev.record(self.stream)
            yield noisy, clean, ev

    def _host_items(self, main):
        import queue
        import threading
        import numpy as np
        q: "queue.Queue" = queue.Queue(maxsize=2)
        copy_done = [None] * self.NPIN             # event behind the H2D copy that last read pinned slot s
        pinned = {}

        def stage(t, slot, name):
            if t.is_pinned():                      # (pin_memory=True in the DataLoader kwargs: a fresh pinned tensor per batch)
                return t
            key = (slot, name, tuple(t.shape), t.dtype)
            if key not in pinned:
                pinned[key] = torch.empty(t.shape, dtype=t.dtype).pin_memory()
            # ONE thread copies (numpy memcpy, GIL released): the batch sits in freshly mapped shared-memory pages, and torch's
            # multi-threaded copy_ faults them in from 128 threads that serialise on the process's mm lock -- 65-78 ms per
            # [64, 64000] pair on the bench host against 2.9 ms for the single memcpy (tools/loader_probe.py)
            if t.is_contiguous() and t.dtype != torch.bfloat16:
                np.copyto(pinned[key].numpy(), t.numpy())
            else:
                pinned[key].copy_(t)
            return pinned[key]

        raw: "queue.Queue" = queue.Queue(maxsize=2)
        stop = threading.Event()                   # set when the consumer leaves early (an exception or a break in the training loop)

        def put(qu, item):
            """qu.put that gives up when the consumer has left (-> False)"""
            while not stop.is_set():
                try:
                    qu.put(item, timeout=0.05)
                    return True
                except queue.Full:
                    pass
            return False

        ring = _install_ring(self.loader) if self.use_ring else None

        def pull():                                # stage 1: the DataLoader's consumer side (~4 ms per 32.8 MB batch on the bench host; with the
            try:                                   # shared ring a few bytes per batch)
                it = iter(self.loader)
                try:
                    for item in it:
                        if ring is not None and isinstance(item, tuple) and len(item) == 3 and item[0] == _RingCollate.TAG:
                            item = (ring[item[1], 0, :item[2]], ring[item[1], 1, :item[2]])      # views: copied out by the staging thread
                        if not put(raw, item):
                            return
                finally:
                    del it                         # (lets a non-persistent DataLoader shut its workers down)
                put(raw, None)
            except BaseException as ex:
                put(raw, ex)

        def work():                                # stage 2: shared-memory pages -> pinned ring (~3 ms), beside stage 1
            try:
                k = 0
                while not stop.is_set():
                    try:
                        item = raw.get(timeout=0.05)
                    except queue.Empty:
                        continue
                    if item is None or isinstance(item, BaseException):
                        put(q, item)
                        return
                    noisy, clean = item
                    s_ = k % self.NPIN
                    # (two batches queued + this one: the copy of the batch that used slot s_ four batches ago was issued long ago)
                    if copy_done[s_] is not None:
                        copy_done[s_].synchronize()
                    if not put(q, (s_, stage(noisy, s_, "n"), stage(clean, s_, "c"))):
                        return
                    k += 1
            except BaseException as ex:            # surfaces in the training loop
                put(q, ex)
        th0 = threading.Thread(target=pull, daemon=True)
        th = threading.Thread(target=work, daemon=True)
        th0.start()
        th.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                s_, hn, hc = item
                with torch.cuda.stream(self.stream):
                    noisy = hn.to(self.device, non_blocking=True)
                    clean = hc.to(self.device, non_blocking=True)
                    if noisy.dtype != torch.float32:
                        noisy, clean = noisy.float(), clean.float()
                    ev = torch.cuda.Event()
                    ev.record(self.stream)
                copy_done[s_] = ev
                yield noisy.contiguous(), clean.contiguous(), ev
        finally:
            # also reached when the generator is CLOSED at its yield (the training loop raised or broke): both threads see `stop`
            # within 50 ms, drop what they hold and end -- the DataLoader iterator, its workers and the pinned ring go with them
            stop.set()
            for qu in (q, raw):
                try:
                    while True:
                        qu.get_nowait()
                except queue.Empty:
                    pass
            th.join(timeout=5.0)
            th0.join(timeout=5.0)

    def __iter__(self):
        main = torch.cuda.current_stream(self.device)
        prev = None
        for item in (self._resident_items(main) if self.resident else self._host_items(main)):
            if prev is not None:
                yield self._hand(prev, main)
            prev = item
        if prev is not None:
            yield self._hand(prev, main)

    @staticmethod
    def _hand(item, main):
        noisy, clean, ev = item
        main.wait_event(ev)
        noisy.record_stream(main); clean.record_stream(main)
        return noisy, clean


def _graph_mode(v):
    """meta.hip_graph: true / false / "auto"."""
    if isinstance(v, str):
        if v.lower() == "auto":
            return "auto"
        return v.lower() in ("1", "true", "yes", "on")
    return bool(v)


class Trainer:
    def __init__(self, dist, rank, config, resume, only_validation, model, loss_function, optimizer,
                 train_dataloader, validation_dataloader):
        self.dist, self.rank, self.config = dist, rank, config
        self.only_validation = only_validation
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.model = model.to(self.device)
        self.optimizer = optimizer
        self.loss_function = loss_function
        self.train_dataloader = train_dataloader
        self.validation_dataloader = validation_dataloader
        ac = config["acoustics"]
        tr = config["trainer"]["train"]
        va = config["trainer"].get("validation", {})
        self.epochs = tr["epochs"]
        self.save_checkpoint_interval = tr.get("save_checkpoint_interval", 1)
        self.max_gru_timeouts_per_epoch = int(config["meta"].get("max_gru_timeouts_per_epoch", 0))
        self.shm_ring = bool(config["meta"].get("shm_ring", True))          # host datasets: workers collate into a shared ring (_RingCollate)
        self.clip_grad_norm_value = tr.get("clip_grad_norm_value", None)
        assert self.save_checkpoint_interval >= 1, \
            "Check the 'save_checkpoint_interval' parameter in the config. It should be large than one."   # base_trainer.py:76
        self.validation_interval = va.get("validation_interval", 1)
        self.save_max_metric_score = va.get("save_max_metric_score", False)
        assert self.validation_interval >= 1, \
            "Check the 'validation_interval' parameter in the config. It should be large than one."       # base_trainer.py:84
        self.save_dir = os.path.join(config["meta"]["save_dir"], config["meta"].get("experiment_name", "exp"))
        self.checkpoints_dir = os.path.join(self.save_dir, "checkpoints")
        tag = getattr(loss_function, "cruse_loss", None)
        if tag is None and isinstance(loss_function, (torch.nn.L1Loss, torch.nn.MSELoss)):
            # l1_loss / mse_loss of train_base/loss.py:3-4 (torch.nn.L1Loss / MSELoss): waveform loss through the iSTFT
            if loss_function.reduction != "mean":
                raise RuntimeError(f"{type(loss_function).__name__}(reduction={loss_function.reduction!r}): the fused form is "
                                   "reduction='mean' (the torch default)")
            tag = ("l1" if isinstance(loss_function, torch.nn.L1Loss) else "mse", {})
        if tag is None:
            raise RuntimeError(f"loss_function {loss_function!r} has no fused HIP form: use l1_loss, mse_loss, wo_male_loss, "
                               "si_snr_loss or sdnr_loss from train_base.loss (config [loss_function].name)")
        loss_name, loss_kwargs = tag
        # meta.use_amp (base_trainer.py:41-42: GradScaler(enabled=use_amp)) has no autocast / scaler here; what mixed precision
        # means on this path is the bf16-operand MFMA mode (f32 storage, statistics and accumulation, no loss scaling needed).
        # meta.precision (this repo) wins; otherwise use_amp = true -> "bf16", false -> "f32" (the reference's arithmetic).
        precision = config["meta"].get("precision", None)
        if precision is None and "use_amp" in config["meta"]:
            precision = "bf16" if config["meta"]["use_amp"] else "f32"
            if rank == 0:
                print(f"[cruse_amd] meta.use_amp = {bool(config['meta']['use_amp'])} -> precision '{precision}' "
                      "(bf16 MFMA operands with f32 accumulation / exact f32; set meta.precision to choose explicitly)", flush=True)
        g = optimizer.param_groups[0]
        self.engine = TrainEngine(self.model, lr=g["lr"], betas=tuple(g["betas"]), eps=g["eps"],
                                  weight_decay=g.get("weight_decay", 0.0), n_fft=ac["n_fft"], hop=ac["hop_length"],
                                  precision=precision,
                                  use_graph=_graph_mode(config["meta"].get("hip_graph", "auto")), loss=loss_name,
                                  clip_grad_norm=float(self.clip_grad_norm_value or 0.0), **loss_kwargs)
        self.start_epoch = 1
        self.best_score = -float("inf") if self.save_max_metric_score else float("inf")      # base_trainer.py:93
        if rank == 0:
            os.makedirs(self.checkpoints_dir, exist_ok=True)
        if resume:
            self._resume_checkpoint()
        pre = config["meta"].get("preloaded_model_path")
        if pre:
            self._preload_model(pre)

    # -- checkpoint schema of base_trainer.py:130-232 -----------------------------------------------------------------
    def _preload_model(self, model_path):
        model_path = os.path.abspath(os.path.expanduser(model_path))
        assert os.path.exists(model_path), f"The file {model_path} is not exist. please check path."
        ck = torch.load(model_path, map_location="cpu")
        self.model.load_state_dict(ck["model"] if "model" in ck else ck, strict=False)     # base_trainer.py:143
        if self.rank == 0:
            print(f"Model preloaded successfully from {model_path}.")

    def _save_checkpoint(self, epoch, is_best_epoch=False):
        state = {"epoch": epoch, "best_score": self.best_score,
                 "optimizer": self.engine.optimizer_state_dict(),          # torch.optim.Adam.state_dict() layout
                 "scaler": {},                                              # GradScaler(enabled=False).state_dict()
                 "model": {k: v.detach().cpu() for k, v in self.model.state_dict().items()}}
        torch.save(state, os.path.join(self.checkpoints_dir, "latest_model.tar"))
        torch.save(state["model"], os.path.join(self.checkpoints_dir, f"model_{str(epoch).zfill(4)}.pth"))
        if is_best_epoch:
            torch.save(state, os.path.join(self.checkpoints_dir, "best_model.tar"))

    def _resume_checkpoint(self):
        path = os.path.join(self.checkpoints_dir, "latest_model.tar")
        assert os.path.exists(path), f"{path} does not exist, can not load latest checkpoint."
        if self.dist is not None and self.dist.is_initialized():
            self.dist.barrier()                                          # base_trainer.py:161
        ck = torch.load(path, map_location="cpu", weights_only=False)
        self.start_epoch = ck["epoch"] + 1
        self.best_score = ck["best_score"]
        self.engine.load_optimizer_state_dict(ck["optimizer"])
        self.model.load_state_dict(ck["model"])
        if self.rank == 0:
            print(f"Model checkpoint loaded. Training will begin at {self.start_epoch} epoch.")

    def _is_best_epoch(self, score, save_max_metric_score=True):        # base_trainer.py:234-246
        if save_max_metric_score and score >= self.best_score:
            self.best_score = score
            return True
        if not save_max_metric_score and score <= self.best_score:
            self.best_score = score
            return True
        return False

    def _train_epoch(self, epoch):
        import gc
        sampler = getattr(self.train_dataloader, "sampler", None)
        if hasattr(sampler, "set_epoch"):
            sampler.set_epoch(epoch)                                     # a different shuffle every epoch
        import sys
        nb, frames = 0, 0
        self.engine.mean_loss(reset=True)
        gc.collect()     # (~50 ms over torch's object graph: before the clock starts)
        gc.disable()     # a generational GC pause (tens of ms) is long enough to drain the device queue of eager launches
        # the prefetcher's staging thread shares the GIL with this loop: at the default 5 ms switch interval every hand-over of the GIL
        # between the two costs up to a whole training step (measured: 41 ms per batch instead of 4.2 with a host DataLoader)
        swi = sys.getswitchinterval()
        sys.setswitchinterval(2e-4)
        t0 = time.time()
        try:
            for noisy, clean in _Prefetcher(self.train_dataloader, self.device, use_ring=self.shm_ring):
                self.engine.step(noisy, clean)                           # no host synchronisation inside the loop
                nb += 1
                frames += noisy.shape[0] * (1 + noisy.shape[1] // self.engine.hop)
        finally:
            gc.enable()
            sys.setswitchinterval(swi)
        mean = self.engine.mean_loss(reset=True)                         # one synchronisation per epoch
        skipped = self.engine.skipped_steps()
        # a timed-out step was skipped on every rank (parameters intact): tolerated up to [meta] max_gru_timeouts_per_epoch
        # (default 0), checked AFTER the epoch's checkpoint is written (train())
        self._pending_health = True
        dt = time.time() - t0
        self.last_epoch_frames_per_s = frames / max(dt, 1e-9)             # (per rank; bench.py secondary.trainer_path reads it)
        if self.rank == 0:
            note = f"  ({skipped} optimizer steps skipped so far: non-finite loss / gradient)" if skipped else ""
            print(f"[epoch {epoch}] loss {mean:.6f}  {frames / max(dt, 1e-9):.0f} frames/s/rank{note}")
        return mean

    @torch.no_grad()
    def _validation_epoch(self, epoch):
        total, nb = 0.0, 0
        for noisy, clean in self.validation_dataloader:
            noisy = noisy.to(self.device).float().contiguous()
            clean = clean.to(self.device).float().contiguous()
            total += self.engine.eval_loss(noisy, clean)
            nb += 1
        return total / max(nb, 1)

    def train(self):
        for epoch in range(self.start_epoch, self.epochs + 1):
            if self.only_validation and self.rank == 0:                  # base_trainer.py:386-396
                self.model.eval()
                score = self._validation_epoch(epoch)
                print(f"[epoch {epoch}] validation loss {score:.6f}")
                if self._is_best_epoch(score, save_max_metric_score=self.save_max_metric_score):
                    self._save_checkpoint(epoch, is_best_epoch=True)
                continue
            if self.only_validation:
                continue
            self.model.train()
            self._train_epoch(epoch)
            if self.rank == 0 and epoch % self.save_checkpoint_interval == 0:
                self._save_checkpoint(epoch)
            if getattr(self, "_pending_health", False):
                self._pending_health = False
                new = self.engine.check_health(self.max_gru_timeouts_per_epoch)      # raises on EVERY rank together
                if new and self.rank == 0:
                    print(f"[epoch {epoch}] {new} step(s) skipped on all ranks after a GRU hand-off time-out")
            if self.rank == 0 and epoch % self.validation_interval == 0 and self.validation_dataloader is not None:
                self.model.eval()
                score = self._validation_epoch(epoch)
                print(f"[epoch {epoch}] validation loss {score:.6f}")
                if self._is_best_epoch(score, save_max_metric_score=self.save_max_metric_score):
                    self._save_checkpoint(epoch, is_best_epoch=True)     # rewrites latest_model.tar with the new best_score
                self.model.train()
