"""The training step the reference never wrote (train/trainer_casual.py is empty):

    STFT(noisy), STFT(clean) -> unet_2 -> mask * spectrum -> loss -> backward
    -> bucketed gradient all-reduce (RCCL, overlapped with backward) -> clip -> Adam

run on one MI355X per process.  Parameters, gradients and Adam moments live in flat,
64-float-aligned buffers ordered by GRADIENT BUCKET (the order the backward pass finishes them), so a bucket is
one contiguous slice = one collective; the kernel sequence is recorded once into HIP graphs and replayed.

Data-parallel schedule (world > 1; SURVEY 8e, base_trainer.py:31 is DDP's bucketed overlap):

    graph 0: STFTs, forward, loss, decoder backward, GGRU backward        -> all-reduce(bucket 0) on RCCL's stream
    graph 1: layer-1 GRU dW GEMMs (side stream) | encoder levels L..L/2+1 -> all-reduce(bucket 1)
    graph 2: encoder levels L/2..1                                          -> all-reduce(bucket 2)
    wait for the three collectives -> [gradient norm] -> fused Adam (1/world folded in)

so the 10 MB of bucket 0 travel while graph 1 runs and the 10 MB of bucket 1 while graph 2 runs; only the encoder's
130 kB are exposed.  With world == 1 the whole step is one graph (no boundaries).
"""
from __future__ import annotations

import os
from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist
import torch.nn as nn

from . import config as _config
from . import ops
from .config import EngineConfig
from .model import cruse_net as _net
from .model.cruse_net import N_BUCKETS, bucket_of, unet2_backward, unet2_forward, unet_2

ALIGN = 64  # floats

LOSSES = ("wo_male", "si_snr", "sdnr", "wo_male_df", "l1", "mse")
WAVE_LOSSES = ("si_snr", "l1", "mse")            # waveform in -> waveform loss: the estimate goes through the iSTFT


def unused_parameter(name: str) -> bool:
    """`fc` (cruse_net.py:146) and `bn1_t` (:142 after R3) never receive gradients."""
    return name.startswith("fc.") or name.startswith("bn1_t.")


def _dist_on() -> bool:
    """a process group with peers -- or CRUSE_FORCE_COLLECTIVES=1, which makes a world of ONE issue its collectives
    anyway (the RCCL code path of the bucketed step can then be exercised on a single-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()):
        return False
    return dist.get_world_size() > 1 or os.environ.get("CRUSE_FORCE_COLLECTIVES") == "1"


class FlatParams:
    """Views of a model's trainable parameters inside one contiguous buffer (plus grads / Adam state), laid out
    bucket by bucket (`bucket(name)` -> int; buckets are contiguous, 64-float aligned slices).

    Works on any device (the CPU/gloo tests exercise the layout and the collectives)."""

    def __init__(self, model: nn.Module, skip=unused_parameter, bucket: Callable[[str], int] = bucket_of,
                 n_buckets: int = N_BUCKETS):
        named = [(n, p) for n, p in model.named_parameters() if not skip(n)]
        self.order = [n for n, _ in model.named_parameters()]            # torch.optim parameter indices
        self.shapes = {n: tuple(p.shape) for n, p in model.named_parameters()}
        self.bucket_id = {n: int(bucket(n)) for n, _ in named}
        assert all(0 <= b < n_buckets for b in self.bucket_id.values())
        named.sort(key=lambda np_: self.bucket_id[np_[0]])                # stable: state-dict order inside a bucket
        self.names = [n for n, _ in named]
        self.offsets: Dict[str, int] = {}
        self.bucket_range: List[Tuple[int, int]] = []
        off = 0
        for b in range(n_buckets):
            start = off
            for n, p in named:
                if self.bucket_id[n] == b:
                    self.offsets[n] = off
                    off += (p.numel() + ALIGN - 1) // ALIGN * ALIGN
            self.bucket_range.append((start, off))
        self.total = off
        dev = named[0][1].device
        self.params = torch.zeros(off, device=dev, dtype=torch.float32)
        self.grads = torch.zeros(off, device=dev, dtype=torch.float32)
        self.exp_avg = torch.zeros(off, device=dev, dtype=torch.float32)
        self.exp_avg_sq = torch.zeros(off, device=dev, dtype=torch.float32)
        self.P: Dict[str, torch.Tensor] = {}
        self.G: Dict[str, torch.Tensor] = {}
        with torch.no_grad():
            for n, p in named:
                o, k = self.offsets[n], p.numel()
                view = self.params[o:o + k].view(p.shape)
                view.copy_(p)
                p.data = view                       # the module now reads the flat buffer
                self.P[n] = view
                self.G[n] = self.grads[o:o + k].view(p.shape)
        # parameters outside the flat buffer (never trained) are still needed by name
        for n, p in model.named_parameters():
            if n not in self.P:
                self.P[n] = p.data

    def view(self, flat: torch.Tensor, name: str) -> torch.Tensor:
        o = self.offsets[name]
        shape = self.shapes[name]
        k = 1
        for d in shape:
            k *= d
        return flat[o:o + k].view(shape)

    def broadcast(self, src: int = 0) -> None:
        """identical initial weights on all ranks (cf. loss_func/distrib.py:57-72)."""
        if _dist_on():
            dist.broadcast(self.params, src)

    def bucket_grads(self, b: int) -> torch.Tensor:
        s, e = self.bucket_range[b]
        return self.grads[s:e]

    def all_reduce_bucket(self, b: int, async_op: bool = True):
        """SUM over ranks of one bucket; the 1/world_size factor is folded into the Adam kernel.  Returns the Work
        handle (None outside a process group or for an empty bucket)."""
        s, e = self.bucket_range[b]
        if not _dist_on() or e == s:
            return None
        return dist.all_reduce(self.grads[s:e], op=dist.ReduceOp.SUM, async_op=async_op)

    def all_reduce_grads(self) -> None:
        """all buckets, blocking (tests / un-bucketed callers)."""
        for w in [self.all_reduce_bucket(b) for b in range(len(self.bucket_range))]:
            if w is not None:
                w.wait()

    def attach_grads(self, model: nn.Module) -> None:
        """expose the flat gradients as param.grad (checkpoint / clip_grad_norm_ compatibility)."""
        for n, p in model.named_parameters():
            if n in self.G:
                p.grad = self.G[n]

    # ---- torch.optim.Adam.state_dict() layout (base_trainer.py:199-221 saves optimizer.state_dict()) -------------
    def adam_state_dict(self, step: int, lr, betas, eps, weight_decay) -> dict:
        state = {}
        for idx, n in enumerate(self.order):
            if n in self.offsets and step > 0:                 # torch creates state only for parameters that got a gradient
                state[idx] = {"step": torch.tensor(float(step)),
                              "exp_avg": self.view(self.exp_avg, n).detach().cpu().clone(),
                              "exp_avg_sq": self.view(self.exp_avg_sq, n).detach().cpu().clone()}
        group = {"lr": lr, "betas": tuple(betas), "eps": eps, "weight_decay": weight_decay, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": False, "params": list(range(len(self.order)))}
        return {"state": state, "param_groups": [group]}

    def load_adam_state_dict(self, sd: dict) -> int:
        """-> step count.  Accepts what torch.optim.Adam(model.parameters()).state_dict() holds (sizes validated)."""
        groups = sd["param_groups"]
        ids = [i for g in groups for i in g["params"]]
        if len(ids) != len(self.order):
            raise RuntimeError(f"optimizer state has {len(ids)} parameters, the model has {len(self.order)}")
        step = 0
        self.exp_avg.zero_(); self.exp_avg_sq.zero_()
        for pos, pid in enumerate(ids):
            st = sd["state"].get(pid)
            if st is None:
                continue
            n = self.order[pos]
            if n not in self.offsets:
                continue                                        # fc / bn1_t: never trained here
            for key, flat in (("exp_avg", self.exp_avg), ("exp_avg_sq", self.exp_avg_sq)):
                t = st[key]
                if tuple(t.shape) != self.shapes[n]:
                    raise RuntimeError(f"optimizer state {key} of {n}: shape {tuple(t.shape)} != {self.shapes[n]}")
                self.view(flat, n).copy_(t)
            step = max(step, int(float(st["step"])))
        return step


class TrainEngine:
    """loss: "wo_male" (loss_func/loss.py:121-148, alpha/beta), "si_snr" (train_base/loss.py:7-25 through the iSTFT),
    "l1" / "mse" (train_base/loss.py:3-4: torch.nn.L1Loss / MSELoss on the enhanced waveform iSTFT(mask * N) vs the clean one),
    "sdnr" (loss_func/loss.py:151-175 with the mask as gain; `snr_db`, `sdnr_beta_db`) or "wo_male_df" -- BASELINE
    config 4: the mask is the real part of a DeepFilter(t_dim=1, f_dim=5) coefficient field (model/deep_filter.py:15-41;
    DECISION recorded in oracle.train_step_loss: filters = (mask padded to 161 bins, 0)), the enhanced spectrum is the
    filter output and WO-MALE is taken on it.
    use_graph: replay the step from HIP graph(s) (True) or launch its ~170 kernels eagerly (False).  On the bench step the
    eager form is the faster one once the host keeps ahead (6.08 vs 6.3 ms); a busy or slow host favours the
    graph.  "auto" decides by measurement on the first real steps: 1 + 3 steps from the graph, 1 + 3 launched eagerly
    (HIP-event times of the measured ones, one host synchronisation each), then the faster form is kept for good -- no
    extra steps are run, every rank measures and the verdict is taken on the MAX over the ranks.  bench.py does the same during its warm-up.
    clip_grad_norm > 0: torch.nn.utils.clip_grad_norm_ semantics on the (averaged) gradient, folded into Adam.
    bucketed: None = when world > 1; True forces the segmented schedule (tests, single-GPU cost measurements).
    config: EngineConfig (cruse_amd/config.py) -- scheduling / numerics options; None = the measured-best defaults."""

    def __init__(self, model: unet_2, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0,
                 n_fft=320, hop=160, precision: Optional[str] = None, use_graph=True,
                 loss_alpha=2.0, loss_beta=1.0, loss: str = "wo_male", clip_grad_norm: float = 0.0,
                 snr_db: float = 0.0, sdnr_beta_db: float = 20.0, bucketed: Optional[bool] = None,
                 config: Optional[EngineConfig] = None):
        if not torch.cuda.is_available():
            raise RuntimeError("cruse_amd.TrainEngine needs a HIP device (there is no CPU path)")
        self.model = model
        # this engine's own options and side-stream scheduler (config.py): nothing is shared with another engine of the process
        self.cfg = config.copy() if config is not None else EngineConfig()
        self.side = _net._SideStream(self.cfg)
        self.n_fft, self.hop = n_fft, hop
        self.f_net = (n_fft // 2 + 1) // 2 * 2
        self.f_stft = n_fft // 2 + 1
        if precision is not None:
            model.set_precision(precision)
        self.prec = model.precision
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.loss_alpha, self.loss_beta = loss_alpha, loss_beta
        self.snr_db, self.sdnr_beta_db = snr_db, sdnr_beta_db
        if loss not in LOSSES:
            raise ValueError(f"unknown loss {loss!r} ({' | '.join(LOSSES)})")
        self.loss = loss
        self.clip = float(clip_grad_norm or 0.0)
        self.flat = FlatParams(model)
        self.flat.broadcast(0)
        self.Bf = dict(model.named_buffers())
        self.world = dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1
        self.bucketed = (self.world > 1) if bucketed is None else bool(bucketed)
        self.step_count = 0
        if not (isinstance(use_graph, bool) or use_graph == "auto"):
            raise ValueError(f"use_graph must be True, False or 'auto', not {use_graph!r}")
        self._auto = {"n": 0, "t": {True: [], False: []}} if use_graph == "auto" else None
        self.use_graph = True if use_graph == "auto" else use_graph
        self.launch_form_timing = None           # {"graph_ms", "eager_ms", "kept"} once "auto" has decided
        self._graphs = None
        self._launch_stream = None               # the stream the current capture is replayed from (None: the current stream)
        self.launch_stream_timing = None
        self._static = None
        self._shape = None
        self._graph_cache: Dict[Tuple[int, ...], tuple] = {}     # input shape -> (graphs, static inputs, static loss)
        dev = self.flat.params.device
        self._gsumsq = torch.zeros(1, device=dev, dtype=torch.float64)
        # [0] skipped steps, [1] of them: a GRU hand-off timed out, [2] of them: the loss was not finite
        self._skipped = torch.zeros(4, device=dev, dtype=torch.int32)
        # per-step health words [time-out, non-finite loss]: decided on the device, all-reduced (MAX) over the ranks, so
        # that every replica takes the SAME skip decision (a rank-local skip would let the replicas drift apart for good)
        self._health = torch.zeros(2, device=dev, dtype=torch.int32)
        self._timeouts_seen = 0
        self._norms: Dict[Tuple[int, ...], float] = {}           # input shape -> loss normalisation of that shape
        self._loss_acc = torch.zeros(2, device=dev, dtype=torch.float64)     # [sum of the applied steps' losses, applied steps]
        self._norm = 1.0
        self._works: List = []
        self._launcher = None

    # -- forward + loss (+ dL/dlogit when training) ------------------------------------------------------------------
    def _forward_loss(self, noisy: torch.Tensor, clean: torch.Tensor, training: bool):
        B, L = noisy.shape
        T = ops.stft_frames(L, self.hop)
        if self.loss == "wo_male_df":
            # config 4: the DeepFilter loss reads both spectra as [2,B,T,F] plane pairs -- the STFTs write them in place
            unp = torch.empty(2, B, T, self.f_stft, device=noisy.device, dtype=torch.float32)
            mag = torch.empty(B, T, self.f_net, device=noisy.device, dtype=torch.float32)
            ops.stft(noisy, self.n_fft, self.hop, mag_eps=1e-8, out=(unp[0], unp[1], mag))
            nre, nim = unp[0], unp[1]
        else:
            unp = None
            nre, nim, mag = ops.stft(noisy, self.n_fft, self.hop, mag_bins=self.f_net, mag_eps=1e-8)
        cre = cim = cmag = ref = None
        # the clean spectrum is only needed by the loss: a leaf queued for the first forward recurrence (beside the encoder
        # it shared HBM with the 1 -> 8 conv: 60 vs 33 us, plus an event record on the main stream); joined by
        # unet2_forward before the decoder
        if self.loss not in WAVE_LOSSES:
            # (outputs are allocated here, on the main stream, so that the leaf itself allocates nothing)
            if self.loss == "wo_male":
                cmag = torch.empty(B, T, self.f_stft, device=clean.device, dtype=torch.float32)
                self.side.defer(lambda: ops.stft(clean, self.n_fft, self.hop, want_ri=False, mag_eps=0.0, out=(None, None, cmag)),
                           clean, cmag, kind=1, lane=0)
            else:
                if self.loss == "wo_male_df":
                    ref = torch.empty(2, B, T, self.f_stft, device=clean.device, dtype=torch.float32)
                    cre, cim = ref[0], ref[1]
                else:
                    cre = torch.empty(B, T, self.f_stft, device=clean.device, dtype=torch.float32)
                    cim = torch.empty_like(cre)
                self.side.defer(lambda: ops.stft(clean, self.n_fft, self.hop, out=(cre, cim, None)), clean, cre, cim, kind=1, lane=0)
        mask, ctx = unet2_forward(mag.view(B, 1, T, self.f_net), self.flat.P, self.Bf, self.model.ch,
                                  self.model.rnn_groups, self.prec, training=training, save=training,
                                  update_running=training)
        self._last_mask = mask
        rows = B * T
        dlogit = None
        if self.loss == "wo_male":
            loss_sum, _, dlogit, _, _ = ops.mask_loss(mask, nre, nim, cmag, rows, self.f_net, self.f_stft,
                                                      self.loss_alpha, self.loss_beta, want_dlogit=training)
            self._norm = float(rows * self.f_stft)
        elif self.loss == "sdnr":
            loss_sum, _, dlogit = ops.mask_sdnr(mask, cre, cim, nre, nim, rows, self.f_net, self.f_stft, B, self.snr_db,
                                                self.sdnr_beta_db, want_dlogit=training)
            self._norm = float(B * self.f_stft)
        elif self.loss == "wo_male_df":
            loss_sum, dlogit = self._deepfilter_loss(mask, unp, ref, B, T, training)
            self._norm = float(rows * self.f_stft)
        else:
            # waveform in -> waveform loss: est = iSTFT(mask * N); SI-SNR(est, clean) and back through both
            ere, eim = ops.mask_apply(mask, nre, nim, rows, self.f_net, self.f_stft)
            est = ops.istft(ere.view(B, T, self.f_stft), eim.view(B, T, self.f_stft), self.n_fft, self.hop, L)
            if self.loss == "si_snr":
                loss_sum, coef = ops.sisnr_fwd(est, clean)
                dwave = ops.sisnr_bwd(est, clean, coef) if training else None
                self._norm = 1.0
            else:                                                     # reduction "mean" over all B * L samples
                loss_sum, dwave = ops.wave_l1_mse(est, clean, self.loss == "mse", want_grad=training)
                self._norm = float(B * L)
            if training:
                dre, dim = ops.istft_bwd(dwave, T, self.n_fft, self.hop)
                dlogit = ops.mask_apply_bwd(dre, dim, nre, nim, mask, rows, self.f_net, self.f_stft)
        return loss_sum, dlogit, ctx

    def _deepfilter_loss(self, mask, unp, ref, B, T, training):
        """DeepFilter(1, 5) head + WO-MALE on its output.  The spectra are frame-major [B,T,F]; the kernel takes [B,F',T']
        planes, so it is called with the roles of the two axes (and of t_dim / f_dim) exchanged -- no transposition."""
        from ._lib import check, lib
        Fs, Fn, rows = self.f_stft, self.f_net, B * T
        dev = mask.device
        key = (B, T)
        nre, nim = unp[0], unp[1]
        if getattr(self, "_df_const", None) is None or self._df_const[0] != key:
            self._df_const = (key, torch.ones(rows, Fs, device=dev), torch.zeros(rows, Fs, device=dev))
        _, ones, zeros = self._df_const
        hr, _ = ops.mask_apply(mask, ones, zeros, rows, Fn, Fs)              # mask padded with the zero Nyquist bin (R8)
        hi = zeros
        f_dim, t_dim = 5, 1
        est = torch.empty(2, B, T, Fs, device=dev)
        o_r, o_i = ops.deepfilter_fwd(nre.view(B, T, Fs), nim.view(B, T, Fs), hr.view(B, T, Fs), hi.view(B, T, Fs), t_dim, f_dim,
                                      out=(est[0], est[1]))
        self._last_est = est
        loss_sum = torch.empty(1, device=dev, dtype=torch.float64)
        dest = torch.empty_like(est) if training else None
        TF = T * Fs
        check(lib.cruse_wo_male_spec(ops._p(ref), ops._p(est), ops._p(unp), B, TF, TF, B * TF, self.loss_alpha, self.loss_beta,
                                     1.0 / float(rows * Fs), ops._p(loss_sum), ops._p(dest), ops._stream()))
        if not training:
            return loss_sum, None
        _, _, dhr, _ = ops.deepfilter_bwd(dest[0], dest[1], nre.view(B, T, Fs), nim.view(B, T, Fs), hr.view(B, T, Fs),
                                          hi.view(B, T, Fs), t_dim, f_dim)
        dlogit = ops.mask_apply_bwd(dhr.view(rows, Fs), zeros, ones, zeros, mask, rows, Fn, Fs)
        return loss_sum, dlogit

    # -- one forward + loss + backward, gradients left in flat.grads --------------------
    def _fwd_bwd(self, noisy: torch.Tensor, clean: torch.Tensor, boundary=None) -> torch.Tensor:
        B, L = noisy.shape
        T = ops.stft_frames(L, self.hop)
        g = self.model.rnn_groups
        with _config.use(self.cfg), _net.use_scheduler(self.side), ops.ARENA.step(noisy.device), \
                _net.STEP_SCRATCH.step(B, g, self.model.hidden_size // g, noisy.device):      # (one arena clear, one scratch clear per step)
            loss_sum, dlogit, ctx = self._forward_loss(noisy, clean, training=True)
            ops.zero_(self.flat.grads)
            unet2_backward(ctx, dlogit.view(B, T, 1, self.f_net), self.flat.P, self.flat.G, boundary=boundary)
        return loss_sum

    @torch.no_grad()
    def eval_loss(self, noisy: torch.Tensor, clean: torch.Tensor) -> float:
        """validation: the configured loss with BatchNorm in eval mode (running statistics), no gradients."""
        norm = self._norm
        with _config.use(self.cfg), _net.use_scheduler(self.side):
            loss_sum, _, _ = self._forward_loss(noisy, clean, training=False)
        v = float(loss_sum.item()) / self._norm
        self._norm = norm
        return v

    # -- graph capture: one graph, or one per segment when the step is bucketed ----------------------------------
    def _capture(self, noisy, clean):
        """Record the step for this input shape and decide where it is replayed from.

        A replay runs the graph's main branch on the launching stream and its side branches on streams the graph created for itself
        when it was instantiated.  ROCm places all streams on a handful of hardware queues in creation order, and a queue whose head
        is a pending wait holds its pipe for a time slice: when a side branch's stream shares the queue / pipe of the launching
        stream the branches serialise (round 6: 5.5 instead of 4.7 ms for one capture in eight on the default stream;
        cruse_amd/streams.py has the eager-launch side of the same story).  Nothing tells which queue the graph's own streams got,
        so it is measured (_pick_launch_stream): the step's real pattern -- a small kernel on the caller's stream, replay from
        launcher L, the caller's stream waits for L -- is timed from the caller's stream and three pool streams and the fastest
        launcher is kept (the caller's own unless another is > 3 % faster).  The timing replays are forward + backward passes of the
        static batch: they rewrite the gradients (the real replay follows) and the BatchNorm running statistics, which are restored.
        NOT curable from here: a HIGH-PRIORITY caller stream.  Replayed from it, or merely waiting for a launcher, it slows every
        capture's side branches (3.5 -> 5.0 ms, 4.8 -> 6.0 ms whatever the launcher; re-instantiating does not help): graph replays
        belong on normal-priority streams (launch_stream_timing["caller_priority"] records what the capture saw)."""
        self._static = (noisy.clone(), clean.clone())
        # the warm-up run below must leave no trace: BatchNorm running statistics and counters are restored
        saved = {k: v.clone() for k, v in self.Bf.items()}
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):           # warm-up on a side stream (allocations, LDS attributes)
            self._fwd_bwd(*self._static, boundary=self._warmup_boundary if self.bucketed else None)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        for k, v in saved.items():
            self.Bf[k].copy_(v)
        graphs, static_loss = self._capture_once()
        launcher = None
        if self.cfg.pick_launch_stream:
            launcher, t_pattern, t_pure, tried = self._pick_launch_stream(graphs)
            for k, v in saved.items():
                self.Bf[k].copy_(v)
            self.launch_stream_timing = {"pure_replay_ms": tried[0], "step_pattern_ms": tried[1], "kept_ms": round(t_pattern, 3),
                                         "kept": "current" if launcher is None else "other",
                                         "caller_priority": getattr(torch.cuda.current_stream(), "priority", None)}
        self._graphs, self._static_loss, self._launch_stream = graphs, static_loss, launcher
        self._shape = tuple(noisy.shape)
        # one capture per input shape (a last, partial batch of an epoch would otherwise force two re-captures per epoch)
        self._graph_cache[self._shape] = (graphs, self._static, self._static_loss, self._launch_stream)
        self._norms[self._shape] = self._norm

    def _capture_once(self):
        graphs: List[Tuple[torch.cuda.CUDAGraph, int]] = []
        pool = torch.cuda.graph_pool_handle()
        cur: Dict[str, object] = {}

        def begin():
            g = torch.cuda.CUDAGraph()
            # thread_local: ProcessGroupNCCL's watchdog thread may query its events while this thread captures
            cm = torch.cuda.graph(g, pool=pool, capture_error_mode="thread_local")
            cm.__enter__()
            cur["g"], cur["cm"] = g, cm

        def end(bucket: int):
            cur["cm"].__exit__(None, None, None)
            graphs.append((cur["g"], bucket))

        def boundary(bucket: int):
            self.side.join(flush=False)           # every ISSUED leaf joined; queued leaves move to the next segment
            end(bucket)
            begin()
            self.side.flush()

        begin()
        try:
            static_loss = self._fwd_bwd(*self._static, boundary=boundary if self.bucketed else None)
        except BaseException:
            cur["cm"].__exit__(None, None, None)
            raise
        end(N_BUCKETS - 1)
        _net.release_capture_events()         # (kept alive until every segment's capture has ended: see cruse_net.record_event)
        return graphs, static_loss

    def _pick_launch_stream(self, graphs):
        """-> (launcher or None for the current stream, ms of the step pattern from it, ms of the fastest back-to-back replay,
        [pure times, pattern times] by candidate)"""
        cur = torch.cuda.current_stream()
        cands, seen = [cur], {int(cur.cuda_stream)}
        for _ in range(6):
            c = torch.cuda.Stream()
            if int(c.cuda_stream) not in seen and len(cands) < 4:
                seen.add(int(c.cuda_stream))
                cands.append(c)

        def replay_on(stream):
            with torch.cuda.stream(stream):
                for g, _ in graphs:
                    g.replay()

        def pure(stream, n=2):
            stream.wait_stream(cur)
            replay_on(stream)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(n):
                replay_on(stream)
            e1.record(stream)
            e1.synchronize()
            cur.wait_stream(stream)
            return e0.elapsed_time(e1) / n

        def pattern(stream, n=2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for i in range(n + 1):
                if i == 1:
                    e0.record(cur)
                ops.zero_(self._gsumsq)               # (stands for the step's own kernels on the caller's stream: input copies, Adam)
                if stream is cur:
                    replay_on(cur)
                else:
                    stream.wait_stream(cur)
                    replay_on(stream)
                    cur.wait_stream(stream)
            e1.record(cur)
            e1.synchronize()
            return e0.elapsed_time(e1) / n

        t_pure = [pure(c) for c in cands]
        t_pat = [pattern(c) for c in cands]
        k = min(range(len(cands)), key=lambda i: t_pat[i])
        if k != 0 and t_pat[k] > 0.97 * t_pat[0]:
            k = 0                                       # the caller's own stream unless another launcher is clearly faster
        return (None if k == 0 else cands[k]), t_pat[k], min(t_pure), [[round(t, 3) for t in t_pure], [round(t, 3) for t in t_pat]]

    def _warmup_boundary(self, bucket: int):
        self.side.join(flush=False)
        self.side.flush()

    def _plain_boundary(self, bucket: int):
        """eager launches: the bucket's collective must follow everything issued so far on the main AND the side streams,
        but the main stream itself need not wait for the leaves -- the collective is issued from a launcher stream that waits
        for both (ProcessGroupNCCL orders its stream after the stream it is called from)."""
        if not _dist_on():
            self.side.flush()
            return
        main = torch.cuda.current_stream()
        if self._launcher is None:
            # a stream proven to run beside the main AND the leaf stream (cruse_amd/streams.py): the launcher's head is a pending wait
            # for most of a segment, and a waiting queue that shares their hardware queue / pipe stalls them (+4 % per step, round 6)
            from . import streams
            self._launcher = streams.stream_beside(main, avoid=tuple(self.side.used), tag="launcher")
        self._launcher.wait_stream(main)
        for s_ in self.side.used:
            self._launcher.wait_stream(s_)
        with torch.cuda.stream(self._launcher):
            self._launch_bucket(bucket)
        self.side.flush()

    def _launch_bucket(self, b: int):
        """Start the all-reduce of every bucket that became final with segment b.  Un-bucketed graphs end with the
        last bucket id: everything goes at once."""
        first = b if self.bucketed else 0
        for i in range(first, b + 1):
            w = self.flat.all_reduce_bucket(i, async_op=True)
            if w is not None:
                self._works.append(w)

    _AUTO_PLAN = (True, True, True, True, False, False, False, False)      # first step of each form is not measured

    def _auto_step(self, noisy, clean):
        st = self._auto
        i = st["n"]
        form = self._AUTO_PLAN[i]
        self.use_graph = form and not st.get("capture_failed")
        measured = i not in (0, 4)
        if measured:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        self._auto = None                        # (step() below must not recurse into the tuner)
        out = self.step(noisy, clean)
        self._auto = st
        if form and not self.use_graph:
            # capture failed on THIS rank: step() has fallen back to eager launches.  The plan is still walked to its end -- the verdict
            # below is a collective (a rank that left early let the others wait in all_reduce for ever) -- with graph
            # times of +inf, so that every rank keeps eager launches
            st["capture_failed"] = True
        if st.get("capture_failed"):
            self.use_graph = False
        if measured:
            e1.record()
            e1.synchronize()
            st["t"][form].append(float("inf") if (form and st.get("capture_failed")) else e0.elapsed_time(e1))
        st["n"] = i + 1
        if st["n"] == len(self._AUTO_PLAN):
            tg, te = sorted(st["t"][True])[1], sorted(st["t"][False])[1]          # medians of three
            if _dist_on():
                # per-rank measurements, the job decides on the SLOWEST rank's figures (MAX): the same verdict on every rank
                tt = torch.tensor([tg, te], device=noisy.device, dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                tg, te = float(tt[0].item()), float(tt[1].item())
            self.use_graph = tg <= te
            self.launch_form_timing = {"graph_ms": round(tg, 3) if tg != float("inf") else None, "eager_ms": round(te, 3),
                                       "kept": "graph" if self.use_graph else "eager"}
            self._auto = None
        return out

    def step(self, noisy: torch.Tensor, clean: torch.Tensor) -> torch.Tensor:
        """One optimizer step; returns the (device, f64) loss sum -- divide by .loss_norm for the loss."""
        if self._auto is not None:
            return self._auto_step(noisy, clean)
        self._works = []
        if self.use_graph:
            if self._graphs is None or self._shape != tuple(noisy.shape):
                hit = self._graph_cache.get(tuple(noisy.shape))
                if hit is not None:
                    self._graphs, self._static, self._static_loss, self._launch_stream = hit
                    self._shape = tuple(noisy.shape)
                else:
                    try:
                        self._capture(noisy, clean)
                    except RuntimeError as ex:
                        # still the HIP path, just launched kernel by kernel: say so loudly and carry on
                        import sys
                        print(f"[cruse_amd] HIP-graph capture failed ({str(ex)[:200]}); continuing WITHOUT graphs", file=sys.stderr, flush=True)
                        self.use_graph = False
                        self._graphs = None
                        self.side.join()
                        torch.cuda.synchronize()
                        return self.step(noisy, clean)
            self._static[0].copy_(noisy)
            self._static[1].copy_(clean)
            self._norm = self._norms[tuple(noisy.shape)]      # (a cache hit runs no Python forward: the norm is per shape)
            ls = self._launch_stream
            if ls is None:
                for g, bucket in self._graphs:
                    g.replay()
                    self._launch_bucket(bucket)      # RCCL's stream waits for the replay; the next replay overlaps it
            else:
                # replayed from the stream _pick_launch_stream measured as the fastest launcher of THIS capture; the collectives are
                # issued from it too (ProcessGroupNCCL orders its stream behind the stream it is called from)
                cur = torch.cuda.current_stream()
                ls.wait_stream(cur)
                with torch.cuda.stream(ls):
                    for g, bucket in self._graphs:
                        g.replay()
                        self._launch_bucket(bucket)
                cur.wait_stream(ls)
            loss_sum = self._static_loss
        else:
            loss_sum = self._fwd_bwd(noisy, clean, boundary=self._plain_boundary if self.bucketed else None)
            self._norms[tuple(noisy.shape)] = self._norm
            self._launch_bucket(N_BUCKETS - 1)
        # health of THIS step: the GRU status word is latched and cleared (a transient hand-off time-out costs one step,
        # not the rest of the epoch), a non-finite loss is flagged, the running loss takes this step's own normalisation
        B = noisy.shape[0]
        g = self.model.rnn_groups
        word = ops.gru_status_word(noisy.device, B, g, self.model.hidden_size // g)
        ops.step_health(word, loss_sum, self._health)
        if _dist_on():
            # one rank's time-out / NaN is inside everybody's reduced gradient: everybody skips
            self._works.append(dist.all_reduce(self._health, op=dist.ReduceOp.MAX, async_op=True))
        for w in self._works:
            w.wait()                             # the compute stream waits for the collectives (no host block on RCCL)
        self.step_count += 1
        gs = None
        if self.clip > 0.0 or _dist_on():        # the norm of the REDUCED gradient: identical on every rank
            gs = ops.sumsq(self.flat.grads, out=self._gsumsq)
        ops.adam_step(self.flat.params, self.flat.grads, self.flat.exp_avg, self.flat.exp_avg_sq, self.lr,
                      self.betas[0], self.betas[1], self.eps, self.wd, self.step_count, 1.0 / self.world,
                      max_norm=self.clip, gsumsq=gs, skip_flag=self._health, skipped=self._skipped,
                      loss_sum=loss_sum, loss_scale=1.0 / self._norm, loss_acc=self._loss_acc)
        return loss_sum

    # -- host-side read-outs (each synchronises) --------------------------------------------------------------------
    @property
    def loss_norm(self) -> float:
        return self._norm

    def loss_value(self, loss_sum: torch.Tensor) -> float:
        return float(loss_sum.item()) / self._norm

    def mean_loss(self, reset: bool = True) -> float:
        """mean loss over the steps since the last reset -- ONE synchronisation per epoch / log interval."""
        acc = self._loss_acc.tolist()            # (each APPLIED step was accumulated with its own normalisation, by the guarded
        v = acc[0] / max(acc[1], 1.0)            #  Adam itself: a skipped step -- time-out, NaN, on any rank -- adds nothing)
        if reset:
            self._loss_acc.zero_()
        return v

    def skipped_steps(self) -> int:
        return int(self._skipped[0].item())

    def timeout_steps(self) -> int:
        """steps skipped because a GRU hand-off timed out on ANY rank (the count is the same on every rank)."""
        return int(self._skipped[1].item())

    def nonfinite_steps(self) -> int:
        return int(self._skipped[2].item())

    def check_health(self, max_new_timeouts: int = 0) -> int:
        """Steps in which a GRU hand-off timed out SINCE THE LAST CALL (those steps were skipped by the guarded Adam on every
        rank, so the parameters are intact); raises CRUSE_E_TIMEOUT when there are more than `max_new_timeouts` of them.  The
        count comes from the all-reduced health words, so all ranks of a data-parallel job see the same number and raise
        together (no rank is left waiting in a collective)."""
        total = self.timeout_steps()
        new = total - self._timeouts_seen
        self._timeouts_seen = total
        if new > max_new_timeouts:
            raise RuntimeError(f"cruse_hip error -5 (CRUSE_E_TIMEOUT): a GRU hand-off timed out in {new} step(s) since the last "
                               "health check -- the persistent recurrence kernel's workgroups were not co-resident (another "
                               "process or kernel holding CUs?); the affected optimizer steps were skipped on every rank")
        ops.check_gru_status()                   # a word set outside step() (inference, eval_loss)
        return new

    # -- optimizer state in torch.optim.Adam layout ------------------------------------------------------------------
    def optimizer_state_dict(self) -> dict:
        # Adam's step = the APPLIED steps (the kernel's bias corrections use step_count - skipped, see cruse_adam_step_guarded)
        return self.flat.adam_state_dict(self.step_count - self.skipped_steps(), self.lr, self.betas, self.eps, self.wd)

    def load_optimizer_state_dict(self, sd: dict) -> None:
        self.step_count = self.flat.load_adam_state_dict(sd) + self.skipped_steps()
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps = g["lr"], tuple(g["betas"]), g["eps"]
        self.wd = g.get("weight_decay", 0.0)
