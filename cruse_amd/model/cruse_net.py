"""MI355X-native `model.cruse_net`: GGRU and unet_2 with the reference's nn.Module surface.

Same class names, constructor arguments, state-dict keys and weight layouts as
model/cruse_net.py:14-55 (GGRU) and :129-165 (unet_2, repairs R1-R8 of SURVEY.md
section 8a), so `initialize_module("model.cruse_net.unet_2", args)` and checkpoints
keep working.  The stock torch.nn sub-modules are used ONLY as parameter containers
(their default initialisation consumes the RNG exactly as the reference would); every
forward/backward op runs in libcruse_hip.so through `cruse_amd.ops`.  There is no
CPU or eager-torch fallback: calling these modules on CPU tensors raises.

Internally activations are frame-major [B,T,C,F] (one 640-float row per frame at every
U-Net level), so GGRU's transpose(1,2)+view (cruse_net.py:39-40) costs nothing.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import contextlib
import torch
import torch.nn as nn

from .. import config, ops, streams

DEFAULT_PREC = "f32"


# Events recorded while a HIP graph is being captured are kept alive until the capture has ended (TrainEngine clears the list):
# hipStreamEndCapture (ROCm 7.2) walks the events that took part in the capture, and an event that torch had already destroyed --
# every temporary of Stream.wait_stream(), every Event dropped at the end of a loop iteration -- is then a dangling pointer there.
# Whether that bites depends on what the allocator did with the freed block: the step captured for four rounds, the backward
# wavefront's extra events made hipStreamEndCapture segfault deterministically ("Add EmptyNode", then a crash).
_KEEP_EVENTS: list = []


def record_event(stream) -> "torch.cuda.Event":
    e = torch.cuda.Event()
    e.record(stream)
    if torch.cuda.is_current_stream_capturing():
        if len(_KEEP_EVENTS) > 200000:
            del _KEEP_EVENTS[:100000]
        _KEEP_EVENTS.append(e)
    return e


def wait_stream(waiter, stream) -> None:
    """waiter.wait_stream(stream) with the temporary event kept alive during graph capture"""
    waiter.wait_event(record_event(stream))


def release_capture_events() -> None:
    _KEEP_EVENTS.clear()


class _SideStream:
    """Runs LEAF kernels (weight gradients, bias sums, skip convs) on a side HIP stream so they fill the ~96 CUs the persistent
    GRU kernels leave idle and overlap the HBM-bound main path elsewhere.  Leaves only read tensors produced on the main stream
    and write parameter gradients / tensors consumed after join(); they allocate nothing.  Tensors handed to the side stream are
    kept alive until join().  One scheduler per TrainEngine (engine.py installs its own with use_scheduler); the module-level
    default serves the autograd path of the nn.Modules.  EngineConfig.overlap = False disables it."""

    def __init__(self, cfg=None):
        cfg = cfg if cfg is not None else config.EngineConfig()
        self.enabled = bool(cfg.overlap)
        # leaves queued for the next recurrence launch instead of issued at once (EngineConfig.defer_mask)
        self.defer_mask = int(cfg.defer_mask)
        self.streams = {}              # (device, main stream) -> side stream proven to overlap with it
        self._capture_streams = {}     # the same for capture streams (unproven: a replay runs on the graph's own streams)
        self.keep = []
        self.deferred = []
        self.active = False
        self.used = []                 # side streams forked from the main stream since the last join()
        self.after_release = None      # one-shot callback run by the next release_around() once its leaves are issued

    def _next(self, lane=None):
        # The side stream is chosen PER MAIN STREAM by measurement (cruse_amd/streams.py): torch's pool streams share a handful of
        # hardware queues, and a side stream that lands on the main stream's queue (every fourth) serialises the leaves with the
        # chain they were meant to hide behind -- 5.8 instead of 3.5 ms per step.  During a HIP-graph capture any stream will do.
        main = torch.cuda.current_stream()
        key = (main.device_index, int(main.cuda_stream))
        capturing = torch.cuda.is_current_stream_capturing()
        table = self._capture_streams if capturing else self.streams
        s = table.get(key)
        if s is None:
            s = table[key] = streams.side_stream_for(main)
        if s not in self.used:         # only forked streams may be recorded on / joined (HIP-graph capture rule)
            self.used.append(s)
        return s

    def defer(self, fn, *tensors, kind=7, lane=None):
        """Queue a leaf for the next release_around(): it then starts WITH the next recurrence kernel (which leaves
        ~96 CUs idle for its whole duration) instead of competing with the throughput-bound kernels before it."""
        if not (self.enabled and (self.defer_mask & kind)):
            self.run(fn, *tensors, lane=lane)
            return
        self.deferred.append((fn, lane, None))
        self.keep.extend(tensors)

    def release_around(self, launch):
        """launch() issues a recurrence kernel on the main stream; the deferred leaves are issued right after it on
        the side stream, ordered only after the work that preceded the recurrence launch."""
        if not (self.enabled and self.deferred):
            return launch()
        main = torch.cuda.current_stream()
        ev = record_event(main)
        out = launch()
        waited = set()
        for fn, lane, dep in self.deferred:
            side = self._next(lane)
            if side not in waited:                # one wait per side stream (a wait in front of every leaf: ~6 us of queue time each; the
                side.wait_event(ev)               # step does not notice -- the leaves beside a recurrence are not what it waits for)
                waited.add(side)
            with torch.cuda.stream(side):
                fn()
        self.deferred.clear()
        self.active = True
        if self.after_release is not None:
            cb, self.after_release = self.after_release, None
            cb()
        return out

    def run(self, fn, *tensors, lane=None, dep=None):
        if not self.enabled:
            fn()
            return
        main = torch.cuda.current_stream()
        side = self._next(lane)
        wait_stream(side, main)
        self.keep.extend(tensors)
        self.active = True
        with torch.cuda.stream(side):
            fn()

    def mark(self):
        """Events after everything issued on the side stream so far (None when nothing runs there)."""
        if not (self.enabled and self.active):
            return None
        evs = []
        for s in self.used:
            evs.append(record_event(s))
        return evs

    def wait(self, evs):
        if evs is not None:
            for ev in evs:
                torch.cuda.current_stream().wait_event(ev)

    def flush(self):
        """Issue everything still queued by defer() on the side stream now."""
        if self.deferred:
            fns, self.deferred = self.deferred, []
            for fn, lane, dep in fns:
                self.run(fn, lane=lane)

    def join(self, flush: bool = True):
        """Main stream waits for the side stream.  flush=False keeps the deferred (not yet issued) leaves queued: a
        SEGMENT boundary of the bucketed data-parallel step (engine.py) ends a graph capture with every ISSUED leaf
        joined, and carries the queued ones into the next segment."""
        if flush and self.enabled:               # nothing left to hide behind: issue what is still queued
            self.flush()
        if self.enabled and self.active:
            for s in self.used:
                wait_stream(torch.cuda.current_stream(), s)
            self.used = []
            if not self.deferred:
                self.keep.clear()
            self.active = False


SIDE = _SideStream()


class use_scheduler:
    """with use_scheduler(side): the step issued inside uses THIS scheduler (TrainEngine owns one: two engines in a process do
    not share side-stream state)."""

    def __init__(self, side: "_SideStream"):
        self.side = side

    def __enter__(self):
        global SIDE
        self.prev, SIDE = SIDE, self.side
        return self.side

    def __exit__(self, *exc):
        global SIDE
        SIDE = self.prev


def _splitk_bf16(M: int, N: int, K: int) -> int:
    """k-slices of the K = 25 664 weight-gradient GEMMs.  Default: 8 slices PINNED to XCDs (negative splitk of
    cruse_gemm_bf16_nt) -- every XCD walks one k-range over all output tiles, so each operand byte leaves HBM about once
    (PMC: 217 MB per launch against 506 MB with (k-slice, n-tile) units dealt round-robin; 14.6 instead of 15.9 GB per
    step) for +0.02 ms of step time.  CRUSE_DW_XCDK=0: the round-robin form that fills the 256 CUs in one round with
    one block per CU (the fastest launch alone: 142 vs 206 us); CRUSE_DW_XCDK=<n>: n pinned slices."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    x = 8 if tiles * 8 >= 192 else 0      # few output tiles (grouped GRUs): keep round-robin
    if x > 1:
        return -x
    return max(1, min(256 // tiles, K // 1024))


def _bf16_gemm_path(prec, Hg: int) -> bool:
    """bf16-operand GEMMs: K = Hg and K = 3*Hg are rounded up to whole 64-deep tiles -- the weight operand is zero
    padded (K-tiled), the activation operand reads on into the next group / row (finite values, zero weights)."""
    return ops.prec_code(prec) == ops.PREC_BF16 and Hg % 32 == 0


class _StepScratch:
    """Recurrence panel scratches of ONE training step: the engine clears the four slots with one launch at the top of the step
    (ops.gru_step_ws_clear) and every recurrence launch of the step then takes the next slot with zeroed=True -- the memset in
    front of each recurrence (~7 us of kernel + ~7 us of gap, four times on the main stream) is gone.  Inactive outside
    `with STEP_SCRATCH.step(...)`, beyond four launches (time-chunk pipelines) and for a different shape: the launch then clears
    its own scratch as before."""

    def __init__(self):
        self.key = None
        self.next = 0

    @contextlib.contextmanager
    def step(self, B, g, Hg, device):
        ops.gru_step_ws_clear(B, g, Hg, device)
        self.key, self.next = (B, g, Hg), 0
        try:
            yield self
        finally:
            self.key = None

    def take(self, B, g, Hg, slot):
        """-> (slot, zeroed) for the next recurrence launch"""
        if self.key != (B, g, Hg) or slot != 0 or self.next >= ops.STEP_SLOTS:
            return slot, False
        self.next += 1
        return ops.STEP_SLOT0 + self.next - 1, True


STEP_SCRATCH = _StepScratch()


def _gi_x3_knob(Hg: int) -> int:
    knob = config.get().gi_x3
    return int(knob) if knob is not None else (7 if Hg <= 320 else 3)


def _gi_takes_bf16_copy(prec, Hg: int) -> bool:
    """The gate projections read ONE bf16 plane of their activation operand, unpadded: the producer (BatchNorm / LayerNorm
    kernel) can then write that copy itself instead of a separate cast pass."""
    return _bf16_gemm_path(prec, Hg) and Hg % 64 == 0 and not (_gi_x3_knob(Hg) & 4)


def _gi_f16(prec, Hg: int, layer: int) -> bool:
    """Forward gate projections as ONE pass on IEEE-f16 operands (EngineConfig.gi_f16, cruse_gemm_f16_nt): 11 significant bits on x
    and W_ih, where the split-bf16 form keeps x at 8 bits and spends a second (Hg <= 320: and a third) pass on low planes.  The
    producers' operand copies are then f16 tensors (g = 1; with g > 1 the interleaving LayerNorm has no fused copy and a cast pass
    makes it) and W_ih is K-tiled as f16.  The operands are BatchNorm + ReLU / LayerNorm outputs and |W_ih| ~ 1 / sqrt(Hg): inside
    f16's range; a value past 65504 would surface as a non-finite loss and the guarded optimizer step skips (engine.step_health).
    Not with the row-major TN weight gradients or the time-chunk pipeline, which read the bf16 copies.
    Only where the split-bf16 form does NOT split x as well (Hg > 320 by default, i.e. g = 1 at H = 640): the three-pass form of the
    grouped configurations carries ~16 bits on both operands and measured BETTER than f16 there (enhanced spectrum at T = 401, closed-form
    init, g = 4: 3.1e-4 against 1.7e-3; g = 2: 9e-5 against 5.5e-4), so those keep it."""
    c = config.get()
    return (bool((int(c.gi_f16) * 3 if isinstance(c.gi_f16, bool) else int(c.gi_f16 or 0)) >> layer & 1) and _bf16_gemm_path(prec, Hg) and not (_gi_x3_knob(Hg) & 4))


_GI16_PLAN: Dict[tuple, bool] = {}


def _gi16_served(B: int, Hg: int, prec: int) -> bool:
    """does the forward recurrence read f16 gi rows at this shape (cruse_gru_seq_fwd_gi16: chains of 8 on the tag-free lean kernel)?  The plan is a
    pure function of the shape (cached: two ctypes round trips per layer and step otherwise); the three library options that select another
    kernel are A/B switches of probes and tests, looked up only when one is set."""
    key = (B, Hg, prec)
    ok = _GI16_PLAN.get(key)
    if ok is None:
        ok = _GI16_PLAN[key] = ops.gru_plan(B, 1, Hg, prec)["clips_per_chain"] == 8
    if ok and config.get().lib_options:
        lo = config.get().lib_options
        ok = lo.get("gru_tf", 1) == 1 and lo.get("gru_wlo", 0) == 0 and lo.get("gru_fwd_lean", 1) == 1
    return ok


def _splitk(M: int, N: int, K: int) -> int:
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    sk = max(1, (512 + tiles - 1) // tiles)
    return max(1, min(sk, K // 256 if K >= 256 else 1))


# ======================================================================================
# GGRU functional core on [B,T,H] rows
# ======================================================================================
def ggru_forward(x: torch.Tensor, P: Dict[str, torch.Tensor], prefix: str, groups: int, prec,
                 residual: Optional[torch.Tensor] = None, save: bool = True, residual_ready=None, late_leaves=None,
                 x_bf16=None):
    """x [B,T,H] -> (ln2(gru2(ln1(interleave(gru1(x))))) [+ residual], ctx).  cruse_net.py:37-55."""
    return _ggru_forward_one(x, P, prefix, groups, prec, residual, save, residual_ready=residual_ready,
                             late_leaves=late_leaves, x_bf16=x_bf16)


def _ggru_forward_one(x, P, prefix, groups, prec, residual=None, save=True, out=None, slot=0, xcd_rot=0, pre_done=None,
                      residual_ready=None, late_leaves=None, x_bf16=None):
    """residual_ready(): called right before the residual is read (the last layer norm) -- the caller may still be
    producing it on a side stream while the recurrences run."""
    B, T, H = x.shape
    g = groups
    Hg = H // g
    rows = B * T
    ctx = dict(B=B, T=T, H=H, g=g, prec=prec, x=x, prefix=prefix, has_res=residual is not None, slot=slot, xcd_rot=xcd_rot)
    hooks = [pre_done] if pre_done is not None else []

    fast = _bf16_gemm_path(prec, Hg)


    def layer(inp, lname, inp_bf=None):
        # gi rows: f32, or IEEE f16 where the recurrence reads them (EngineConfig.gi_store_f16: bf16 mode, one group of 640, chains of 8 clips
        # -- cruse_gru_seq_fwd_gi16): the largest tensor of the forward pass, written once and read once -- 197 -> 98 MB per layer at the bench shape
        gi16 = bool(config.get().gi_store_f16) and fast and g == 1 and Hg == 640 and slot == 0 and _gi16_served(B, Hg, ops.prec_code(prec))
        gi = torch.empty(B, T, g * 3 * Hg, device=x.device, dtype=torch.float16 if gi16 else torch.float32)
        # The forward projection corrects the bf16 rounding of W_ih (a second pass with its low plane): that rounding
        # dominates the forward error of the bf16 mode (enhanced spectrum 1.25e-3 -> 5.1e-4 rel-L2 on fixture G6;
        # correcting x too only reaches 4.8e-4).  CRUSE_GI_X3: bit 0 / 1 = layer 1 / 2 corrected, bit 2 = also split x.
        # Default: W_ih split on both layers; x split as well when Hg <= 320 (K is then short enough that the third pass
        # costs < 0.04 ms per step, and the grouped configurations need it for the 1e-3 bar at T = 401: DESIGN.md section 2)
        knob = _gi_x3_knob(Hg)
        x3 = (knob >> (0 if lname == "gru_list1" else 1)) & 1
        split_x = bool(knob & 4) and x3
        pad = 64 if Hg % 64 else 0
        if _gi_f16(prec, Hg, 0 if lname == "gru_list1" else 1) and (inp_bf is None or inp_bf.dtype != torch.float16):
            # (eval, g > 1 or Hg % 64 != 0: no fused copy -- the same f16 operand from a cast pass, zero tail for the K round-up)
            inp_bf = torch.empty(rows * H + pad, device=x.device, dtype=torch.float16)
            if pad:
                inp_bf[rows * H:].zero_()
            ops.check(ops.lib.cruse_cast_f16(ops._p(inp), ops._p(inp_bf), rows * H, 1, ops._stream()))
        if inp_bf is not None:                   # written by the kernel that produced inp (_gi_takes_bf16_copy)
            inp_hi, inp_lo = inp_bf, None
        elif fast and split_x:
            inp_hi, inp_lo = ops.cast_bf16_padded(inp, pad=pad, split=True)
        else:
            inp_hi, inp_lo = (ops.cast_bf16_padded(inp, pad=pad) if fast else None), None
        kp = (Hg + 63) // 64 * 64
        b_ihs = [P[f"{prefix}{lname}.{i}.bias_ih_l0"] for i in range(g)]
        bstep = ops.uniform_stride(b_ihs)
        if (fast and g > 1 and inp_hi is not None and inp_hi.dtype == torch.bfloat16
                and bstep is not None):
            # all groups of the layer in ONE launch: the K-tiled W_ih planes stacked [g][kp / 64][3 Hg][64], columns of x / gi per group
            W_hi = torch.empty(g, kp // 64, 3 * Hg, 64, device=x.device, dtype=torch.bfloat16)
            W_lo = torch.empty_like(W_hi) if x3 else None
            for i in range(g):
                ops.ktile_bf16(P[f"{prefix}{lname}.{i}.weight_ih_l0"], 3 * Hg, Hg, split=bool(x3), out=(W_hi[i], W_lo[i] if x3 else None))
            ops.gemm_bf16_nt_groups(rows, 3 * Hg, kp, g, inp_hi, inp_lo if x3 else None, H, Hg, W_hi, W_lo, 64, W_hi[0].numel(), gi, 3 * H, 3 * Hg,
                                    bias=b_ihs[0], bias_gstep=bstep, b_kstride=3 * Hg * 64)
            grouped = True
        else:
            grouped = False
        for i in range(0 if not grouped else g, g):
            w_ih, b_ih = P[f"{prefix}{lname}.{i}.weight_ih_l0"], P[f"{prefix}{lname}.{i}.bias_ih_l0"]
            if inp_bf is not None and inp_bf.dtype == torch.float16 and x3 and lname == "gru_list1":   # layer 1 on f16 (gi_f16 bit 0) with its gi_x3 bit: f16 x against W_ih hi + lo planes
                w_hi, w_lo = ops.ktile_f16(w_ih, 3 * Hg, Hg, split=True)
                ops.gemm_f16_nt(rows, 3 * Hg, kp, inp_bf, i * Hg, H, w_hi, 0, 64, gi, i * 3 * Hg, 3 * H, bias=b_ih, b_kstride=3 * Hg * 64, B_lo=w_lo)
            elif inp_bf is not None and inp_bf.dtype == torch.float16:        # _gi_f16: one pass, 11-bit operands
                ops.gemm_f16_nt(rows, 3 * Hg, kp, inp_bf, i * Hg, H, ops.ktile_f16(w_ih, 3 * Hg, Hg), 0, 64, gi, i * 3 * Hg, 3 * H,
                                bias=b_ih, b_kstride=3 * Hg * 64)
            elif fast and x3:
                w_hi, w_lo = ops.ktile_bf16(w_ih, 3 * Hg, Hg, split=True)
                ops.gemm_bf16x3_nt(rows, 3 * Hg, kp, inp_hi, inp_lo, i * Hg, H, w_hi, w_lo, 0, 64, gi, i * 3 * Hg, 3 * H,
                                   bias=b_ih, b_kstride=3 * Hg * 64)
            elif fast:
                ops.gemm_bf16_nt(rows, 3 * Hg, kp, inp_hi, i * Hg, H, ops.ktile_bf16(w_ih, 3 * Hg, Hg), 0, 64, gi,
                                 i * 3 * Hg, 3 * H, bias=b_ih, b_kstride=3 * Hg * 64)
            else:
                ops.gemm(False, True, rows, 3 * Hg, Hg, inp, i * Hg, H, w_ih, 0, Hg, gi, i * 3 * Hg, 3 * H, bias=b_ih,
                         prec=prec)
        w_hh = [P[f"{prefix}{lname}.{i}.weight_hh_l0"] for i in range(g)]
        b_hh = [P[f"{prefix}{lname}.{i}.bias_hh_l0"] for i in range(g)]
        if hooks:
            hooks.pop()()                    # the pre-stage of this slice is issued: the next slice may start its own
        slot_, zeroed = STEP_SCRATCH.take(B, g, Hg, slot)
        return SIDE.release_around(lambda: ops.gru_seq_fwd(gi, w_hh, b_hh, B, T, g, Hg, prec, save=save, slot=slot_,
                                                           xcd_rot=xcd_rot, zeroed=zeroed))

    # The K-tiled time-major bf16 copies of x, h1, l1, h2 -- the K operands of the four weight-gradient GEMMs -- depend on
    # the forward pass only.  From the first backward recurrence on, the side streams' queue is what the optimizer step
    # waits for, while in the forward pass they idle: three of the copies are made beside the second forward recurrence,
    # the fourth beside the decoder (late_leaves: the caller issues it after it has joined the side streams).
    fwd_T = save and fast and SIDE.enabled and slot == 0
    tt = {}

    def queue_layer1_leaves(h1, l1, l1_bf):
        """queued for the launch of the second forward recurrence"""
        if not fwd_T:
            return
        ldT = (rows + 63) // 64 * 64
        tt["xT"], tt["h1T"], tt["l1T"], tt["h2T"] = (torch.empty(ldT // 64, H, 64, device=x.device, dtype=torch.bfloat16)
                                                     for _ in range(4))
        w_ts = {}                                # K-tiled W_ih^T of both layers: the B operand of the backward dX GEMMs

        def t_layer1(x=x, h1=h1, l1=l1):
            ops.transpose_bf16(x, rows, H, out=tt["xT"])
            ops.transpose_bf16(h1, rows, H, shift_T=T, out=tt["h1T"])
            ops.transpose_bf16(l1, rows, H, out=tt["l1T"])
            for lname in ("gru_list1", "gru_list2"):
                # (one stacked tensor per layer [g][ceil(3 Hg / 64)][Hg][64]: the grouped dX launch walks it with a group stride)
                stack = torch.empty(g, (3 * Hg + 63) // 64, Hg, 64, device=x.device, dtype=torch.bfloat16)
                for i in range(g):
                    w_ts[(lname, i)] = ops.transpose_bf16(P[f"{prefix}{lname}.{i}.weight_ih_l0"], 3 * Hg, Hg, out=stack[i])
                w_ts[(lname, "stack")] = stack
        ctx["w_ts"] = w_ts
        SIDE.defer(t_layer1, x, h1, l1, tt["xT"], tt["h1T"], tt["l1T"], kind=1, lane=2)

    h1, c1, a1, z1 = layer(x, "gru_list1", x_bf16)
    f16 = _gi_f16(prec, Hg, 1)
    l1_bf = (torch.empty(rows * H, device=x.device, dtype=torch.float16 if f16 else torch.bfloat16)
             if _gi_takes_bf16_copy(prec, Hg) and not (f16 and g > 1) else None)
    l1, m1, s1 = ops.ln_fwd(h1, P[prefix + "ln1.weight"], P[prefix + "ln1.bias"], None, rows, H, g, save=save, out_bf16=l1_bf)
    queue_layer1_leaves(h1, l1, l1_bf)
    h2, c2, a2, z2 = layer(l1, "gru_list2", l1_bf)
    if residual_ready is not None:
        residual_ready()
    out, m2, s2 = ops.ln_fwd(h2, P[prefix + "ln2.weight"], P[prefix + "ln2.bias"], residual, rows, H, 1, save=save, out=out)
    if fwd_T:
        h2T = tt["h2T"]

        def t_h2(h2=h2):
            ops.transpose_bf16(h2, rows, H, shift_T=T, out=h2T)
        if late_leaves is not None:
            late_leaves.append((t_h2, (h2, h2T)))
        else:
            SIDE.defer(t_h2, h2, h2T, kind=1, lane=2)
        ctx.update(T1=(tt["xT"], tt["h1T"]), T2=(tt["l1T"], h2T))
    if save:
        ctx.update(h1=h1, c1=c1, a1=a1, z1=z1, l1=l1, m1=m1, s1=s1,
                   h2=h2, c2=c2, a2=a2, z2=z2, m2=m2, s2=s2)
    return out, ctx


def ggru_backward(ctx, dout: torch.Tensor, P: Dict[str, torch.Tensor], G: Dict[str, torch.Tensor],
                  need_dx: bool = True, join: bool = True, dx_init: Optional[torch.Tensor] = None,
                  dx_ready=None, defer_last: bool = False) -> Optional[torch.Tensor]:
    """dout [B,T,H] -> dx; parameter gradients are ACCUMULATED into G[name].  dx_init: a [B,T,H] tensor the input
    gradient is ADDED to (and returned) instead of a fresh one; dx_ready() is called right before it is touched.
    defer_last: queue layer 1's weight-gradient leaf (SIDE.defer) instead of issuing it -- the caller ends a segment
    right after this function and issues it with SIDE.flush() at the start of the next one."""
    B, T, H, g, prec, prefix = ctx["B"], ctx["T"], ctx["H"], ctx["g"], ctx["prec"], ctx["prefix"]
    dx = None
    if need_dx:
        if dx_init is not None:
            dx = dx_init.view(B, T, H)
        else:
            dx = torch.empty(B, T, H, device=dout.device, dtype=torch.float32)
    _ggru_backward_one(ctx, dout, P, G, need_dx, dx, dx_init is not None, dx_ready, defer_last)
    if join:
        SIDE.join()
    return dx


def _ggru_backward_one(ctx, dout, P, G, need_dx, dx, dx_accum, dx_ready, defer_last, pre_done=None):
    """One batch slice of ggru_backward on the current stream; dx: the [B,T,H] rows its input gradient is written to
    (added to when dx_accum)."""
    B, T, H, g, prec, prefix = ctx["B"], ctx["T"], ctx["H"], ctx["g"], ctx["prec"], ctx["prefix"]
    slot, xcd_rot = ctx.get("slot", 0), ctx.get("xcd_rot", 0)
    Hg = H // g
    rows = B * T
    hooks = [pre_done] if pre_done is not None else []

    def run_bwd(dout_h, w_hh, coef, z, an=None, dg_slabs=3):
        if hooks:
            hooks.pop()()
        slot_, zeroed = STEP_SCRATCH.take(B, g, Hg, slot)
        return SIDE.release_around(lambda: ops.gru_seq_bwd(dout_h, w_hh, coef, z, B, T, g, Hg, prec, slot=slot_,
                                                           xcd_rot=xcd_rot, an=an, want_dgi=an is not None, dg_slabs=dg_slabs,
                                                           zeroed=zeroed))

    def dinp_buffer(dout_h, need_dinp, last):
        if not need_dinp:
            return None, False
        if last:
            if dx_accum and dx_ready is not None:
                dx_ready()
            return dx, dx_accum
        return torch.empty(B, T, H, device=dout_h.device, dtype=torch.float32), False

    def layer_bwd_bf16(dout_h, lname, inp, h, coef, an, z, need_dinp, last):
        """CRUSE_PREC_BF16: every product as gemm_bf16_nt on bf16 operand copies (see gemm_bf16.hip)."""
        names = [f"{prefix}{lname}.{i}." for i in range(g)]
        w_hh = [P[nm + "weight_hh_l0"] for nm in names]
        bias_ih = [G[nm + "bias_ih_l0"] for nm in names]
        bias_hh = [G[nm + "bias_hh_l0"] for nm in names]
        ldT = (rows + 63) // 64 * 64
        dh = run_bwd(dout_h, w_hh, coef, z)
        # EngineConfig.dx_atr: dX reads the time-major tensor dgT through transposing LDS reads (cruse_gemm_bf16_nt_atr) -- no row-major dgi is written
        dx_atr = bool(config.get().dx_atr) and Hg % 64 == 0 and need_dinp and g == 1
        dgi, dgT, ldT = ops.gru_gate_grads_bf16(dh, coef, an, rows, g, Hg, bias_ih, bias_hh, want_dgi=not dx_atr)
        early = early_T.pop(lname, None)         # layer 1: transposed by a leaf of the FIRST recurrence (see below)
        if early is not None:
            inpT, hpT = early
        else:
            inpT = torch.empty(ldT // 64, H, 64, device=dh.device, dtype=torch.bfloat16)
            hpT = torch.empty(ldT // 64, H, 64, device=dh.device, dtype=torch.bfloat16)

        def weight_grads():                      # leaves: overlap with the next recurrence / encoder backward
            if early is None:
                ops.transpose_bf16(inp, rows, H, out=inpT)
                ops.transpose_bf16(h, rows, H, shift_T=T, out=hpT)
            ka, kb = 4 * H * 64, H * 64          # k-tile strides of the K-tiled time-major operands (lda = ldb = 64)
            for i, nm in enumerate(names):
                # dW_ih += (r, z, n_i)^T x ; dW_hh += (r, z)^T h_{t-1} and n_h^T h_{t-1}
                a0, b0 = 4 * i * Hg * 64, i * Hg * 64
                g_ih, g_hh = G[nm + "weight_ih_l0"], G[nm + "weight_hh_l0"]
                sk = _splitk_bf16(6 * Hg, Hg, ldT)       # (k-slices for the 6 Hg x Hg concatenated output)
                if (abs(sk) > 1
                        and g_hh.data_ptr() == g_ih.data_ptr() + 4 * 3 * Hg * Hg):
                    # the three products as ONE launch on the concatenated output [dW_ih ; dW_hh] (back to back in the flat gradient buffer)
                    ops.gemm_bf16_nt_cat([3 * Hg, 2 * Hg, Hg], Hg, ldT, dgT, [4 * i * Hg, 4 * i * Hg, 4 * i * Hg + 3 * Hg], 64,
                                         [inpT, hpT, hpT], b0, 64, g_ih, 0, Hg, sk, a_kstride=ka, b_kstride=kb)
                    continue
                ops.gemm_bf16_nt(3 * Hg, Hg, ldT, dgT, a0, 64, inpT, b0, 64, G[nm + "weight_ih_l0"], 0, Hg,
                                 accumulate=True, splitk=_splitk_bf16(3 * Hg, Hg, ldT), slabs=True, a_kstride=ka, b_kstride=kb)
                ops.gemm_bf16_nt(2 * Hg, Hg, ldT, dgT, a0, 64, hpT, b0, 64, G[nm + "weight_hh_l0"], 0, Hg,
                                 accumulate=True, splitk=_splitk_bf16(2 * Hg, Hg, ldT), slabs=True, a_kstride=ka, b_kstride=kb)
                ops.gemm_bf16_nt(Hg, Hg, ldT, dgT, a0 + 3 * Hg * 64, 64, hpT, b0, 64, G[nm + "weight_hh_l0"],
                                 2 * Hg * Hg, Hg, accumulate=True, splitk=_splitk_bf16(Hg, Hg, ldT), slabs=True, a_kstride=ka,
                                 b_kstride=kb)

        # The last layer's weight-gradient leaf goes to the side stream BEFORE the dX GEMM is issued: its event then follows the
        # gate-gradient pass, not the GEMM, and the three dW products start ~150 us earlier (5.40 vs 5.46 ms).
        early_leaf = last and not defer_last
        if early_leaf:
            SIDE.run(weight_grads, dgT, h, inp, inpT, hpT, dh, lane=2)
        dinp, acc_dx = dinp_buffer(dout_h, need_dinp, last)
        stack = ctx.get("w_ts", {}).get((lname, "stack"))
        if need_dinp and g > 1 and stack is not None:
            # all groups in ONE launch (cruse_gemm_bf16_nt_groups): columns [q * 3 Hg, ...) of dgi against W_ih,q^T into columns [q * Hg, ...) of dX
            ops.gemm_bf16_nt_groups(rows, Hg, stack.shape[1] * 64, g, dgi, None, 3 * H, 3 * Hg, stack, None, 64, stack[0].numel(), dinp, H, Hg,
                                    accumulate=acc_dx, b_kstride=Hg * 64)
        elif need_dinp:
            for i, nm in enumerate(names):
                w_t = ctx.get("w_ts", {}).get((lname, i))                             # made in the forward pass (side stream)
                if w_t is None:
                    w_t = ops.transpose_bf16(P[nm + "weight_ih_l0"], 3 * Hg, Hg)      # K-tiled [ceil(3*Hg/64), Hg, 64]
                if dgi is None:
                    ops.gemm_bf16_nt_atr(rows, Hg, 3 * Hg, dgT, 4 * i * Hg * 64, g * 4 * Hg * 64, ldT // 64, w_t, 0, 64, dinp, i * Hg, H,
                                         accumulate=acc_dx, b_kstride=Hg * 64)
                    continue
                ops.gemm_bf16_nt(rows, Hg, w_t.shape[0] * 64, dgi, i * 3 * Hg, 3 * H, w_t, 0, 64, dinp, i * Hg, H,
                                 accumulate=acc_dx, b_kstride=Hg * 64)
        if early_leaf:
            pass
        elif last and defer_last:
            SIDE.defer(weight_grads, dgT, h, inp, inpT, hpT, dh, kind=0xffff, lane=2)
        elif last:
            SIDE.run(weight_grads, dgT, h, inp, inpT, hpT, dh, lane=2)
        else:
            SIDE.defer(weight_grads, dgT, h, inp, inpT, hpT, dh, kind=4, lane=2)
        return dinp

    def layer_bwd(dout_h, lname, inp, h, coef, an, z, need_dinp, last):
        """last: no recurrence follows, so the weight-gradient leaves start at once instead of with the next one."""
        if _bf16_gemm_path(prec, Hg):
            return layer_bwd_bf16(dout_h, lname, inp, h, coef, an, z, need_dinp, last)
        w_hh = [P[f"{prefix}{lname}.{i}.weight_hh_l0"] for i in range(g)]
        dh = run_bwd(dout_h, w_hh, coef, z)
        dgi, dgh = ops.gru_gate_grads(dh, coef, an, rows, g, Hg, prec)
        sk = _splitk(3 * Hg, Hg, rows)

        def weight_grads():                      # leaves: overlap with the next recurrence / encoder backward
            for i in range(g):
                nm = f"{prefix}{lname}.{i}."
                # dW_hh += dgh^T h_{t-1}
                ops.gemm(True, False, 3 * Hg, Hg, rows, dgh, i * 3 * Hg, 3 * H, h, i * Hg, H, G[nm + "weight_hh_l0"], 0,
                         Hg, accumulate=True, splitk=sk, b_shift_T=T, prec=prec)
                ops.col_sum(dgh, i * 3 * Hg, rows, 3 * Hg, 3 * H, G[nm + "bias_hh_l0"])
                # dW_ih += dgi^T x
                ops.gemm(True, False, 3 * Hg, Hg, rows, dgi, i * 3 * Hg, 3 * H, inp, i * Hg, H, G[nm + "weight_ih_l0"], 0,
                         Hg, accumulate=True, splitk=sk, prec=prec)
                ops.col_sum(dgi, i * 3 * Hg, rows, 3 * Hg, 3 * H, G[nm + "bias_ih_l0"])

        dinp, acc_dx = dinp_buffer(dout_h, need_dinp, last)
        if need_dinp:
            for i in range(g):
                ops.gemm(False, False, rows, Hg, 3 * Hg, dgi, i * 3 * Hg, 3 * H,
                         P[f"{prefix}{lname}.{i}.weight_ih_l0"], 0, Hg, dinp, i * Hg, H, accumulate=acc_dx, prec=prec)
        if last and defer_last:
            SIDE.defer(weight_grads, dgi, dgh, h, inp, kind=0xffff, lane=2)
        elif last:
            SIDE.run(weight_grads, dgi, dgh, h, inp, lane=2)
        else:
            SIDE.defer(weight_grads, dgi, dgh, h, inp, kind=4, lane=2)
        return dinp

    # The time-major bf16 copies of layer 1's x and h_{t-1} (the K operands of its weight-gradient GEMMs) depend on the
    # forward pass only: they are made by a leaf of the FIRST backward recurrence, whose window the side streams do not
    # fill -- not after the second one, where the side streams' queue (layer 1's GEMMs, the encoder's weight gradients)
    # is what the optimizer step waits for.
    early_T = {}
    if "T1" in ctx:                                      # made in the forward pass (_ggru_forward_one)
        early_T["gru_list1"], early_T["gru_list2"] = ctx["T1"], ctx["T2"]
    elif _bf16_gemm_path(prec, Hg) and SIDE.enabled:
        ldT1 = (rows + 63) // 64 * 64
        x1T = torch.empty(ldT1 // 64, H, 64, device=dout.device, dtype=torch.bfloat16)
        h1T = torch.empty(ldT1 // 64, H, 64, device=dout.device, dtype=torch.bfloat16)
        early_T["gru_list1"] = (x1T, h1T)

        def early_transposes(x1=ctx["x"], h1=ctx["h1"]):
            ops.transpose_bf16(x1, rows, H, out=x1T)
            ops.transpose_bf16(h1, rows, H, shift_T=T, out=h1T)
        SIDE.defer(early_transposes, x1T, h1T, kind=4, lane=2)
    dh2 = ops.ln_bwd(dout, ctx["h2"], ctx["m2"], ctx["s2"], P[prefix + "ln2.weight"], rows, H, 1,
                     G[prefix + "ln2.weight"], G[prefix + "ln2.bias"])
    dl1 = layer_bwd(dh2, "gru_list2", ctx["l1"], ctx["h2"], ctx["c2"], ctx["a2"], ctx["z2"], True, False)
    dh1 = ops.ln_bwd(dl1, ctx["h1"], ctx["m1"], ctx["s1"], P[prefix + "ln1.weight"], rows, H, g,
                     G[prefix + "ln1.weight"], G[prefix + "ln1.bias"])
    layer_bwd(dh1, "gru_list1", ctx["x"], ctx["h1"], ctx["c1"], ctx["a1"], ctx["z1"], need_dx, True)


# ======================================================================================
# unet_2 functional core
# ======================================================================================
BN_EPS = 1e-5
BN_MOMENTUM = 0.1


_PENDING_COUNTERS = []      # num_batches_tracked buffers of this forward: bumped by ONE kernel at its end, not seven


def _flush_counters():
    if _PENDING_COUNTERS:
        ops.counters_add(list(_PENDING_COUNTERS), 1)
        _PENDING_COUNTERS.clear()


def _bn_act(y, rows, C, F, P, Bf, name, training, update_running, skip=None, sums=None, out_bf16=None):
    """BatchNorm2d (train: batch statistics, running stats updated; eval: running stats) + ReLU (+ skip) -> (out, mean, rstd).
    sums: the batch sums of y when the conv that produced it has already accumulated them (ops.conv_*_bnstats)."""
    gamma, beta = P[name + ".weight"], P[name + ".bias"]
    if training:
        if sums is None:
            sums = ops.bn_stats(y, rows, C, F)
        rm = Bf[name + ".running_mean"] if update_running else None
        rv = Bf[name + ".running_var"] if update_running else None
        if update_running:
            _PENDING_COUNTERS.append(Bf[name + ".num_batches_tracked"])
        return ops.bn_finalize_act_fwd(y, sums, rows * F, BN_EPS, BN_MOMENTUM, gamma, beta, skip, rows, C, F, relu=True,
                                       running_mean=rm, running_var=rv, out_bf16=out_bf16)
    mean, rstd = ops.bn_eval_stats(Bf[name + ".running_mean"], Bf[name + ".running_var"], BN_EPS)
    return ops.bn_act_fwd(y, mean, rstd, gamma, beta, skip, rows, C, F, relu=True), mean, rstd


def unet2_forward(x: torch.Tensor, P: Dict[str, torch.Tensor], Bf: Dict[str, torch.Tensor], ch, groups: int,
                  prec, training: bool, save: bool = True, update_running: bool = True, dec_mode: str = "transposed"):
    """x [B,1,T,F0] (== frame-major [B,T,1,F0]) -> (mask [B,1,T,F0], ctx).  cruse_net.py:147-165.

    dec_mode "upsample": the decoder of model/cruse.py:14 (CRUSE4MagAddSkipUpsample) -- cust_conv.convkxf(mode="upsample"),
    :159-167: nearest FreqUpsample(2) + Conv2d (1,3) pad (0,1) with weights conv{k}_t.weight [Cout][Cin][1][3] (no bias under a
    BatchNorm) -- on the same frame-major kernels: the upsampled tensor is materialised once (cruse_upsample_w), kept for the
    weight gradient, and the conv is the gather form at stride 1."""
    if dec_mode not in ("transposed", "upsample"):
        raise RuntimeError(f"unet2_forward: unknown dec_mode {dec_mode!r}")
    ups = dec_mode == "upsample"
    if x.dim() != 4 or x.shape[1] != ch[0]:
        raise RuntimeError(f"unet_2 expects [B,{ch[0]},T,F], got {tuple(x.shape)}")
    if ch[0] != 1:
        raise RuntimeError("cruse_amd unet_2 supports ch[0] == 1 (magnitude input) only")
    B, _, T, F0 = x.shape
    L = len(ch) - 1
    if F0 % (1 << L) != 0:
        raise RuntimeError(f"unet_2: {F0} input bins are not divisible by 2**{L} (the network runs on in_feat//2*2 bins)")
    rows = B * T
    Fk = [F0 >> k for k in range(L + 1)]
    ctx = dict(B=B, T=T, F=Fk, ch=tuple(ch), L=L, prec=prec, training=training, x=x)
    cur = x
    _PENDING_COUNTERS.clear()
    ys, es, ss, stats = [None], [x], [None], [None]
    # EngineConfig.fuse_bn_fwd (bf16 mode, training): BatchNorm-apply + ReLU (+ skip add) of a level run inside the staging of the
    # convs that consume it -- `pend` describes such a VIRTUAL tensor (ops.BnIn); es[k] / us[k] are then the bf16 copies the
    # consuming conv writes for the weight gradients, mean / rstd are published by that conv's block 0
    # (the MFMA convs stage whole frame rows: up to 640 elements per row and level -- wider networks run the VALU kernels, unfused)
    fz = (training and save and config.get().fuse_bn_fwd and config.get().fuse_bn_stats
          and all(ch[k] * Fk[k] <= 640 for k in range(1, L + 1)))
    dev = x.device

    def virtual(y_, sums_, C, F, name, add=None):
        mean_ = torch.empty(C, device=dev, dtype=torch.float32); rstd_ = torch.empty(C, device=dev, dtype=torch.float32)
        rm = Bf[name + ".running_mean"] if update_running else None
        rv = Bf[name + ".running_var"] if update_running else None
        if update_running:
            _PENDING_COUNTERS.append(Bf[name + ".num_batches_tracked"])
        return ops.BnIn(y_, sums_, ops.BN_STAT_REPLICAS, rows * F, BN_EPS, BN_MOMENTUM, P[name + ".weight"], P[name + ".bias"], mean_, rstd_,
                        rm, rv, add=add)
    pend = None
    e_bf = None
    for k in range(1, L + 1):
        if pend is not None:
            y, sums = ops.conv_gather_bnin(pend, P[f"conv{k}.weight"], P[f"conv{k}.bias"], B, T, ch[k - 1], Fk[k - 1], ch[k], Fk[k], KT=2, S=2,
                                           pad=1, prec=prec, publish=True, copy_bf16=es[k - 1], want_sums=True)
        elif training and config.get().fuse_bn_stats:
            y, sums = ops.conv_gather_bnstats(cur, P[f"conv{k}.weight"], P[f"conv{k}.bias"], B, T, ch[k - 1], Fk[k - 1], ch[k],
                                              Fk[k], KT=2, S=2, pad=1, prec=prec)
        else:
            y, sums = ops.conv_gather(cur, P[f"conv{k}.weight"], P[f"conv{k}.bias"], B, T, ch[k - 1], Fk[k - 1], ch[k], Fk[k],
                                      KT=2, S=2, pad=1, prec=prec), None
        s = torch.empty(B, T, ch[k], Fk[k], device=x.device, dtype=torch.float32)
        # e_k stays virtual when both of its forward consumers -- conv_{k+1} and skip_k -- take it fused (level L feeds the GGRU)
        if fz and k < L and ops.bnin_eligible(prec, ch[k], ch[k + 1]) and ops.bnin_eligible(prec, ch[k], ch[k]):
            pend = virtual(y, sums, ch[k], Fk[k], f"bn{k}")
            e = torch.empty(B, T, ch[k], Fk[k], device=dev, dtype=torch.bfloat16)        # written by conv_{k+1} while it stages
            mean, rstd = pend.mean, pend.rstd

            def skip_conv(bn=pend, s=s, k=k):
                ops.conv_gather_bnin(bn, P[f"skip_connect_{k}.weight"], None, B, T, ch[k], Fk[k], ch[k], Fk[k], KT=1, S=1, pad=1,
                                     prec=prec, out=s)
            SIDE.defer(skip_conv, y, s, kind=1, lane=1)
        else:
            pend = None
            e_bf = None
            if k == L and training and _gi_takes_bf16_copy(prec, ch[L] * Fk[L] // groups):
                e_bf = torch.empty(rows * ch[L] * Fk[L], device=x.device,                          # gate GEMM 1's operand
                                   dtype=torch.float16 if _gi_f16(prec, ch[L] * Fk[L] // groups, 0) else torch.bfloat16)
            e, mean, rstd = _bn_act(y, rows, ch[k], Fk[k], P, Bf, f"bn{k}", training, update_running, sums=sums, out_bf16=e_bf)

            def skip_conv(e=e, s=s, k=k):
                ops.conv_gather(e, P[f"skip_connect_{k}.weight"], None, B, T, ch[k], Fk[k], ch[k], Fk[k], KT=1, S=1, pad=1,
                                out=s, prec=prec)
            # needed by the decoder only (level L: by the layer norm that closes the GGRU block): issued with the GRU forward
            SIDE.defer(skip_conv, e, s, kind=1, lane=1)
        ys.append(y); es.append(e); ss.append(s); stats.append((mean, rstd))
        cur = e
    H = ch[L] * Fk[L]
    late = []
    u, gctx = ggru_forward(cur.view(B, T, H), P, "gru.", groups, prec, residual=ss[L].view(B, T, H), save=save,
                           residual_ready=SIDE.join, late_leaves=late, x_bf16=e_bf)
    u = u.view(B, T, ch[L], Fk[L])
    SIDE.join()
    for fn, keep in late:                               # beside the decoder: nothing on the main stream waits for these
        SIDE.run(fn, *keep, lane=2)
    us, vs, dstats, uus = {L: u}, {}, {}, {}
    pend = None
    for k in range(L, 1, -1):
        if ups:
            uus[k] = ops.upsample_w(u, rows * ch[k], Fk[k], 2).view(B, T, ch[k], Fk[k - 1])
            if training and config.get().fuse_bn_stats:
                v, sums = ops.conv_gather_bnstats(uus[k], P[f"conv{k}_t.weight"], P.get(f"conv{k}_t.bias"), B, T, ch[k], Fk[k - 1], ch[k - 1],
                                                  Fk[k - 1], KT=1, S=1, pad=1, prec=prec)
            else:
                v, sums = ops.conv_gather(uus[k], P[f"conv{k}_t.weight"], P.get(f"conv{k}_t.bias"), B, T, ch[k], Fk[k - 1], ch[k - 1],
                                          Fk[k - 1], KT=1, S=1, pad=1, prec=prec), None
        elif pend is not None:                            # u_k = relu(bn(v_{k+1})) + skip_k applied while conv{k}_t stages v_{k+1} and skip_k
            v, sums = ops.conv_scatter2_bnin(pend, P[f"conv{k}_t.weight"], P[f"conv{k}_t.bias"], B, T, ch[k], Fk[k], ch[k - 1], KT=1, pad=0,
                                             prec=prec, publish=True, copy_bf16=us[k], want_sums=True)
        elif training and config.get().fuse_bn_stats:
            v, sums = ops.conv_scatter2_bnstats(u, P[f"conv{k}_t.weight"], P[f"conv{k}_t.bias"], B, T, ch[k], Fk[k], ch[k - 1],
                                                KT=1, pad=0, prec=prec)
        else:
            v, sums = ops.conv_scatter2(u, P[f"conv{k}_t.weight"], P[f"conv{k}_t.bias"], B, T, ch[k], Fk[k], ch[k - 1], KT=1,
                                        pad=0, prec=prec), None
        # u_{k-1} stays virtual when its consumer conv{k-1}_t takes it fused (the last decoder layer, Cout = 1, is a VALU kernel)
        if fz and not ups and k - 1 >= 2 and ops.bnin_eligible(prec, ch[k - 1], ch[k - 2]):
            pend = virtual(v, sums, ch[k - 1], Fk[k - 1], f"bn{k}_t", add=ss[k - 1])
            u = torch.empty(B, T, ch[k - 1], Fk[k - 1], device=dev, dtype=torch.bfloat16)      # written by conv{k-1}_t while it stages
            mean, rstd = pend.mean, pend.rstd
        else:
            pend = None
            u, mean, rstd = _bn_act(v, rows, ch[k - 1], Fk[k - 1], P, Bf, f"bn{k}_t", training, update_running, skip=ss[k - 1],
                                    sums=sums)
        vs[k] = v; dstats[k] = (mean, rstd); us[k - 1] = u
    if ups:
        uus[1] = ops.upsample_w(u, rows * ch[1], Fk[1], 2).view(B, T, ch[1], Fk[0])
        mask = ops.conv_gather(uus[1], P["conv1_t.weight"], P.get("conv1_t.bias"), B, T, ch[1], Fk[0], ch[0], Fk[0], KT=1, S=1, pad=1,
                               act=1, prec=prec)
    else:
        mask = ops.conv_scatter2(u, P["conv1_t.weight"], P["conv1_t.bias"], B, T, ch[1], Fk[1], ch[0], KT=1, pad=0, act=1,
                                 prec=prec)
    _flush_counters()
    if save:
        ctx.update(ys=ys, es=es, stats=stats, gctx=gctx, us=us, vs=vs, dstats=dstats, mask=mask, uus=uus, dec_mode=dec_mode)
        if e_bf is not None and e_bf.dtype == torch.bfloat16:
            ctx["e_bf"] = e_bf.view(B, T, ch[L], Fk[L])        # RNE(e_L): the operand bits the bf16 weight gradient forms from e_L anyway
    return mask.view(B, ch[0], T, F0), ctx


def bucket_of(name: str) -> int:
    """Gradient bucket of a unet_2 parameter, in the order the backward pass FINISHES them:
    0 = decoder convT / BN, skip convs and GGRU layer 2 (+ ln2) -- final when the layer-1 recurrence has run;
    1 = GGRU layer 1 (+ ln1) -- its dW GEMMs run beside the first half of the encoder backward;
    2 = encoder convs / BN."""
    if name.startswith("gru.gru_list1.") or name.startswith("gru.ln1."):
        return 1
    if name.startswith("gru.") or name.startswith("skip_connect_") or "_t." in name:
        return 0
    return 2


N_BUCKETS = 3


def unet2_backward(ctx, dlogit: torch.Tensor, P: Dict[str, torch.Tensor], G: Dict[str, torch.Tensor],
                   boundary=None, need_dx: bool = False):
    """dlogit = dL/d(pre-sigmoid) [B,T,1,F0]; parameter gradients are ACCUMULATED into G[name].  need_dx: also return the
    gradient wrt the input magnitude [B,T,1,F0] (the nn.Module surface; the training step never asks for it).

    boundary(b): called at the two points where gradient bucket b (bucket_of) has just become final once the issued
    side-stream leaves are joined -- after the GGRU backward (b = 0) and half way down the encoder (b = 1).  The
    callback must SIDE.join(flush=False), may end a graph capture / launch the bucket's all-reduce, and must
    SIDE.flush() before returning.  Bucket 2 is final when this function returns."""
    B, T, Fk, ch, L, training = ctx["B"], ctx["T"], ctx["F"], ctx["ch"], ctx["L"], ctx["training"]
    prec = ctx["prec"]
    # data-gradient convolutions of the bf16 mode: plain bf16 operands (one MFMA, one conversion per element) like every
    # other backward contraction of that mode -- the split-bf16 x3 form is what the FORWARD convs need for the 1e-3 bar
    dprec = prec
    if ops.prec_code(prec) == ops.PREC_BF16:
        dprec = ops.PREC_BF16
    rows = B * T
    ys, es, stats, us, vs, dstats = ctx["ys"], ctx["es"], ctx["stats"], ctx["us"], ctx["vs"], ctx["dstats"]
    _INLINE = config.get().inline_mask
    # backward-only tensors in bf16 (EngineConfig.bf16_dy): only where every consumer rounds them to bf16 operands anyway
    # ... and every consumer is an MFMA kernel (the VALU fallbacks of the convs and weight gradients take f32 only: channel counts
    # beyond 64 or not a power of two keep the f32 tensors -- ADVICE r4)
    all_mfma = all(8 <= c <= 64 and (c & (c - 1)) == 0 for c in ch[1:]) and all(ch[k] * Fk[k] <= 640 for k in range(1, L + 1))
    dy_bf16 = bool(config.get().bf16_dy) and ops.prec_code(prec) == ops.PREC_BF16 and dprec == ops.PREC_BF16 and all_mfma
    ups = ctx.get("dec_mode", "transposed") == "upsample"
    uus = ctx.get("uus", {})

    def dec_wgrad(k, dv_):
        """weight gradient of decoder conv k"""
        if ups and ch[k - 1] % 4 != 0:
            # (one output channel: the kernel wants the narrow tensor in the second role -- sum uu[ci, f] dv[co, f - 1 + kf] is the
            # same sum with the taps mirrored)
            tmp = torch.zeros(ch[k], ch[k - 1], 1, 3, device=dlogit.device, dtype=torch.float32)
            ops.conv_wgrad(uus[k], dv_, tmp, B, T, ch[k], Fk[k - 1], ch[k - 1], Fk[k - 1], KT=1, S=1, pad=1, prec=prec)
            G[f"conv{k}_t.weight"].add_(tmp.flip(-1).permute(1, 0, 2, 3))
        elif ups:       # Conv2d (1,3) on the upsampled input: dW[co][ci][kf] = sum dv[co, f] uu[ci, f - 1 + kf]
            ops.conv_wgrad(dv_, uus[k], G[f"conv{k}_t.weight"], B, T, ch[k - 1], Fk[k - 1], ch[k], Fk[k - 1], KT=1, S=1, pad=1, prec=prec)
        else:
            ops.conv_wgrad(us[k], dv_, G[f"conv{k}_t.weight"], B, T, ch[k], Fk[k], ch[k - 1], Fk[k - 1], KT=1, S=2, pad=0, prec=prec)

    def dec_dgrad(k, dv_, bn_bwd, out_bf16=False):
        """gradient wrt u_k of decoder conv k -> (du_k, backward sums of the BatchNorm above or None)"""
        if ups:         # W^T dv on the upsampled grid, then the sum of each pair of bins (gradient of the nearest upsample)
            duu = ops.conv_gather(dv_, P[f"conv{k}_t.weight"], None, B, T, ch[k - 1], Fk[k - 1], ch[k], Fk[k - 1], KT=1, S=1, pad=1,
                                  w_layout=1, prec=dprec)
            return ops.downsum_w(duu, rows * ch[k], Fk[k], 2).view(B, T, ch[k], Fk[k]), None
        return split(ops.conv_gather(dv_, P[f"conv{k}_t.weight"], None, B, T, ch[k - 1], Fk[k - 1], ch[k], Fk[k], KT=1, S=2, pad=0,
                                     prec=dprec, bn_bwd=bn_bwd, out_bf16=out_bf16))
    # ---- decoder level 1: v1 = convT_1(u1) -------------------------------------------
    dv = dlogit

    def leaf_dec1(dv=dv):
        if "conv1_t.bias" in G:
            ops.channel_sum(dv, rows, ch[0], Fk[0], G["conv1_t.bias"])
        dec_wgrad(1, dv)
    SIDE.defer(leaf_dec1, dv, kind=2, lane=0)                           # decoder leaves: issued with the first GRU backward
    # A data gradient that feeds a BatchNorm backward accumulates that BatchNorm's backward sums (sum g, sum g*xhat per channel)
    # in its own epilogue (cruse_conv_*_bnbwd): the reduce pass over (du, v) / (de, y) -- 132 MB and ~47 us per level -- is only
    # left for the level the GGRU feeds.  EngineConfig.fuse_bn_bwd_stats.
    fuse_bwd = config.get().fuse_bn_bwd_stats

    def bn_of(k, dec):
        if not fuse_bwd:
            return None
        mean_, rstd_ = dstats[k] if dec else stats[k]
        nm = f"bn{k}_t" if dec else f"bn{k}"
        return ((vs[k] if dec else ys[k]), mean_, rstd_, P[nm + ".weight"], P[nm + ".bias"], True)

    def split(r):
        return r if isinstance(r, tuple) else (r, None)
    # ... and the data gradients of the levels between two MFMA kernels (EngineConfig.bf16_de): du_k / de_k for 2 <= k < L
    de_bf16 = dy_bf16 and bool(config.get().bf16_de) and fuse_bwd and not ups

    def lvl_bf16(k):
        return de_bf16 and 2 <= k < L
    du, du_sums = dec_dgrad(1, dv, bn_of(2, True) if L >= 2 else None)
    ds = {1: du}                                        # gradient wrt skip_{k} output = du_k
    # skip_k = conv1x3(e_k) is a leaf of the decoder: its data gradient W^T ds_k and its weight gradient ds_k (*) e_k
    # are issued here, on the side stream, into the buffer de_pre[k] that the encoder backward later ACCUMULATES its
    # own path into -- four convs and four weight gradients less on the serial tail of the backward pass
    de_pre = {}

    def skip_leaves(k):
        de_pre[k] = torch.empty(B, T, ch[k], Fk[k], device=dlogit.device, dtype=torch.bfloat16 if lvl_bf16(k) else torch.float32)

        def dgrad(k=k, dsk=ds[k], out=de_pre[k]):
            ops.conv_gather(dsk, P[f"skip_connect_{k}.weight"], None, B, T, ch[k], Fk[k], ch[k], Fk[k], KT=1, S=1, pad=1,
                            w_layout=1, out=out, prec=dprec)

        def wgrad(k=k, dsk=ds[k]):
            ek = es[k]
            if k == L and "e_bf" in ctx and ops.prec_code(prec) == ops.PREC_BF16 and all_mfma:
                ek = ctx["e_bf"]                            # (the gate GEMM's bf16 copy of e_L: half the bytes, the same operand bits)
            ops.conv_wgrad(dsk, ek, G[f"skip_connect_{k}.weight"], B, T, ch[k], Fk[k], ch[k], Fk[k], KT=1, S=1, pad=1,
                           prec=prec)
        # Deferred to the first backward recurrence (issuing them here costs an event record on the main stream per level
        # and makes the decoder's BatchNorm backward share HBM with them).  Since the recurrences got shorter the side
        # queue -- not the main stream -- is what the optimizer step waits for, and a leaf beside a recurrence takes 2-2.7x
        # its time alone (96 free CUs): _INLINE moves leaves back onto the main stream where that shortens the step
        # (bit 0 skip data gradients, bit 1 skip weight gradients, bit 2 decoder weight gradients).
        if _INLINE & 1:
            dgrad()
        else:
            SIDE.defer(dgrad, ds[k], de_pre[k], kind=8, lane=1)
        if _INLINE & 2:
            wgrad()
        else:
            SIDE.defer(wgrad, ds[k], kind=8, lane=1)
    skip_leaves(1)
    # ---- decoder levels 2..L ------------------------------------------------------------
    # EngineConfig.fuse_bn_bwd_apply: where the incoming gradient is a bf16 tensor that arrived with its sums, the BatchNorm-backward apply
    # pass runs inside the staging of the data-gradient conv (which also writes the bf16 dy for the weight gradient)
    fuse_apply = bool(config.get().fuse_bn_bwd_apply) and dy_bf16 and fuse_bwd and not ups
    for k in range(2, L + 1):
        mean, rstd = dstats[k]
        fused = fuse_apply and du.dtype == torch.bfloat16 and du_sums is not None
        if fused:
            du, du_sums, dv = ops.conv_gather_bwd_in(
                du, (vs[k], mean, rstd, P[f"bn{k}_t.weight"], P[f"bn{k}_t.bias"], du_sums, True, training, G[f"bn{k}_t.weight"],
                     G[f"bn{k}_t.bias"], G.get(f"conv{k}_t.bias")),
                P[f"conv{k}_t.weight"], B, T, ch[k - 1], Fk[k - 1], ch[k], Fk[k], KT=1, S=2, pad=0, prec=dprec,
                bn_bwd=bn_of(k + 1, True) if k < L else None, out_bf16=lvl_bf16(k))
        else:
            dv = ops.bn_act_bwd(du, vs[k], mean, rstd, P[f"bn{k}_t.weight"], P[f"bn{k}_t.bias"], rows, ch[k - 1],
                                Fk[k - 1], True, training, G[f"bn{k}_t.weight"], G[f"bn{k}_t.bias"],
                                dbias=G.get(f"conv{k}_t.bias"), sums=du_sums, out_bf16=dy_bf16)

        def leaf_dec(dv=dv, k=k):
            dec_wgrad(k, dv)
        if _INLINE & 4:
            leaf_dec()
        else:
            SIDE.defer(leaf_dec, dv, kind=2, lane=0)
        if not fused:
            du, du_sums = dec_dgrad(k, dv, bn_of(k + 1, True) if k < L else None, out_bf16=lvl_bf16(k))
        ds[k] = du
        skip_leaves(k)
    # ---- bottleneck: u_L = ggru(e_L) + skip_L --------------------------------------------
    H = ch[L] * Fk[L]
    # de_pre[L] is written by a skip leaf, issued with the first backward recurrence: the GGRU waits, right before it adds
    # into de_pre[L], for what the side streams had been given by THAT launch -- not for the layer-2 weight-gradient GEMMs
    # queued behind it later (waiting for those stalled the main stream for 0.27 ms once the recurrence got shorter)
    marks = {}
    SIDE.after_release = lambda: marks.__setitem__("ev", SIDE.mark())

    def de_pre_ready():
        SIDE.after_release = None
        SIDE.wait(marks["ev"] if "ev" in marks else SIDE.mark())
    de = ggru_backward(ctx["gctx"], du.view(B, T, H), P, G, join=False, dx_init=de_pre[L].view(B, T, H),
                       dx_ready=de_pre_ready, defer_last=boundary is not None).view(B, T, ch[L], Fk[L])
    if boundary is not None:
        boundary(0)
    cut = max(L // 2, 1)                          # levels L..cut+1, [bucket 1 final], levels cut..1
    # ---- encoder levels L..1: de_k already holds the skip path ------------------------------
    de_sums = None
    for k in range(L, 0, -1):
        mean, rstd = stats[k]
        fused = fuse_apply and k > 1 and de.dtype == torch.bfloat16 and de_sums is not None
        if fused:
            de_new, de_sums_new, dy = ops.conv_scatter2_bwd_in(
                de, (ys[k], mean, rstd, P[f"bn{k}.weight"], P[f"bn{k}.bias"], de_sums, True, training, G[f"bn{k}.weight"], G[f"bn{k}.bias"],
                     G[f"conv{k}.bias"]),
                P[f"conv{k}.weight"], B, T, ch[k], Fk[k], ch[k - 1], KT=2, pad=1, out=de_pre[k - 1], accum=True, prec=dprec,
                bn_bwd=bn_of(k - 1, False))
        else:
            # (level 1 with need_dx: its data gradient into the one-channel input is a VALU conv -- f32 dy)
            dy = ops.bn_act_bwd(de, ys[k], mean, rstd, P[f"bn{k}.weight"], P[f"bn{k}.bias"], rows, ch[k], Fk[k], True,
                                training, G[f"bn{k}.weight"], G[f"bn{k}.bias"], dbias=G[f"conv{k}.bias"], sums=de_sums,
                                out_bf16=dy_bf16 and not (need_dx and k == 1))

        def leaf_enc(dy=dy, k=k):
            ops.conv_wgrad(dy, es[k - 1], G[f"conv{k}.weight"], B, T, ch[k], Fk[k], ch[k - 1], Fk[k - 1], KT=2, S=2, pad=1, prec=prec)
        # the optimizer step waits for the side queue, not for the main stream: the LAST levels' weight gradients run on
        # the main stream itself (_INLINE bits 3, 4: level 1, level 2), beside what is still queued on the side
        if _INLINE & (8 << (k - 1)):                       # bits 3.. = levels 1..
            leaf_enc()
        else:
            SIDE.run(leaf_enc, dy, lane=0)
        if fused:
            de, de_sums = de_new, de_sums_new
        elif k > 1:
            de, de_sums = split(ops.conv_scatter2(dy, P[f"conv{k}.weight"], None, B, T, ch[k], Fk[k], ch[k - 1], KT=2, pad=1,
                                                  out=de_pre[k - 1], accum=True, prec=dprec, bn_bwd=bn_of(k - 1, False)))
        if boundary is not None and k == cut + 1:
            boundary(1)
    dx = None
    if need_dx:
        dx = ops.conv_scatter2(dy, P["conv1.weight"], None, B, T, ch[1], Fk[1], ch[0], KT=2, pad=1, prec=dprec)
    SIDE.join()
    return dx


# ======================================================================================
# autograd glue + nn.Module surface
# ======================================================================================
class _GGRUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod, names, *params):
        P = dict(zip(names, params))
        need = any(p.requires_grad for p in params) or x.requires_grad
        out, c = ggru_forward(x, P, "", mod.groups, mod.precision, save=need)
        ctx.c, ctx.P, ctx.names = c, P, names
        ctx.need_dx = x.requires_grad
        return out

    @staticmethod
    def backward(ctx, dout):
        P = ctx.P
        G = {n: torch.zeros_like(P[n]) for n in ctx.names}
        dx = ggru_backward(ctx.c, dout.contiguous(), P, G, need_dx=ctx.need_dx)
        return (dx, None, None) + tuple(G[n] for n in ctx.names)


class GGRU(nn.Module):
    """model/cruse_net.py:14-55 (repair R1).  forward([B,C,T,F]) -> [B,C,T,F]."""

    def __init__(self, in_features=None, out_features=None, mid_features=None, hidden_size=1024, groups=2,
                 precision: str = DEFAULT_PREC):
        super().__init__()
        hidden_size_t = hidden_size // groups
        self.gru_list1 = nn.ModuleList([nn.GRU(hidden_size_t, hidden_size_t, 1, batch_first=True) for _ in range(groups)])
        self.gru_list2 = nn.ModuleList([nn.GRU(hidden_size_t, hidden_size_t, 1, batch_first=True) for _ in range(groups)])
        self.ln1 = nn.LayerNorm(hidden_size)
        self.ln2 = nn.LayerNorm(hidden_size)
        self.groups = groups
        self.mid_features = mid_features
        self.hidden_size = hidden_size
        self.precision = precision

    def forward(self, x):
        if x.dim() != 4:
            raise RuntimeError(f"GGRU expects [B,C,T,F], got {tuple(x.shape)}")
        B, C, T, F = x.shape
        if C * F != self.hidden_size:
            raise RuntimeError(f"GGRU: C*F = {C * F} does not match hidden_size = {self.hidden_size}")
        rows = x.transpose(1, 2).contiguous().view(B, T, C * F)          # cruse_net.py:39-40
        names = [n for n, _ in self.named_parameters()]
        params = [p for _, p in self.named_parameters()]
        out = _GGRUFn.apply(rows, self, names, *params)
        return out.view(B, T, C, F).transpose(1, 2).contiguous()         # cruse_net.py:53-54


class _Unet2Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mod, names, *params):
        P = dict(zip(names, params))
        Bf = dict(mod.named_buffers())
        need = any(p.requires_grad for p in params)
        mask, c = unet2_forward(x, P, Bf, mod.ch, mod.rnn_groups, mod.precision, mod.training, save=need)
        ctx.c, ctx.P, ctx.names = c, P, names
        return mask

    @staticmethod
    def backward(ctx, dmask):
        P, c = ctx.P, ctx.c
        used = [n for n in ctx.names if not (n.startswith("fc.") or n.startswith("bn1_t."))]
        G = {n: torch.zeros_like(P[n]) for n in used}
        B, T, F0 = c["B"], c["T"], c["F"][0]
        dlogit = ops.sigmoid_bwd(dmask.contiguous().view(B, T, 1, F0), c["mask"])
        unet2_backward(c, dlogit, P, G)
        return (None, None, None) + tuple(G.get(n) for n in ctx.names)


class unet_2(nn.Module):
    """model/cruse_net.py:129-165 with repairs R2-R8.  forward([B,1,T,160]) -> mask [B,1,T,160]."""

    def __init__(self, in_feat=161, ch=(1, 8, 16, 32, 64), stride=(1, 2), rnn_groups=4,
                 precision: str = DEFAULT_PREC):
        super().__init__()
        if tuple(stride) != (1, 2):
            raise RuntimeError("cruse_amd unet_2 implements stride (1,2) only (model/cruse_net.py:130 default)")
        self.laynum = len(ch) - 1
        hidden_size = in_feat // 2 ** self.laynum * ch[-1]
        self.ker_x = 2
        self.stride = stride
        self.padding = [self.ker_x - stride[0], 3 - stride[1]]
        for i in range(len(ch) - 1):
            k = i + 1
            setattr(self, f"conv{k}", nn.Conv2d(ch[k - 1], ch[k], (self.ker_x, 3), self.stride, self.padding))
            setattr(self, f"conv{k}_t", nn.ConvTranspose2d(ch[k], ch[k - 1], (1, 3), self.stride))
            setattr(self, f"bn{k}", nn.BatchNorm2d(ch[k]))
            setattr(self, f"bn{k}_t", nn.BatchNorm2d(ch[k - 1]))
            setattr(self, f"skip_connect_{k}", nn.Conv2d(ch[k], ch[k], (1, 3), padding=(0, 1), bias=False))
        self.gru = GGRU(hidden_size=hidden_size, groups=rnn_groups, precision=precision)
        self.elu = nn.ReLU()                      # named `elu` in the reference (:145), is a ReLU
        self.fc = nn.Linear(in_feat, in_feat)     # unused by forward (:146); kept for checkpoints
        self.ch = tuple(ch)
        self.rnn_groups = rnn_groups
        self.in_feat = in_feat
        self.hidden_size = hidden_size
        self.precision = precision

    def set_precision(self, precision: str) -> None:
        self.precision = precision
        self.gru.precision = precision

    def forward(self, x):
        if x.shape[-1] * self.ch[-1] // (1 << self.laynum) != self.hidden_size:
            raise RuntimeError(f"unet_2: {x.shape[-1]} input bins give hidden size "
                               f"{x.shape[-1] * self.ch[-1] // (1 << self.laynum)}, expected {self.hidden_size}")
        names = [n for n, _ in self.named_parameters()]
        params = [p for _, p in self.named_parameters()]
        return _Unet2Fn.apply(x.contiguous(), self, names, *params)
