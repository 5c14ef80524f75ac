"""Scheduling / numerics options of the training step as ONE explicit object.

`EngineConfig()` is the measured-best configuration (what bench.py runs); nothing in the package reads environment variables
at import or in the step.  `TrainEngine(model, config=EngineConfig(...))` takes its own copy, so two engines in one process do
not share option or side-stream state.  `EngineConfig.from_env()` -- called by bench.py and tools/*_probe.py only -- builds a
config from the CRUSE_* variables the A/B measurements of DESIGN.md section 6 were taken with.

Options that measured slower and are gone (DESIGN.md section 6 keeps the figures): half-batch pipelines through the GGRU
block (CRUSE_GRU_PIPES, 7.55 vs 6.94 ms), several side streams / lane maps (CRUSE_SIDE_STREAMS, CRUSE_SIDE_MAP: 6.43 vs 6.05
eager), joining the leaves on the main stream at an eager bucket boundary (CRUSE_EAGER_BOUNDARY=join).
"""
from __future__ import annotations

import contextlib
import os
from dataclasses import dataclass, field, fields, replace
from typing import Dict, Optional


@dataclass
class EngineConfig:
    overlap: bool = True            # leaf kernels (weight gradients, skip convs, dW GEMMs, clean STFT) on a side stream
    defer_mask: int = 15            # leaves queued for the next recurrence launch: 1 skip convs (fwd), 2 decoder weight gradients,
                                    # 4 GRU weight gradients, 8 skip-conv backward leaves
    inline_mask: int = 8            # backward leaves kept on the main stream: 1 skip dgrads, 2 skip wgrads, 4 decoder wgrads,
                                    # 8 << (k - 1) the level-k encoder wgrad, k = 1..4 (8: -0.025 ms; the others measured slower or equal)
    fuse_bn_stats: bool = True      # BatchNorm batch sums in the producing conv's epilogue (False: separate bn_stats pass)
    fuse_bn_bwd_stats: bool = True  # BatchNorm BACKWARD sums in the epilogue of the data-gradient conv that produces the incoming gradient
    gi_x3: Optional[int] = None     # forward gate projections: bit 0 / 1 = W_ih low-plane pass on layer 1 / 2, bit 2 = also split x
                                    # (None: 7 for Hg <= 320, else 3)
    gi_f16: int = 2                 # forward gate projections on IEEE-f16 operands (bit 0 / 1 = GGRU layer 1 / 2; with g = 1 the BatchNorm / LayerNorm
                                    # producers write the f16 operand copies) instead of bf16 x with W_ih hi / lo planes (gi_x3): x carries 11 bits instead of 8.
                                    # Layer 2 (default): ONE pass (cruse_gemm_f16_nt), 4.85 -> 4.76 ms (r4).  Layer 1 (bit 0, off by default): f16 x against f16
                                    # hi + lo planes of W_ih (cruse_gemm_f16x2_nt: two passes, the time of the bf16 form it replaces) -- mask 3 measures
                                    # 1.6e-4 instead of 2.8e-4 (torch init) / 3.7e-4 instead of 4.1e-4 (closed-form) at T = 401 and better gradients at bench
                                    # length and on fixture G6, but moves ONE vector of the closed-form fixture G16 (gru.ln1.bias) to 0.306 of its 0.30
                                    # tolerance, with or without the low plane: DESIGN.md section 2.  0: the gi_x3 forms on both layers
    fuse_bn_fwd: bool = True        # bf16 mode, training: BatchNorm-apply + ReLU (+ decoder skip add) of a level run inside the STAGING of
                                    # the convs that consume it (cruse_conv_*_bnin) -- the normalised tensors e_k (k < L) and u_k (k >= 2)
                                    # never exist in f32; the weight gradients read a bf16 copy the consuming conv writes while staging
    fuse_bn_bwd_apply: bool = True  # bf16 mode: the BatchNorm-backward "apply" pass (dy from dout, the pre-BN tensor and the batch sums) runs inside the
                                    # STAGING of the data-gradient conv that consumes dy (cruse_conv_*_bnbwd_in), which also writes the bf16 dy the
                                    # weight gradient reads; levels whose incoming gradient is f32 (from the GGRU / the last decoder layer) keep the pass
    bf16_dy: bool = True            # bf16 mode: the BatchNorm-backward outputs (dy, dv) are STORED as bf16 -- their two consumers (the
                                    # data-gradient conv on plain bf16 operands and the weight gradient) round them to bf16 anyway,
                                    # so the MFMAs see the same bits; 7 x 65.7 MB less written and 2 x that less read per step
    bf16_de: bool = True            # ... and the data gradients between a conv and the BatchNorm backward it feeds (du_k, de_k of the
                                    # levels whose producer AND consumers are MFMA kernels: 2 <= k < L) are stored as bf16 as well:
                                    # f32 accumulation, one rounding per pass of a backward-only tensor (needs bf16_dy and
                                    # fuse_bn_bwd_stats: the conv that stores the tensor also delivers its BatchNorm sums)
    gi_store_f16: bool = False      # bf16 mode, g = 1 at Hg = 640, chains of 8 clips: the gate pre-activations gi = x W_ih^T + b_ih STORED as IEEE f16 rows
                                    # (cruse_gemm_nt_out16 -> cruse_gru_seq_fwd_gi16): f32 accumulation, one rounding to 11 bits at the store; 12.72 -> 12.33 GB
                                    # of HBM traffic per step (PMC, profiles/r06_pmc_hbm_traffic_gi_store_f16.csv).  Enhanced spectrum at T = 401: 4.10e-4 ->
                                    # 4.25e-4 (closed-form) / 2.81e-4 -> 2.81e-4 (torch init), every fixture of the bf16 mode holds (bf16 rows do not, r3).
                                    # OFF by default: the step is 0.05-0.08 ms SLOWER with it on two boxes of three (equal on the third; interleaved A/B of
                                    # 4 x 100 steps in one process) although the kernels alone net +3..10 us (projections -5 us each, the recurrence's helper
                                    # wave +40 cycles per step for the conversions) -- bytes are not what this step waits for at that point
    dx_atr: bool = True             # bf16 mode, g = 1: the input gradient dX = dgi . W_ih is formed from the TIME-MAJOR gate gradients dgT (the operand of the
                                    # weight-gradient GEMMs) through transposing LDS reads (cruse_gemm_bf16_nt_atr): the gate-gradient pass writes no row-major dgi
                                    # (98 MB per layer written and read).  Same products, same summation order (bit-identical GEMM: tests/test_gpu_kernels.py).
                                    # Round 4 measured it time-neutral and pruned it; round 6 (fixed stream conditions): 4.676 against 4.688 ms in four
                                    # interleaved A/B pairs of 100 steps, -0.2 GB of HBM traffic per step
    pick_launch_stream: bool = True # HIP-graph form: time the replay of a fresh capture from the current stream and three pool streams and
                                    # keep the fastest launcher (engine._pick_launch_stream: a graph's own side-branch streams may share
                                    # the launching stream's hardware queue, which serialises the branches -- +25..35 % per step)
    lib_options: Dict[str, int] = field(default_factory=dict)       # cruse_set_option(name, value) while this config is active

    _ENV = {"overlap": ("CRUSE_OVERLAP", lambda v: v == "1"), "defer_mask": ("CRUSE_DEFER", int), "inline_mask": ("CRUSE_INLINE", int),
            "fuse_bn_stats": ("CRUSE_FUSE_BN_STATS", lambda v: v != "0"),
            
            "fuse_bn_bwd_stats": ("CRUSE_FUSE_BN_BWD", lambda v: v != "0"), 
            "gi_x3": ("CRUSE_GI_X3", int), "gi_f16": ("CRUSE_GI_F16", int), 
             
            "fuse_bn_fwd": ("CRUSE_FUSE_BN_FWD", lambda v: v != "0"), "fuse_bn_bwd_apply": ("CRUSE_FUSE_BN_BWD_APPLY", lambda v: v != "0"), "bf16_dy": ("CRUSE_BF16_DY", lambda v: v != "0"), "bf16_de": ("CRUSE_BF16_DE", lambda v: v != "0"),
            "pick_launch_stream": ("CRUSE_PICK_LAUNCH_STREAM", lambda v: v != "0"), "gi_store_f16": ("CRUSE_GI_STORE_F16", lambda v: v != "0"),
            "dx_atr": ("CRUSE_DX_ATR", lambda v: v != "0")}
    _LIB_ENV = {"CRUSE_GRU_BWD_RS": "gru_bwd_rs", "CRUSE_GRU_FWD_LEAN": "gru_fwd_lean", "CRUSE_GRU_WLO": "gru_wlo", "CRUSE_GRU_DBG": "gru_dbg",
                "CRUSE_GRU_TF": "gru_tf", "CRUSE_GRU_POLL_FWD": "gru_poll_fwd", "CRUSE_GRU_POLL_BWD": "gru_poll_bwd", "CRUSE_CM_KINT": "cm_kint",
                "CRUSE_CM_SWAP": "cm_swap", "CRUSE_CM_NW": "cm_nw", "CRUSE_PW_VALU": "pw_valu", "CRUSE_WG_DBG": "wg_dbg",
                "CRUSE_WG_RD": "wg_rd", "CRUSE_WG_GRID": "wg_grid"}

    @classmethod
    def from_env(cls, env=None) -> "EngineConfig":
        """bench.py / tools only: the CRUSE_* variables of the A/B measurements -> an explicit config."""
        env = os.environ if env is None else env
        kw = {}
        for name, (var, conv) in cls._ENV.items():
            if var in env:
                kw[name] = conv(env[var])
        lib = {opt: int(env[var]) for var, opt in cls._LIB_ENV.items() if var in env}
        return cls(lib_options=lib, **kw)

    def non_default(self) -> dict:
        d = EngineConfig()
        return {f.name: getattr(self, f.name) for f in fields(self) if getattr(self, f.name) != getattr(d, f.name)}

    def copy(self, **kw) -> "EngineConfig":
        return replace(self, lib_options=dict(self.lib_options), **kw)


_ACTIVE = EngineConfig()


def get() -> EngineConfig:
    """the configuration of the step being issued (the engine's own inside TrainEngine calls, the default outside)."""
    return _ACTIVE


@contextlib.contextmanager
def use(cfg: EngineConfig):
    """make cfg the active configuration (and its library options) for the duration of the block; one Python thread drives a
    GPU, so this is plain save / restore."""
    global _ACTIVE
    from . import ops
    prev = _ACTIVE
    saved = {k: ops.get_option(k) for k in cfg.lib_options}
    _ACTIVE = cfg
    for k, v in cfg.lib_options.items():
        ops.set_option(k, v)
    try:
        yield cfg
    finally:
        _ACTIVE = prev
        for k, v in saved.items():
            ops.set_option(k, v)
