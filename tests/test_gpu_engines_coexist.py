"""Engines of ONE process must not influence each other.

BENCH_r05's config-4 row went 3.56 -> 5.92 ms when one more bench row was inserted in front of it.  Cause (round 6,
cruse_amd/streams.py): `torch.cuda.Stream()` hands out pool streams round-robin and ROCm places streams on four hardware queues
in creation order -- every fourth side stream shared the main stream's queue and serialised the leaves with the recurrences;
which engine drew it depended on how many streams the process had drawn before.  The side stream is now chosen by measuring
that it overlaps, and a captured step's launching stream by timing its replay.  These tests pin both, and the promise of
cruse_amd/config.py that two engines with different EngineConfigs share no option state.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

L = 64000          # 4 s clips: the shapes of BASELINE configs 2 / 4 (a step of 3.5-4.7 ms -- long enough to time to a few %)


def _engine(groups, cfg=None, graph=False, loss="wo_male", lr=0.0):
    from cruse_amd.engine import TrainEngine
    from cruse_amd.model.cruse_net import unet_2
    torch.manual_seed(0)
    m = unet_2(rnn_groups=groups, precision="bf16").cuda()
    return TrainEngine(m, lr=lr, use_graph=graph, loss=loss, config=cfg)


def _median_step_ms(eng, pool, n=9, warm=3):
    for i in range(warm):
        eng.step(*pool[i % len(pool)])
    torch.cuda.synchronize()
    ts = []
    for i in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eng.step(*pool[i % len(pool)]); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    return sorted(ts)[len(ts) // 2]


def test_step_time_does_not_depend_on_how_many_streams_the_process_has_drawn():
    """six engines built one after the other, 0..5 extra pool streams drawn in between: every eager step within 10 % of the fastest
    (measured: 1-2 %), and the same for the graph form (before the fix: +15 % ... +65 % for every fourth)"""
    from cruse_amd import streams
    from cruse_amd.data import synth_batch
    pool = [synth_batch(32, L, torch.device("cuda"), 7000 + i) for i in range(2)]
    for graph in (False, True):
        times = []
        keep = []
        for k in range(6):
            keep.append([torch.cuda.Stream() for _ in range(k)])          # shift the round-robin position of what follows
            e = _engine(4, graph=graph, loss="wo_male_df")
            times.append(_median_step_ms(e, pool))
            keep.append(e)
        assert max(times) <= 1.10 * min(times), (graph, times, streams.REPORT)
        del keep


def test_side_stream_probe_tells_serialised_pairs_from_overlapping_ones():
    """a stream against itself serialises by construction (ratio ~2); among eight pool streams at least one overlaps with the current
    stream and the chosen one does"""
    from cruse_amd import streams
    main = torch.cuda.current_stream()
    ok_self, r_self = streams.pair_overlaps(main, main)
    assert not ok_self and r_self > 0.7, r_self
    s = streams.side_stream_for(main)
    ok, r = streams.pair_overlaps(main, s)
    assert ok, (r, streams.REPORT)
    assert streams.side_stream_for(main) is s            # remembered per main stream


def test_two_engines_with_different_configs_interleaved():
    """engine A (g = 1, default config) and engine B (g = 1, gi_f16 = 3 + a library option) stepped alternately on the same batches give
    bit-identical losses and the gradients of their own solo runs (tolerance: the run-to-run spread of the BatchNorm f64 atomics), and
    take their solo step time +- 6 % (measured: within 1 %)"""
    from cruse_amd.config import EngineConfig
    from cruse_amd.data import synth_batch
    pool = [synth_batch(16, L, torch.device("cuda"), 8000 + i) for i in range(3)]
    cfg_b = EngineConfig(gi_f16=3, bf16_de=False, lib_options={"gru_poll_fwd": 4})

    def solo(cfg):
        e = _engine(1, cfg=cfg)
        losses = [float(e.step(*pool[i % 3]).item()) for i in range(6)]
        grads = e.flat.grads.clone()
        return losses, grads, _median_step_ms(e, pool)

    la, ga, ta = solo(None)
    lb, gb, tb = solo(cfg_b)
    assert la != lb                                       # (the two configurations really are different arithmetic)
    ea, eb = _engine(1), _engine(1, cfg=cfg_b)
    la2, lb2 = [], []
    for i in range(6):
        la2.append(float(ea.step(*pool[i % 3]).item()))
        lb2.append(float(eb.step(*pool[i % 3]).item()))
    assert la2 == la and lb2 == lb
    from util import rel_l2
    assert rel_l2(ea.flat.grads, ga) < 5e-5 and rel_l2(eb.flat.grads, gb) < 5e-5
    # timed alternately: A, B, A, B, ...
    tsa, tsb = [], []
    for i in range(9):
        for eng, ts in ((ea, tsa), (eb, tsb)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); eng.step(*pool[i % 3]); e1.record(); e1.synchronize()
            ts.append(e0.elapsed_time(e1))
    ma, mb = sorted(tsa)[4], sorted(tsb)[4]
    assert abs(ma / ta - 1) < 0.06 and abs(mb / tb - 1) < 0.06, (ma, ta, mb, tb)
    from cruse_amd import ops
    assert ops.get_option("gru_poll_fwd") is None         # B's library option did not outlive B's step
