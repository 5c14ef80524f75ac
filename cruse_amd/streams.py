"""Which HIP stream overlaps with which -- measured, not assumed.

ROCm maps HIP streams onto a handful of hardware queues (4 per priority by default), in creation order.  Two streams that
land on one queue -- or a high-priority stream and the normal queue that shares its pipe -- do NOT run side by side: a leaf
kernel issued on such a "side" stream serialises with the main stream, and a step whose schedule hides 3 ms of leaves behind
the recurrences takes 5.8 instead of 3.5 ms (round 6: every fourth torch pool stream did this to an engine; which engine got
the bad stream depended on how many streams the process had drawn before -- BENCH_r05's config-4 row went 3.56 -> 5.92 ms
when one more bench row was inserted in front of it).  `torch.cuda.Stream()` hands out pool streams round-robin, so the
only reliable statement about a pair of streams is a measurement:

    pair_overlaps(main, side): one workgroup idles ~100 us on each stream (cruse_cu_hog), the side one ordered behind an
    event of the main stream as the engine's leaves are; read against one such kernel alone and two back to back on the
    main stream, overlapping streams come out at a serial fraction of 0.1-0.3, serialised ones at 0.85-1.1.

`side_stream_for(main)` draws pool streams until one overlaps (at most 8 draws, the best otherwise) and remembers the choice
per main stream for the process, so every engine of a process uses the same proven pair.  Nothing here runs during HIP-graph
capture (a capture only records topology; the streams of a replay are the graph's own, see TrainEngine._pick_launch_stream).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from . import ops

HOG_US = 100.0
MAX_DRAWS = 8

_CHOSEN: Dict[Tuple[int, int], torch.cuda.Stream] = {}
REPORT: List[dict] = []     # one entry per selection: what was measured (bench.py prints it)


def _timed(fn, main, trials: int = 3) -> float:
    best = float("inf")
    for _ in range(trials):
        e0 = torch.cuda.Event(enable_timing=True)
        e1 = torch.cuda.Event(enable_timing=True)
        e0.record(main)
        fn()
        e1.record(main)
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best


def _hog(stream, n: int = 1) -> None:
    with torch.cuda.stream(stream):
        for _ in range(n):
            ops.cu_hog(1, HOG_US)


def calibrate(main) -> Tuple[float, float]:
    """-> (ms of one idling kernel, ms of two back to back) on `main`: the two ends of the scale a pair is read against"""
    main.synchronize()
    return _timed(lambda: _hog(main, 1), main), _timed(lambda: _hog(main, 2), main)


def pair_ms(main, side) -> float:
    """both streams idle one workgroup for HOG_US, the side stream behind an event of the main stream (as the engine's leaves are);
    main joins the side stream"""
    def one():
        ev = torch.cuda.Event()
        ev.record(main)
        _hog(main)
        side.wait_event(ev)
        _hog(side)
        main.wait_stream(side)
    return _timed(one, main)


def serial_fraction(main, side, cal=None) -> float:
    """0 = the two kernels ran side by side, 1 = one after the other (self-calibrated: event and cross-stream overheads cancel to
    first order; overlapping pairs measure 0.1-0.3, serialised ones 0.85-1.1)"""
    t1, t2 = cal if cal is not None else calibrate(main)
    return (pair_ms(main, side) - t1) / max(t2 - t1, 1e-6)


def pair_overlaps(main, side, cal=None) -> Tuple[bool, float]:
    f = serial_fraction(main, side, cal)
    return f < 0.5, f


def side_stream_for(main: Optional[torch.cuda.Stream] = None) -> torch.cuda.Stream:
    """a normal-priority pool stream that runs beside `main` (default: the current stream), chosen once per main stream"""
    main = torch.cuda.current_stream() if main is None else main
    key = (main.device_index if hasattr(main, "device_index") else torch.cuda.current_device(), int(main.cuda_stream))
    s = _CHOSEN.get(key)
    if s is not None:
        return s
    if torch.cuda.is_current_stream_capturing():
        # no measurement inside a capture (and none needed: the replay's streams are the graph's own) -- not remembered
        return torch.cuda.Stream()
    tried, best = [], None
    cal = calibrate(main)
    for _ in range(MAX_DRAWS):
        cand = torch.cuda.Stream()
        if int(cand.cuda_stream) == int(main.cuda_stream):
            continue
        ok, frac = pair_overlaps(main, cand, cal)
        tried.append(round(frac, 2))
        if best is None or frac < best[0]:
            best = (frac, cand)
        if ok:
            break
    s = best[1]
    _CHOSEN[key] = s
    REPORT.append({"main_stream": hex(key[1]), "main_priority": getattr(main, "priority", None), "serial_fraction_by_draw": tried,
                   "kept": round(best[0], 2), "overlaps": bool(best[0] < 0.5)})
    return s


def stream_beside(main: Optional[torch.cuda.Stream] = None, avoid=(), tag: str = "copy") -> torch.cuda.Stream:
    """a further pool stream that overlaps with `main` AND with every stream in `avoid` (e.g. the trainer's H2D copy stream beside the
    compute stream and its leaf stream), chosen once per (main stream, tag) for the process -- so a stream is not drawn anew every epoch
    (blocks the caching allocator holds for a dropped stream cannot be reused by the next one)"""
    main = torch.cuda.current_stream() if main is None else main
    key = (main.device_index, int(main.cuda_stream), tag)
    s = _CHOSEN.get(key)
    if s is not None:
        return s
    taken = {int(main.cuda_stream)} | {int(a.cuda_stream) for a in avoid}
    cal = calibrate(main)
    tried, best = [], None
    for _ in range(MAX_DRAWS):
        cand = torch.cuda.Stream()
        if int(cand.cuda_stream) in taken:
            continue
        frac = max([serial_fraction(main, cand, cal)] + [serial_fraction(a, cand) for a in avoid])
        tried.append(round(frac, 2))
        if best is None or frac < best[0]:
            best = (frac, cand)
        if frac < 0.5:
            break
    s = best[1] if best is not None else torch.cuda.Stream()
    _CHOSEN[key] = s
    REPORT.append({"main_stream": hex(key[1]), "tag": tag, "serial_fraction_by_draw": tried, "kept": round(best[0], 2) if best else None})
    return s


def reset() -> None:
    _CHOSEN.clear()
    REPORT.clear()
