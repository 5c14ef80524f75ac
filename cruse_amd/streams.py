"""Which HIP stream overlaps with which -- measured, not assumed.

ROCm maps HIP streams onto a handful of hardware queues (4 per priority by default), in creation order.  Two streams that
land on one queue -- or a high-priority stream and the normal queue that shares its pipe -- do NOT run side by side: a leaf
kernel issued on such a "side" stream serialises with the main stream, and a step whose schedule hides 3 ms of leaves behind
the recurrences takes 5.8 instead of 3.5 ms (round 6: every fourth torch pool stream did this to an engine; which engine got
the bad stream depended on how many streams the process had drawn before -- BENCH_r05's config-4 row went 3.56 -> 5.92 ms
when one more bench row was inserted in front of it).  `torch.cuda.Stream()` hands out pool streams round-robin, so the
only reliable statement about a pair of streams is a measurement:

    pair_overlaps(main, side): one workgroup idles ~60 us on each stream (cruse_cu_hog), the side one ordered behind an
    event of the main stream as the engine's leaves are; overlapping streams finish both in ~1.0x one kernel's time,
    serialised ones in ~1.6-1.8x.

`side_stream_for(main)` draws pool streams until one overlaps (at most 8 draws, the best otherwise) and remembers the choice
per main stream for the process, so every engine of a process uses the same proven pair.  Nothing here runs during HIP-graph
capture (a capture only records topology; the streams of a replay are the graph's own, see TrainEngine._pick_launch_stream).
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from . import ops

HOG_US = 60.0
GOOD_RATIO = 1.30          # pair time / single time: overlapping pairs measure 1.00-1.10, serialised ones 1.55-1.80
MAX_DRAWS = 8

_CHOSEN: Dict[Tuple[int, int], torch.cuda.Stream] = {}
REPORT: List[dict] = []     # one entry per selection: what was measured (bench.py prints it)


def _timed(fn, main) -> float:
    e0 = torch.cuda.Event(enable_timing=True)
    e1 = torch.cuda.Event(enable_timing=True)
    e0.record(main)
    fn()
    e1.record(main)
    e1.synchronize()
    return e0.elapsed_time(e1)


def single_ms(main, trials: int = 2) -> float:
    def one():
        with torch.cuda.stream(main):
            ops.cu_hog(1, HOG_US)
    return min(_timed(one, main) for _ in range(trials))


def pair_ms(main, side, trials: int = 2) -> float:
    """both streams idle one workgroup for HOG_US, the side stream behind an event of the main stream; main joins the side"""
    def one():
        ev = torch.cuda.Event()
        ev.record(main)
        with torch.cuda.stream(main):
            ops.cu_hog(1, HOG_US)
        side.wait_event(ev)
        with torch.cuda.stream(side):
            ops.cu_hog(1, HOG_US)
        main.wait_stream(side)
    return min(_timed(one, main) for _ in range(trials))


def pair_overlaps(main, side) -> Tuple[bool, float]:
    main.synchronize()
    t1 = single_ms(main)
    tp = pair_ms(main, side)
    return tp < GOOD_RATIO * t1, tp / max(t1, 1e-6)


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
    for _ in range(MAX_DRAWS):
        cand = torch.cuda.Stream()
        if int(cand.cuda_stream) == int(main.cuda_stream):
            continue
        ok, ratio = pair_overlaps(main, cand)
        tried.append(round(ratio, 3))
        if best is None or ratio < best[0]:
            best = (ratio, cand)
        if ok:
            break
    s = best[1]
    _CHOSEN[key] = s
    REPORT.append({"main_stream": hex(key[1]), "main_priority": getattr(main, "priority", None), "pair_over_single": tried,
                   "kept": round(best[0], 3), "overlaps": bool(best[0] < GOOD_RATIO)})
    return s


def reset() -> None:
    _CHOSEN.clear()
    REPORT.clear()
