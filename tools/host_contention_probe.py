"""what the first 8-GPU run does to the HOST side of a rank, measured without the node.  Eight ranks each issue
~170 eager launches per 5 ms step from one Python thread; this probe runs the bench step on the one GPU while N - 1 sibling
processes spin the same kind of launch loop against the library's host-only entry points (ctypes calls that validate and return:
no device), pinned or unpinned, and reports ms/step for eager launches and for HIP-graph replay -- the figure `auto`
(TrainEngine(use_graph="auto"), bench.py's warm-up) decides on, per rank, reduced with MAX over the ranks.
    python tools/host_contention_probe.py [siblings=7] [steps=30]"""
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def sibling(stop, idx):
    from cruse_amd._lib import lib                      # host-side work of a launch loop: ctypes marshalling + argument checks
    n = 0
    while not stop.is_set():
        for _ in range(200):
            lib.cruse_gemm(0, 0, 0, 4, 4, None, 4, None, 4, None, 4, None, 0, 1, 0, 0, None)      # rejected on the host: no device call
            n += 1
    return n


def measure(steps):
    import torch
    from cruse_amd.data import synth_batch
    from cruse_amd.engine import TrainEngine
    from cruse_amd.model.cruse_net import unet_2
    torch.manual_seed(0)
    pool = [synth_batch(64, 64000, "cuda", 40 + i) for i in range(2)]
    out = {}
    for graph in (True, False):
        eng = TrainEngine(unet_2(rnn_groups=1, precision="bf16").cuda(), use_graph=graph)
        for s in range(4):
            eng.step(*pool[s % 2])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for s in range(steps):
            eng.step(*pool[s % 2])
        torch.cuda.synchronize()
        out["graph" if graph else "eager"] = (time.perf_counter() - t0) / steps * 1e3
    return out


def main():
    nsib = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    print(f"host: {os.cpu_count()} logical CPUs; OMP_NUM_THREADS={os.environ.get('OMP_NUM_THREADS')}")
    base = measure(steps)
    print(f"alone:               graph {base['graph']:.3f} ms/step   eager {base['eager']:.3f} ms/step")
    ctx = mp.get_context("spawn")
    for label, pin in (("siblings, unpinned", False), ("siblings, all pinned to this rank's cores", True)):
        stop = ctx.Event()
        procs = [ctx.Process(target=sibling, args=(stop, i)) for i in range(nsib)]
        for p in procs:
            p.start()
        if pin and hasattr(os, "sched_setaffinity"):
            cores = sorted(os.sched_getaffinity(0))[:max(2, len(os.sched_getaffinity(0)) // 8)]
            for p in procs:
                os.sched_setaffinity(p.pid, cores)       # the worst case: every rank's launch loop on one eighth of the cores
            os.sched_setaffinity(0, cores)
        time.sleep(1.0)
        r = measure(steps)
        stop.set()
        for p in procs:
            p.join(timeout=10)
        print(f"{nsib} {label}: graph {r['graph']:.3f} ms/step   eager {r['eager']:.3f} ms/step   -> auto keeps {'graph' if r['graph'] <= r['eager'] else 'eager'}")


if __name__ == "__main__":
    main()
