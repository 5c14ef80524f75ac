"""Serial-chain view of ONE training step from a rocprofv3 rocpd database (eager launches): every dispatch of the step in
start order with its stream, duration and the idle gap on ITS stream before it, plus per-stream busy time -- what the
step's wall time is made of once kernels overlap.

    python tools/rocpd_chain.py gpurun_out/.../p_results.db [step_index_from_end=3] [--all]
"""
import sqlite3
import sys


def short(name: str) -> str:
    n = name.replace("(anonymous namespace)::", "").replace("void ", "")
    if n.startswith("_ZN"):
        for key in ("gru_gate_grads_bf16", "gru_gate_grads", "gemm_bf16"):
            if key in n:
                return key + "<mangled>"
    return n.split("(")[0][:46]


def main():
    db = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else 3
    c = sqlite3.connect(db)
    rows = c.execute("""select S.display_name, K.start, K.end, K.stream_id
                        from rocpd_kernel_dispatch K join rocpd_info_kernel_symbol S on S.id = K.kernel_id and S.guid = K.guid
                        order by K.start""").fetchall()
    stft = [i for i, r in enumerate(rows) if "stft320_kernel<0>" in r[0]]
    # the noisy STFT opens a step; the clean one follows within the same step
    opens = [i for j, i in enumerate(stft) if j == 0 or rows[i][1] - rows[stft[j - 1]][1] > 2_000_000]
    a, b = opens[-back - 1], opens[-back]
    step = rows[a:b]
    t0 = step[0][1]
    # the step's first dispatch may be the arena zero just before the STFT
    print(f"step of {len(step)} dispatches, {(rows[b][1] - t0) / 1e6:.3f} ms")
    main_stream = step[0][3]
    last_end = {}
    busy = {}
    for name, st, en, sid in step:
        gap = st - last_end.get(sid, st)
        last_end[sid] = max(en, last_end.get(sid, en))
        busy[sid] = busy.get(sid, 0) + (en - st)
        if "--all" in sys.argv or sid == main_stream:
            print(f"{(st - t0) / 1e3:9.1f} us  s{sid:<3d} {short(name):46s} {(en - st) / 1e3:8.1f} us   gap {gap / 1e3:7.1f}")
    for sid, v in sorted(busy.items()):
        print(f"stream {sid}: busy {v / 1e6:.3f} ms" + ("  (main)" if sid == main_stream else ""))


if __name__ == "__main__":
    main()
