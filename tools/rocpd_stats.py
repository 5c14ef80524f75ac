"""rocprofv3 (ROCm 7.2) writes a rocpd SQLite database by default; this turns its kernel dispatches into the
kernel-stats CSV kept under profiles/ (same columns as `rocprofv3 --stats` CSV output) and, with --timeline, prints the
step period and the serial phases of one training step (first dispatch of each kernel family after the STFT).

    python tools/rocpd_stats.py gpurun_out/.../p_results.db profiles/r02_kernel_stats.csv [--timeline]
"""
import csv
import sqlite3
import statistics
import sys


def main():
    db, out = sys.argv[1], sys.argv[2]
    c = sqlite3.connect(db)
    rows = c.execute("""select S.display_name, K.start, K.end, K.stream_id, S.arch_vgpr_count, S.accum_vgpr_count, S.sgpr_count
                        from rocpd_kernel_dispatch K join rocpd_info_kernel_symbol S on S.id = K.kernel_id and S.guid = K.guid
                        order by K.start""").fetchall()
    by = {}
    for name, st, en, *_ in rows:
        by.setdefault(name, []).append(en - st)
    total = sum(sum(v) for v in by.values())
    with open(out, "w", newline="") as f:
        w = csv.writer(f, quoting=csv.QUOTE_NONNUMERIC)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for name, v in sorted(by.items(), key=lambda kv: -sum(kv[1])):
            w.writerow([name, len(v), sum(v), round(sum(v) / len(v), 1), round(100.0 * sum(v) / total, 4), min(v), max(v),
                        round(statistics.pstdev(v), 1)])
    print(f"{len(rows)} dispatches, {len(by)} kernels -> {out}")
    if "--timeline" in sys.argv:
        stft = [st for name, st, *_ in rows if "stft320_kernel<0>" in name]
        starts = stft[::2]                                   # two STFT launches (noisy, clean) per step
        per = [b - a for a, b in zip(starts, starts[1:])]
        if per:
            tail = per[len(per) // 2:]
            print(f"step period (stft -> stft), last {len(tail)} steps: median {statistics.median(tail) / 1e6:.3f} ms")
        regs = {}
        for name, _, _, _, av, acc, sg in rows:
            regs[name] = (av, acc, sg)
        for name in sorted(regs):
            if "gru_" in name and "kernel" in name:
                print("  regs", name[:90], "arch_vgpr %s accum_vgpr %s sgpr %s" % regs[name])


if __name__ == "__main__":
    main()
