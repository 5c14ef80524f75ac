"""Profiling aid: from a rocprofv3 kernel trace CSV, report per step what ran concurrently with the GRU kernels."""
import csv, sys, collections


def main(path, skip_steps=4):
    rows = list(csv.DictReader(open(path)))
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", ""))
          for r in rows]
    ev.sort()
    gru = [(s, e, n) for s, e, n in ev if "gru_fwd" in n or "gru_bwd" in n]
    # a step = 2 fwd + 2 bwd GRU launches; take the last complete one
    last = gru[-4:]
    t0 = last[0][0] - 3_000_000
    t1 = last[-1][1] + 6_000_000
    print(f"window {(t1 - t0) / 1e6:.2f} ms")
    for s, e, n in last:
        tot = collections.Counter()
        for s2, e2, n2 in ev:
            if n2 == n and s2 == s:
                continue
            ov = min(e, e2) - max(s, s2)
            if ov > 0:
                tot[n2.split("(")[0][:50]] += ov
        print(f"{n[:40]:40s} {(e - s) / 1e3:8.1f} us; concurrent: " + ", ".join(f"{k} {v / 1e3:.0f}" for k, v in tot.most_common(8)))
    # timeline of the last step
    stepev = [(s, e, n) for s, e, n in ev if s >= last[0][0] - 2_500_000 and s <= last[-1][1] + 5_000_000]
    base = stepev[0][0]
    for s, e, n in stepev:
        print(f"{(s - base) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {n.split('(')[0][:70]}")


if __name__ == "__main__":
    main(sys.argv[1])
