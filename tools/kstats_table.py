"""Print a rocprofv3 --stats kernel table (p_kernel_stats.csv) as per-step microseconds.  usage: kstats_table.py <csv> <steps> [top]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:top]:
    print(f"{r['Name'][:90]:90s} n={r['Calls']:>5s} avg {float(r['AverageNs']) / 1e3:7.1f} us {float(r['TotalDurationNs']) / tot * 100:5.1f}%  per step {float(r['TotalDurationNs']) / steps / 1e3:7.1f} us")
print(f"all kernels per step: {tot / steps / 1e3:.1f} us")
