import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2])
tot = sum(float(r['TotalDurationNs']) for r in rows)
print(tot / steps / 1e6, 'ms/step', sum(int(r['Calls']) for r in rows) / steps, 'launches/step')
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 24]:
    print(f"{float(r['TotalDurationNs'])/steps/1e3:7.1f} us/step calls/step {int(r['Calls'])/steps:5.1f} avg {float(r['AverageNs'])/1e3:7.1f} us  {r['Name'][:110]}")
