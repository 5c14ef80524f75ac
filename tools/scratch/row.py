"""print one secondary row of the bench line on stdin: python tools/scratch/row.py name [name...]"""
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
for n in sys.argv[1:]:
    print(n, json.dumps(d.get("secondary", {}).get(n)))
