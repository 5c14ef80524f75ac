"""LDS bank-conflict share per kernel from a rocprofv3 PMC pass:
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace --output-format csv -d DIR -o p -- python bench.py ...
    python tools/pmc_lds.py DIR/p_counter_collection.csv [out.csv]
conflict share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE (cycles the LDS spent re-issuing conflicting accesses / cycles it was
busy).  This is how the 8-way conflict of the forward recurrences' K-reduction reads was found (r03)."""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    m = re.match(r"_ZN\d+_GLOBAL__N_1(\d+)([A-Za-z_0-9]+)", name)
    return m.group(2)[:int(m.group(1))] if m else name


def main():
    acc = {}
    with open(sys.argv[1]) as f:
        for r in csv.DictReader(f):
            a = acc.setdefault(short(r["Kernel_Name"]), {})
            c = a.setdefault(r["Counter_Name"], [0, 0.0])
            c[0] += 1; c[1] += float(r["Counter_Value"])
    rows = []
    for k, a in acc.items():
        conf = a.get("SQ_LDS_BANK_CONFLICT", [0, 0.0]); act = a.get("SQ_LDS_IDX_ACTIVE", [0, 0.0]); ins = a.get("SQ_INSTS_LDS", [0, 0.0])
        n = max(conf[0], act[0], 1)
        rows.append((conf[1] / n, k, n, act[1] / n, ins[1] / n))
    rows.sort(reverse=True)
    out = ["kernel,launches,lds_bank_conflict_cycles_per_launch,lds_active_cycles_per_launch,conflict_share,lds_instructions_per_launch"]
    for cf, k, n, ac, ins in rows:
        out.append(f"\"{k}\",{n},{cf:.0f},{ac:.0f},{cf / ac if ac else 0:.3f},{ins:.0f}")
    text = "\n".join(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text + "\n")
    print(text)


if __name__ == "__main__":
    main()
