#!/usr/bin/env python3
"""Per-kernel register / scratch / LDS table of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage).
usage: tools/kernel_resources.py cruse_amd/csrc/gru_w16.hip [name filter]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src, "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: ([A-Za-z ]+?)(?: \[[^\]]*\])?: (\S+)", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k == "Function Name":
        name = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(anonymous namespace\)::|cruse_gru::GruArgs", "", name)}
        rows.append(cur)
    elif cur is not None:
        cur[k] = v
print(f"{'kernel':70s} {'VGPR':>5s} {'AGPR':>5s} {'SGPR':>5s} {'spill':>6s} {'scratch':>8s} {'occ':>4s} {'LDS':>7s}")
for r in rows:
    if flt in r["name"]:
        print(f"{r['name'][:70]:70s} {r.get('VGPRs','?'):>5s} {r.get('AGPRs','?'):>5s} {r.get('TotalSGPRs','?'):>5s} {r.get('VGPRs Spill','?'):>6s} "
              f"{r.get('ScratchSize','?'):>8s} {r.get('Occupancy','?'):>4s} {r.get('LDS Size','?'):>7s}")
