"""Where do a kernel's register spills and VALU ops sit?  Reads the gfx950 assembly of one source file (hipcc -S
--cuda-device-only) and reports, per kernel whose name matches argv[2], the backward branches (loops), the MFMA range and
the scratch_* instructions relative to them.  usage: python tools/isa_loop_report.py file.s name-substring"""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ''
funcs = re.split(r'\n(?=_Z\w+:)', s)
for f in funcs:
    name = f.split(':')[0]
    if pat not in name:
        continue
    lines = f.split('\n')
    mf = [i for i, l in enumerate(lines) if 'v_mfma' in l]
    if not mf:
        continue
    lab = {}
    for i, l in enumerate(lines):
        m = re.match(r'(\.LBB\d+_\d+):', l)
        if m:
            lab[m.group(1)] = i
    loops = []
    for i, l in enumerate(lines):
        m = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if m and m.group(1) in lab and lab[m.group(1)] < i:
            loops.append((lab[m.group(1)], i))
    sc = [i for i, l in enumerate(lines) if 'scratch_' in l]
    print(name[:90])
    print('  mfma range', mf[0], mf[-1], 'count', len(mf), '| scratch ops', len(sc))
    for a, b in loops:
        n_sc = sum(1 for i in sc if a <= i <= b)
        n_mf = sum(1 for i in mf if a <= i <= b)
        if n_mf == 0 and n_sc == 0:
            continue
        valu = sum(1 for i in range(a, b) if re.match(r'\s+v_', lines[i]) and 'v_mfma' not in lines[i])
        ds = sum(1 for i in range(a, b) if re.match(r'\s+ds_', lines[i]))
        bar = sum(1 for i in range(a, b) if 's_barrier' in lines[i])
        vm = sum(1 for i in range(a, b) if re.match(r'\s+(buffer_|global_)', lines[i]))
        print(f'  loop [{a},{b}] len {b - a}: mfma {n_mf} scratch {n_sc} valu {valu} ds {ds} vmem {vm} barriers {bar}')
