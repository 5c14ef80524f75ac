"""per kernel (name substring argv[2:]) of a gfx950 .s file: loads, s_waitcnt vmcnt, conditional branches, VGPRs, occupancy"""
import re, sys
s = open(sys.argv[1]).read()
for f in re.split(r'\n(?=_Z\w+:)', s):
    name = f.split(':')[0]
    if not any(p in name for p in sys.argv[2:]):
        continue
    lines = f.split('\n')
    vm = sum(1 for l in lines if re.match(r'\s+(global_load|buffer_load)', l))
    wc = sum(1 for l in lines if 's_waitcnt vmcnt' in l)
    br = sum(1 for l in lines if re.match(r'\s+s_cbranch', l))
    print(name[:80], 'loads', vm, 'waitcnt vm', wc, 'cbranch', br, 'lines', len(lines), 'vgprs', re.findall(r'; NumVgprs: (\d+)', f),
          'scratch', re.findall(r'; ScratchSize: (\d+)', f), 'occ', re.findall(r'; Occupancy: (\d+)', f))
