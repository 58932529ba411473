#!/usr/bin/env python3
"""Count instructions per loop of one kernel in the hipcc -S output (tools/loop_stats.py file.s mangled-name-regex)."""
import re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
m = re.search(r'^(' + pat + r'):.*?^\s*s_endpgm', s, re.S | re.M)
lines = m.group(0).split('\n')
labels = {}
for i, l in enumerate(lines):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm:
        labels[mm.group(1)] = i
loops = []
for i, l in enumerate(lines):
    mm = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
    if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
        loops.append((labels[mm.group(1)], i))
print("kernel lines", len(lines))
for a, b in sorted(set(loops)):
    ops = [x.strip().split()[0] for x in lines[a:b + 1] if x.strip() and not x.strip().startswith(('.', ';'))]
    v = sum(o.startswith('v_') for o in ops); sa = sum(o.startswith('s_') for o in ops)
    g = sum(o.startswith(('global_', 'buffer_', 'flat_')) for o in ops)
    print(f"loop {a}-{b}: insts={len(ops)} valu={v} salu={sa} vmem={g}")
if len(sys.argv) > 3:
    a, b = map(int, sys.argv[3].split('-'))
    for l in lines[a:b + 1]:
        t = l.strip()
        if t and not t.startswith((';', '.loc', '.cfi')):
            print(t)
