#!/usr/bin/env python3
"""Attribute the static VALU/SALU/VMEM instructions of one kernel to source functions via -gline-tables-only .loc lines.
usage: tools/asm_attrib.py file.s mangled_kernel_name source.hip [--lines]"""
import re, sys, collections
asm, kern, src = sys.argv[1:4]
per_line = "--lines" in sys.argv
# function start lines of the source
funcs = []
for i, l in enumerate(open(src), 1):
    m = re.match(r'^(?:__device__|__global__|static|template|extern).*?\b(\w+)\s*\(', l)
    if m and not l.startswith('template'):
        funcs.append((i, m.group(1)))
    elif re.match(r'^__global__ void.*\b(\w+)\(', l):
        funcs.append((i, re.match(r'^__global__ void.*\b(\w+)\(', l).group(1)))
def func_of(line):
    name = "?"
    for s, n in funcs:
        if s <= line: name = n
        else: break
    return name
text = open(asm).read()
m = re.search(r'^' + re.escape(kern) + r':.*?s_endpgm', text, re.S | re.M)
cur = 0
cnt = collections.Counter(); lcnt = collections.Counter()
for l in m.group(0).split('\n'):
    t = l.strip()
    mm = re.match(r'\.loc\s+(\d+)\s+(\d+)', t)
    if mm:
        if mm.group(1) == '0': cur = int(mm.group(2))
        continue
    if not t or t.startswith(('.', ';')) or t.endswith(':'): continue
    op = t.split()[0]
    kind = 'valu' if op.startswith('v_') else 'salu' if op.startswith('s_') else 'vmem' if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'other'
    cnt[(func_of(cur), kind)] += 1
    if kind == 'valu': lcnt[cur] += 1
names = sorted({f for f, _ in cnt}, key=lambda f: -cnt[(f, 'valu')])
print(f"{'function':28s} valu salu vmem")
for f in names:
    print(f"{f:28s} {cnt[(f,'valu')]:5d} {cnt[(f,'salu')]:4d} {cnt[(f,'vmem')]:4d}")
print("total valu", sum(v for (f, k), v in cnt.items() if k == 'valu'))
if per_line:
    sl = open(src).read().split('\n')
    for ln, c in sorted(lcnt.items()):
        print(f"{ln:5d} {c:4d}  {sl[ln-1].strip()[:110] if ln else ''}")
