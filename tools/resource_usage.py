#!/usr/bin/env python3
"""Summarise `make -C moonrtx_amd/csrc asm` output (-Rpass-analysis=kernel-resource-usage): one line per kernel.
usage: python tools/resource_usage.py /tmp/asm.log [filter]"""
import re, subprocess, sys
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
rows = []
for blk in txt.split("remark: Function Name: ")[1:]:
    name = blk.split()[0]
    try:
        name = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        pass
    g = lambda k: (re.search(k + r":\s*(\d+)", blk) or [None, "?"])[1]
    rows.append((name, g("TotalSGPRs"), g("VGPRs"), g("AGPRs"), g(r"ScratchSize \[bytes/lane\]"), g("SGPRs Spill"), g("VGPRs Spill"),
                 g(r"Occupancy \[waves/SIMD\]"), g(r"LDS Size \[bytes/block\]")))
print(f"{'kernel':90s} sgpr vgpr agpr scratch sspill vspill occ lds")
for r in rows:
    if flt in r[0]:
        print(f"{r[0][:90]:90s} " + " ".join(f"{x:>5s}" for x in r[1:]))
