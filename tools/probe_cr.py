#!/usr/bin/env python3
"""Exhaustive device check of the kernels' domain-restricted reciprocal / square root against the IEEE expansions (mrtx_probe_cr):
which exponent ranges are exact?  python tools/probe_cr.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from moonrtx_amd import _lib
lib = _lib.load()
names = {0: "rcp + 1 Newton step", 1: "rcp + 2 Newton steps", 2: "sqrt_cr"}
for which in (0, 1, 2):
    bad_exps = []
    total = 0
    for sign in ((0, 1) if which < 2 else (0,)):
        for e in range(1, 255):                      # every normal exponent, all 2^23 mantissas
            lo = (sign << 31) | (e << 23)
            n = C.c_uint64(); first = C.c_uint32()
            rc = lib.mrtx_probe_cr(0, which, lo, 1 << 23, C.byref(n), C.byref(first))
            assert rc == 0, rc
            total += n.value
            if n.value:
                bad_exps.append((sign, e - 127, n.value, hex(first.value)))
    n = C.c_uint64(); first = C.c_uint32()
    lib.mrtx_probe_cr(0, which, 0, 1, C.byref(n), C.byref(first))       # x = +0
    print(f"{names[which]}: {total} mismatches over all normal floats; x = +0 mismatch {n.value}; exponents (sign, e, count, first) with mismatches: "
          f"{bad_exps[:6]}{' ...' if len(bad_exps) > 6 else ''} [{len(bad_exps)} exponents: e from {min([b[1] for b in bad_exps], default=None)} to {max([b[1] for b in bad_exps], default=None)}]")
    if bad_exps:
        clean = [e for e in range(-126, 128) if not any(b[1] == e for b in bad_exps)]
        print(f"   clean exponent range: {min(clean)} .. {max(clean)}" if clean else "   no clean exponent")
