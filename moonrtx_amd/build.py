"""Build recipe for libmoonrt.so (hipcc, gfx950 only)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB = os.path.join(_HERE, "libmoonrt.so")
SOURCES = ["mrtx_kernels.hip", "mrtx_api.hip", "mrtx_device.h", os.path.join("..", "..", "include", "moonrt.h")]


def stale():
    if not os.path.isfile(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, s)) > t for s in SOURCES)


def build_native(force=False):
    """Compile the HIP extension in-tree; returns the library path."""
    if force or stale():
        subprocess.check_call(["make", "-C", CSRC, "-s", "all"])
    return LIB


SPILL_LIB = os.path.join(_HERE, "libmoonrt_spilltest.so")


def build_spilltest(force=False):
    """The TEST build of the heavily spilled in-wave counting kernel (see csrc/Makefile, tools/spill_repro.py); the product
    never loads it.  Built in-tree so that it travels to the GPU box with the snapshot."""
    if force or not os.path.isfile(SPILL_LIB) or any(os.path.getmtime(os.path.join(CSRC, s)) > os.path.getmtime(SPILL_LIB) for s in SOURCES):
        subprocess.check_call(["make", "-C", CSRC, "-s", "spilltest"])
    return SPILL_LIB
