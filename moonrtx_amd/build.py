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
