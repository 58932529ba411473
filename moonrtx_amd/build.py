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


def _locked(fn):
    """Run fn() under an exclusive lock on the source directory: the ranks of one node (bench.py --gpus N starts N processes
    at once) must not run `make` on the same objects together -- the first one in builds, the others find the library fresh."""
    import fcntl
    fd = os.open(CSRC, os.O_RDONLY)
    try:
        try:
            fcntl.flock(fd, fcntl.LOCK_EX)
        except OSError:     # a file system without advisory locks: build unlocked, as before
            pass
        return fn()
    finally:
        os.close(fd)        # closing the descriptor releases the lock


def build_native(force=False):
    """Compile the HIP extension in-tree; returns the library path."""
    def go():
        if force or stale():
            subprocess.check_call(["make", "-C", CSRC, "-s", "all"])
    _locked(go)
    return LIB


SPILL_LIB = os.path.join(_HERE, "libmoonrt_spilltest.so")


def build_spilltest(force=False):
    """The TEST build of the heavily spilled in-wave counting kernel (see csrc/Makefile, tools/spill_repro.py); the product
    never loads it.  Built in-tree so that it travels to the GPU box with the snapshot."""
    def go():
        if force or not os.path.isfile(SPILL_LIB) or any(os.path.getmtime(os.path.join(CSRC, s)) > os.path.getmtime(SPILL_LIB) for s in SOURCES):
            subprocess.check_call(["make", "-C", CSRC, "-s", "spilltest"])
    _locked(go)
    return SPILL_LIB
