"""Run under `rocprofv3 --pmc FETCH_SIZE`: streams 4 GiB three times with 8-byte-per-lane loads (known bytes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from moonrtx_amd import _lib
lib = _lib.load()
BYTES = 4 << 30
assert lib.mrtx_probe_stream(0, BYTES, 3) == 0
print("streamed", BYTES, "bytes x 3")
