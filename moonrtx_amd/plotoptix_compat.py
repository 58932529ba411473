"""Register this backend under the module names MoonRTX imports from PlotOptiX.

    import moonrtx_amd.plotoptix_compat as pc; pc.install()
    # from here on the reference's own statements resolve to this backend, unmodified:
    #   import plotoptix; from plotoptix import TkOptiX              (moon_renderer.py:10-11)
    #   from plotoptix.materials import m_diffuse, m_flat            (moon_renderer.py:12, renderer_labels.py:12)
    #   from plotoptix.utils import read_image, get_gpu_architecture  (data_loader.py:10, main.py:16)
    #   from plotoptix.enums import GpuArchitecture                   (main.py:17)
    #   from plotoptix.install import download_file_from_google_drive (main.py:18)

Only the names the reference uses exist; everything else about PlotOptiX is deliberately absent.
"""
import enum
import sys
import types

from . import materials as _materials
from . import tkoptix as _tkoptix


class GpuArchitecture(enum.Enum):
    """main.py:177-183 compares `.value >= Compute_75.value` (an RTX gate); gfx950 is reported above it."""
    Auto = 0
    Compute_50 = 500
    Compute_60 = 600
    Compute_70 = 700
    Compute_75 = 750
    Compute_80 = 800
    Compute_86 = 860
    Compute_90 = 900
    CDNA4_gfx950 = 9500


def get_gpu_architecture(rt=None):
    """`plotoptix.utils.get_gpu_architecture` (main.py:179): the native library must load, i.e. be built for gfx950."""
    from . import _lib
    _lib.load()
    return GpuArchitecture.CDNA4_gfx950


def download_file_from_google_drive(file_id, destination):
    """main.py:168 fetches the default colour map this way; this backend never touches the network."""
    raise RuntimeError(f"no network access in this backend: place the file at {destination} yourself")


def install():
    """Create `plotoptix`, `plotoptix.materials`, `.utils`, `.enums`, `.install` in sys.modules (idempotent)."""
    if "plotoptix" in sys.modules and getattr(sys.modules["plotoptix"], "__moonrtx_amd__", False):
        return sys.modules["plotoptix"]
    from . import ingest
    root = types.ModuleType("plotoptix")
    root.__moonrtx_amd__ = True
    root.__version__ = _tkoptix.__version__
    root.TkOptiX = _tkoptix.TkOptiX
    root.NpOptiX = _tkoptix.TkOptiX
    mats = types.ModuleType("plotoptix.materials")
    mats.m_diffuse, mats.m_flat = _materials.m_diffuse, _materials.m_flat
    utils = types.ModuleType("plotoptix.utils")
    utils.read_image = ingest.read_image
    utils.get_gpu_architecture = get_gpu_architecture
    enums = types.ModuleType("plotoptix.enums")
    enums.GpuArchitecture = GpuArchitecture
    inst = types.ModuleType("plotoptix.install")
    inst.download_file_from_google_drive = download_file_from_google_drive
    for name, mod in (("materials", mats), ("utils", utils), ("enums", enums), ("install", inst)):
        setattr(root, name, mod)
        sys.modules["plotoptix." + name] = mod
    sys.modules["plotoptix"] = root
    return root
