"""The `plotoptix` module names MoonRTX imports, provided by this backend (moonrtx_amd/plotoptix_compat.py)."""
import re
import sys

import numpy as np


def test_reference_import_statements_resolve(native_lib, tmp_path):
    from moonrtx_amd import plotoptix_compat
    for k in [k for k in sys.modules if k == "plotoptix" or k.startswith("plotoptix.")]:
        del sys.modules[k]
    plotoptix_compat.install()
    # the reference's own import lines (moon_renderer.py:10-12, renderer_labels.py:12, data_loader.py:10, main.py:15-18)
    import plotoptix
    from plotoptix import TkOptiX
    from plotoptix.materials import m_diffuse, m_flat
    from plotoptix.utils import read_image, get_gpu_architecture
    from plotoptix.enums import GpuArchitecture
    from plotoptix.install import download_file_from_google_drive
    # main.py:190-201 version gate, main.py:177-183 architecture gate
    m = re.match(r"(\d+)\.(\d+)\.(\d+)", plotoptix.__version__)
    assert tuple(int(g) for g in m.groups()) >= (0, 19, 2)
    arch = get_gpu_architecture()
    assert arch is not None and arch.value >= GpuArchitecture.Compute_75.value
    # moon_renderer.py:615-616 / renderer_labels.py:132-139 usage of the material dicts
    mat = m_diffuse.copy(); mat["ColorTextures"] = ["moon_color"]
    assert "ColorTextures" not in m_flat and m_diffuse["ColorTextures"] == []
    flat = dict(m_flat); flat["OcclusionProgram"] = "p"; flat["VarFloat4"] = {}
    assert TkOptiX.__name__ == "TkOptiX" and callable(read_image) and callable(download_file_from_google_drive)
    assert plotoptix_compat.install() is plotoptix     # idempotent
