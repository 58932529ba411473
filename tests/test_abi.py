"""The C-ABI library loads without a GPU and exports every symbol include/moonrt.h declares."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "moonrt.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mrtx_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(native_lib):
    from moonrtx_amd import _lib
    names = header_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(native_lib, n), f"libmoonrt.so does not export {n}"
    assert set(names) == set(_lib.SIGNATURES), set(names) ^ set(_lib.SIGNATURES)


def test_abi_version_and_struct_layouts(native_lib):
    from moonrtx_amd import _lib
    assert native_lib.mrtx_abi_version() == 7
    assert C.sizeof(_lib.MrtxConfig) == 28
    assert C.sizeof(_lib.MrtxParams) == 56
    assert C.sizeof(_lib.MrtxStats) == 152
    p = _lib.MrtxParams()
    native_lib.mrtx_default_params(C.byref(p))
    # moon_renderer.py:99-101, :583, :598, :130
    assert (round(p.scene_epsilon, 7), round(p.marching_step, 6), round(p.marching_step_eps, 7)) == (1e-4, 5e-3, 3e-4)
    assert (p.path_seg_min, p.path_seg_max, p.spp_per_launch, p.max_spp) == (2, 4, 64, 64)
    assert abs(p.tonemap_exposure - 0.9) < 1e-7


def test_argument_validation_needs_no_gpu(native_lib):
    from moonrtx_amd import _lib
    ctx = C.c_void_p()
    bad = _lib.MrtxConfig(0, 0, 10, 0, 1, 0, 0)
    assert native_lib.mrtx_create(C.byref(bad), C.byref(ctx)) == -1 and not ctx.value
    bad = _lib.MrtxConfig(0, 16, 16, 2, 2, 0, 0)
    assert native_lib.mrtx_create(C.byref(bad), C.byref(ctx)) == -1
    bad = _lib.MrtxConfig(0, 16, 16, 0, 1, 24, 16)     # tile not a multiple of 16
    assert native_lib.mrtx_create(C.byref(bad), C.byref(ctx)) == -1
    assert native_lib.mrtx_render(None, 1, None) == -1
    assert native_lib.mrtx_last_error(None) == b"null context"


def test_product_never_imports_the_oracle():
    """The shipped package must not route through the CPU oracle."""
    pkg = os.path.join(ROOT, "moonrtx_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f
                assert "liborc" not in src and "mrtx_oracle" not in src, f


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from moonrtx_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.NativeLibraryError, match="no CPU fallback"):
        _lib.load()


def test_runtime_preload_checks_the_soname(tmp_path):
    """_lib preloads the torch wheel's HIP runtime only if its SONAME is one libmoonrt.so asks for (DT_NEEDED): read from the
    ELF files themselves."""
    from moonrtx_amd import _lib, build
    build.build_native()
    soname, needed = _lib._elf_dynamic_strings(_lib.LIB_PATH)
    assert soname is None or soname.startswith("libmoonrt")
    hip = [n for n in needed if n.startswith("libamdhip64.so")]
    assert len(hip) == 1 and "libstdc++.so.6" in needed
    import importlib.util, os
    spec = importlib.util.find_spec("torch")
    wheel = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.isfile(wheel):
        assert _lib._elf_dynamic_strings(wheel)[0] == hip[0]        # this image: the wheel's copy may stand in
    bad = tmp_path / "x.so"
    bad.write_bytes(b"not an elf file")
    with pytest.raises(ValueError):
        _lib._elf_dynamic_strings(str(bad))
