"""ctypes loader for libmoonrt.so (the C ABI of include/moonrt.h).

The library is built in-tree by `moonrtx_amd.build.build_native()` / `make -C moonrtx_amd/csrc`.
There is NO CPU fallback: if the shared object is missing or a symbol is absent this module raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MOONRT_LIB") or os.path.join(_HERE, "libmoonrt.so")   # MOONRT_LIB: A/B builds only

ABI_VERSION = 7


class MrtxConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("rank", C.c_int32), ("world", C.c_int32), ("tile_w", C.c_int32), ("tile_h", C.c_int32)]


class MrtxParams(C.Structure):
    _fields_ = [("scene_epsilon", C.c_float), ("marching_step", C.c_float), ("marching_step_eps", C.c_float),
                ("tonemap_exposure", C.c_float), ("tonemap_gamma", C.c_float),
                ("path_seg_min", C.c_uint32), ("path_seg_max", C.c_uint32),
                ("spp_per_launch", C.c_uint32), ("max_spp", C.c_uint32), ("seed", C.c_uint32),
                ("const_albedo", C.c_float * 3), ("flags", C.c_uint32)]


class MrtxStats(C.Structure):
    _fields_ = [("primary_rays", C.c_uint64), ("primary_hits", C.c_uint64), ("shadow_rays", C.c_uint64),
                ("height_samples", C.c_uint64), ("colour_fetches", C.c_uint64),
                ("background_fetches", C.c_uint64), ("dem_fetches", C.c_uint64), ("mip_fetches", C.c_uint64), ("bounce_rays", C.c_uint64),
                ("bounce_sun_hits", C.c_uint64),
                ("kernel_ms", C.c_double), ("primary_ms", C.c_double), ("paths_ms", C.c_double),
                ("launches", C.c_uint32), ("reserved", C.c_uint32),
                # ABI 7: render_kernel's own share of the counters (the rest is path_kernel's)
                ("camera_height_samples", C.c_uint64), ("camera_dem_fetches", C.c_uint64), ("camera_mip_fetches", C.c_uint64),
                ("camera_colour_fetches", C.c_uint64), ("camera_background_fetches", C.c_uint64)]


F_COUNT_STATS = 1
F_FORCE_WIDE = 2
F_NO_SKIP = 4
F_NO_CULL = 8
F_NO_SORT = 16
F_INWAVE_PATHS = 32
BUF_ACCUM, BUF_HITS, BUF_DEM, BUF_COLOR = 0, 1, 2, 3

_D3 = C.POINTER(C.c_double)
_VP = C.c_void_p

# name -> (restype, argtypes): every symbol include/moonrt.h declares
SIGNATURES = {
    "mrtx_abi_version": (C.c_int, []),
    "mrtx_get_config": (C.c_int, [_VP, C.POINTER(MrtxConfig)]),
    "mrtx_set_gather_hits": (C.c_int, [_VP, C.c_int32]),
    "mrtx_create": (C.c_int, [C.POINTER(MrtxConfig), C.POINTER(_VP)]),
    "mrtx_destroy": (None, [_VP]),
    "mrtx_last_error": (C.c_char_p, [_VP]),
    "mrtx_upload_dem": (C.c_int, [_VP, _VP, C.c_int32, C.c_int32]),
    "mrtx_bind_dem_device": (C.c_int, [_VP, _VP, C.c_int32, C.c_int32]),
    "mrtx_upload_color": (C.c_int, [_VP, _VP, C.c_int32, C.c_int32]),
    "mrtx_bind_color_device": (C.c_int, [_VP, _VP, C.c_int32, C.c_int32]),
    "mrtx_upload_background": (C.c_int, [_VP, _VP, C.c_int32, C.c_int32]),
    "mrtx_upload_overlay": (C.c_int, [_VP, _VP, C.c_int32, C.c_int32]),
    "mrtx_set_params": (C.c_int, [_VP, C.POINTER(MrtxParams)]),
    "mrtx_default_params": (None, [C.POINTER(MrtxParams)]),
    "mrtx_set_camera": (C.c_int, [_VP, _D3, _D3, _D3, C.c_double]),
    "mrtx_set_moon_frame": (C.c_int, [_VP, _D3, C.c_double, _D3, _D3]),
    "mrtx_set_light": (C.c_int, [_VP, _D3, C.c_double, C.c_double]),
    "mrtx_set_sun_disk": (C.c_int, [_VP, _D3, C.c_double, C.c_double]),
    "mrtx_set_capsules": (C.c_int, [_VP, _VP, C.c_int32]),
    "mrtx_reset_accum": (C.c_int, [_VP]),
    "mrtx_render": (C.c_int, [_VP, C.c_int32, C.POINTER(MrtxStats)]),
    "mrtx_render_part": (C.c_int, [_VP, C.c_int32, C.c_int32, C.c_int32, C.POINTER(MrtxStats)]),
    "mrtx_read_linear": (C.c_int, [_VP, _VP]),
    "mrtx_read_rgba8": (C.c_int, [_VP, _VP]),
    "mrtx_read_rgb16": (C.c_int, [_VP, _VP]),
    "mrtx_read_hits": (C.c_int, [_VP, _VP]),
    "mrtx_read_hit": (C.c_int, [_VP, C.c_int32, C.c_int32, C.POINTER(C.c_float)]),
    "mrtx_samples_done": (C.c_int, [_VP, C.POINTER(C.c_uint32)]),
    "mrtx_shard_bytes": (C.c_int, [_VP, C.c_int32, C.POINTER(C.c_uint64)]),
    "mrtx_shard_bytes_active": (C.c_int, [_VP, C.POINTER(C.c_uint64)]),
    "mrtx_shard_parts": (C.c_int, [_VP, C.c_int32, C.POINTER(C.c_int32)]),
    "mrtx_pack_part": (C.c_int, [_VP, _VP, C.c_int32, C.c_int32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), _VP]),
    "mrtx_pack_shard": (C.c_int, [_VP, _VP, _VP]),
    "mrtx_unpack_shard": (C.c_int, [_VP, C.c_int32, _VP, _VP]),
    "mrtx_unpack_all": (C.c_int, [_VP, C.POINTER(_VP), C.c_int32]),
    "mrtx_device_ptr": (C.c_int, [_VP, C.c_int32, C.POINTER(_VP), C.POINTER(C.c_uint64)]),
    "mrtx_dem_from_ldem": (C.c_int, [C.c_int32, _VP, C.c_int32, C.c_int32, C.c_int32, _VP,
                                     C.POINTER(C.c_float), C.c_char_p, C.c_int32]),
    "mrtx_synth_ldem": (C.c_int, [C.c_int32, _VP, C.c_int32, C.c_int32, C.c_uint32, C.c_char_p, C.c_int32]),
    "mrtx_synth_color": (C.c_int, [C.c_int32, _VP, C.c_int32, C.c_int32, C.c_uint32, C.c_char_p, C.c_int32]),
    "mrtx_dev_alloc": (C.c_int, [C.c_int32, C.c_uint64, C.POINTER(_VP)]),
    "mrtx_dev_free": (C.c_int, [C.c_int32, _VP]),
    "mrtx_dev_download": (C.c_int, [C.c_int32, _VP, _VP, C.c_uint64]),
    "mrtx_dev_upload": (C.c_int, [C.c_int32, _VP, _VP, C.c_uint64]),
    "mrtx_probe_stream": (C.c_int, [C.c_int32, C.c_uint64, C.c_int32]),
    "mrtx_probe_cr": (C.c_int, [C.c_int32, C.c_int32, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32)]),
    "mrtx_probe_latlon": (C.c_int, [C.c_int32, _VP, _VP, _VP, _VP, _VP, C.c_int32]),
}

_lib = None
_preloaded_runtime = None


class NativeLibraryError(RuntimeError):
    pass


def hip_runtimes_loaded():
    """Paths of every libamdhip64 copy mapped into this process (more than one means two HIP/HSA runtimes are alive:
    the second one to initialise finds no GPUs)."""
    seen = []
    try:
        with open("/proc/self/maps") as f:
            for line in f:
                path = line.rsplit(" ", 1)[-1].strip()
                if "libamdhip64" in os.path.basename(path):
                    real = os.path.realpath(path)
                    if real not in seen:
                        seen.append(real)
    except OSError:
        pass
    return seen


def _elf_dynamic_strings(path):
    """(DT_SONAME or None, [DT_NEEDED ...]) of a little-endian ELF64 shared object, read from the file (no loader involved)."""
    import struct
    with open(path, "rb") as f:
        data = f.read()
    if data[:6] != b"\x7fELF\x02\x01":
        raise ValueError(f"{path}: not a little-endian ELF64 file")
    shoff, = struct.unpack_from("<Q", data, 0x28)
    shentsize, shnum = struct.unpack_from("<HH", data, 0x3A)
    secs = [struct.unpack_from("<IIQQQQIIQQ", data, shoff + i * shentsize) for i in range(shnum)]
    soname, needed = None, []
    for sec in secs:
        if sec[1] != 6:                       # SHT_DYNAMIC
            continue
        stroff = secs[sec[6]][4]              # sh_link -> the string table's file offset
        for i in range(sec[5] // 16):
            tag, val = struct.unpack_from("<qQ", data, sec[4] + 16 * i)
            if tag == 0:
                break
            if tag in (1, 14):                # DT_NEEDED, DT_SONAME
                end = data.index(b"\0", stroff + val)
                name = data[stroff + val:end].decode()
                if tag == 14:
                    soname = name
                else:
                    needed.append(name)
    return soname, needed


def _preload_torch_hip_runtime():
    """ONE HIP runtime per process, whatever the import order.

    libmoonrt.so needs `libamdhip64.so.7` (a SONAME); the torch wheel bundles its own copy of the runtime and asks for it by
    FILE name (`libamdhip64.so`, RPATH $ORIGIN).  The dynamic loader reuses an already-loaded library when the requested name
    equals its SONAME or its file -- so "torch first" gives one runtime (torch's copy satisfies libmoonrt's SONAME request),
    but "libmoonrt first" used to give two (/opt/rocm's for libmoonrt, then torch's for torch), and torch / RCCL then
    reported "no GPUs found".  When a torch installation is present its copy is therefore loaded here, BEFORE
    libmoonrt.so and without importing torch: either order now ends on that single copy.  Without torch the system
    runtime is used.  MOONRT_HIP_RUNTIME=system skips this (then import torch before this package)."""
    global _preloaded_runtime
    if os.environ.get("MOONRT_HIP_RUNTIME") == "system" or hip_runtimes_loaded():
        return
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.isfile(path):
        # Only a copy that IS what libmoonrt asks for may stand in for it: the wheel's SONAME must be one of libmoonrt's
        # DT_NEEDED names.  Otherwise the loader would pull the system runtime in beside it (two runtimes), or libmoonrt would
        # silently run on a runtime of another major version: leave the system runtime alone then (round-2 advisor finding).
        try:
            soname, _ = _elf_dynamic_strings(path)
            _, needed = _elf_dynamic_strings(LIB_PATH)
        except (OSError, ValueError, IndexError, __import__("struct").error):
            return
        if soname is None or soname not in needed:
            return
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
            _preloaded_runtime = path
        except OSError:
            pass            # fall back to the system runtime; load() checks for duplicates below


def load():
    """Return the loaded library; raise NativeLibraryError if it is missing or incomplete."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise NativeLibraryError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C moonrtx_amd/csrc` (needs hipcc). There is no CPU fallback.")
    _preload_torch_hip_runtime()
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as e:
        raise NativeLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise NativeLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.mrtx_abi_version() != ABI_VERSION:
        raise NativeLibraryError(f"ABI version mismatch: library {lib.mrtx_abi_version()}, host {ABI_VERSION}")
    rts = hip_runtimes_loaded()
    if len(rts) > 1:
        raise NativeLibraryError("two HIP runtimes are loaded in this process (" + ", ".join(rts) + "): the second one to "
                                 "initialise will see no GPUs. Import moonrtx_amd (or torch) before anything else loads "
                                 "another libamdhip64, or set LD_LIBRARY_PATH so that both resolve to one copy.")
    _lib = lib
    return lib


def assert_single_hip_runtime():
    """Called before a torch.distributed / RCCL group is created: fail loudly instead of 'no GPUs found'."""
    rts = hip_runtimes_loaded()
    if len(rts) > 1:
        raise NativeLibraryError("two HIP runtimes are loaded in this process: " + ", ".join(rts))
    return rts


def vec3(v):
    a = (C.c_double * 3)(float(v[0]), float(v[1]), float(v[2]))
    return a
