"""moonrtx_amd -- MI355X (gfx950) renderer backend for MoonRTX's displaced-sphere Moon scene.

Layout (only what the hot path needs):
  csrc/         HIP kernels + the C ABI of include/moonrt.h  -> libmoonrt.so (in-tree)
  _lib.py       ctypes loader (raises if the library is missing: there is no CPU fallback)
  renderer.py   MoonRT, object wrapper over the C ABI
  tkoptix.py    the PlotOptiX-named surface moon_renderer.py drives (`self.rt`)
  scene.py      headless restatement of the scene MoonRenderer pushes (camera, light, Sun disk)
  dist.py       image-tile sharding across GPUs + RCCL gather of the framebuffer
  build.py      hipcc build recipe
"""
__version__ = "0.1.0"
