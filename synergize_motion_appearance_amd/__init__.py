"""MI355X-native per-frame talking-head reenactment hot path.

Drop-in for the `basicsr.archs` plugin surface of ShaelynZ/synergize-motion-appearance
(ARCH_REGISTRY names, options/*.yml kwargs, checkpoint key layout) whose forward passes
run as hand-written HIP kernels for gfx950 behind a C-ABI shared library
(`include/smx.h`, `csrc/`).  See DESIGN.md.
"""
__version__ = "0.1.0"
