"""vaq_amd -- MI355X (gfx950) implementation of VAQ's quantized-distance search path.

Product surface:
  include/vaqhip.h            C ABI (the drop-in boundary)
  include/vaqhip.hpp          C++ adapter with the reference's names (class VaqHip)
  vaq_amd.VaqHip              Python mirror of the same interface, over the C ABI
There is no CPU path: importing works anywhere, but every compute call needs
the HIP library and a GPU and raises otherwise.
"""
from ._lib import VaqHipError, lib_path, load  # noqa: F401
from .index import LabelDistVec, NNMethod, VaqHip  # noqa: F401

__all__ = ["VaqHip", "NNMethod", "LabelDistVec", "VaqHipError", "load", "lib_path"]
