"""nsgp-repre_amd: MI355X-native NSGP-RePRE hot path.

Python host layer (mirror of the reference's optimizer / runner-hook / RoI-head
interface) over hand-written gfx950 HIP kernels reached through the C ABI in
``include/nsgp_repre.h`` (``libnsgp_repre_hip.so``, loaded with ctypes).
Import as ``nsgp_repre_amd`` (see the alias package next to this directory).

There is no CPU or eager-PyTorch fallback for the kernels: every compute entry
point raises if the HIP library is missing or the tensors are not on a GPU.
"""
from . import registry  # noqa: F401
from ._lib import lib_path, load_library  # noqa: F401
from .optim import AdamNSCL, AdamWNSCL, SGDNSCL, SGDNSCLNA  # noqa: F401
from . import datasets, detectors, roi_heads, runner  # noqa: F401,E402

__version__ = "0.1.0"
