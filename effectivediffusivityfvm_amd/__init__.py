"""effectivediffusivityfvm_amd -- MI355X-native hot path of adama-wzr/EffectiveDiffusivityFVM.

The product is libdeff_amd.so (HIP kernels + C ABI, see include/deff_amd.h and
csrc/); this package is the thin Python host layer used by tests and bench.py.
Importing it never touches the GPU; creating a Solver does and fails loudly if
the library or a device is missing.
"""
from ._capi import DeffError, KERNEL_NAMES, LIB_PATH  # noqa: F401
from .solver import (OMEGA_REFERENCE, SlabGroup, SlabRank, Solver, SolveResult, TorchDistTransport, flood_fill,  # noqa: F401
                     recommended_batch, load_jpeg_gray, rccl_unique_id)
