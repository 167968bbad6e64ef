"""fplll_amd — MI355X-native (gfx950, HIP) implementation of fplll's GSO / SVP-enumeration hot path.

Only what the path needs lives here: ``csrc/`` (HIP kernels + the C ABI of include/fplll_hip.h +
the host C++ adapter for fplll's external-enumerator hook) and a thin Python mirror of the
reference interface used by tests and bench.  See DESIGN.md / INTEGRATION.md.
"""
from ._lib import Context, HipError, load, LIB_PATH  # noqa: F401
from .enumeration import (  # noqa: F401
    EVALSTRATEGY_BEST_N_SOLUTIONS, EVALSTRATEGY_FIRST_N_SOLUTIONS,
    EVALSTRATEGY_OPPORTUNISTIC_N_SOLUTIONS, FastEvaluator, Unsupported, enumerate_block)
