"""fvgp_amd -- MI355X-native exact-GP engine behind the fvgp.GP train / log_likelihood /
posterior API.  The compute path is libfvgp_hip.so (hand-written gfx950 kernels, C ABI in
include/fvgp_hip.h); this package is the Python mirror of the reference's interface."""
__version__ = "0.1.0"

from .gp import GP                      # noqa: E402,F401
from .fvgp import fvGP                  # noqa: E402,F401
from .gp_lin_alg import NonPositiveDefiniteError   # noqa: E402,F401
from . import kernels                   # noqa: E402,F401
from .gp_training import ProposalDistribution   # noqa: E402,F401
