"""digdriver_amd -- MI355X-native implementation of DIGDriver's mutation-rate + burden-test
hot path (hand-written HIP kernels behind a C ABI; Python host mirroring the reference's
function names).  See DESIGN.md for scope and INTEGRATION.md for the drop-in boundary."""
__version__ = "0.1.0"

import os as _os

# MIOpen's default "find" runs an exhaustive search the first time it meets a convolution shape: ~1-3 s per layer and
# direction, again for every new batch size (the last, shorter batch of an epoch).  On a 40 000-bin k-fold training run
# that search was 45 of 52 s; the fast find mode picks kernels that run the steady-state step in the same 13.0 ms.
# MIOpen reads the variable when torch loads it, so it has to be in the environment before `import torch` (import this
# package first); a value set by the user wins.
_os.environ.setdefault("MIOPEN_FIND_MODE", "2")
