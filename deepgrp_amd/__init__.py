"""deepgrp_amd -- MI355X (gfx950) implementation of DeepGRP's prediction hot path.

Python host code over hand-written HIP kernels (``libdeepgrp_hip.so``, C ABI in
``include/deepgrp_hip.h``).  The module names mirror the reference package
(``deepgrp.sequence``, ``deepgrp.mss``, ``deepgrp.prediction``, ``deepgrp.model``,
``deepgrp.__main__``) so that it is a drop-in for ``deepgrp predict``.

There is no CPU fallback: importing a compute function without the built
library, or calling it without a gfx950 device, raises.
"""
__version__ = "0.1.0"
