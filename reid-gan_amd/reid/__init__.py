"""Drop-in `reid` package (model side of FD-GAN-master/reid): same factories, classes and state_dict keys,
executed by the MI355X HIP kernels through rg_hip."""
from __future__ import absolute_import

from rg_hip.overlay import extend as _rg_extend  # noqa: E402
_rg_extend(globals(), run_init=True)       # see rg_hip/overlay.py: the reference tree may sit behind this one on sys.path
from . import models  # noqa: F401,E402
