"""Drop-in `reid` package (model side of FD-GAN-master/reid): same factories, classes and state_dict keys,
executed by the MI355X HIP kernels through rg_hip."""
from __future__ import absolute_import

from . import models  # noqa: F401
