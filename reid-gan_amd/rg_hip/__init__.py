"""rg_hip — host runtime of the MI355X ReID-GAN kernels: C-ABI binding (lib), tensor-level ops (ops),
tape-based module runtime (tape, nn), fused optimizers (optim) and the data-parallel reducer (parallel)."""
from __future__ import absolute_import

from .lib import build, LIB_PATH, FAMILIES  # noqa: F401
