from __future__ import absolute_import

from rg_hip.overlay import extend as _rg_extend  # noqa: E402
_rg_extend(globals(), run_init=True)       # see rg_hip/overlay.py: the reference tree may sit behind this one on sys.path
from .device_pipeline import PoseMapGenerator, flip_images  # noqa: F401,E402
