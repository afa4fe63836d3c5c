"""`reid.feature_extraction` — eval-mode feature extraction on the MI355X (FD-GAN-master/reid/feature_extraction/)."""
from __future__ import absolute_import

from rg_hip.overlay import extend as _rg_extend
_rg_extend(globals(), run_init=False)

from .cnn import extract_cnn_feature  # noqa: E402,F401

__all__ = ['extract_cnn_feature']
