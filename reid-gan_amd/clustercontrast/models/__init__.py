"""Model registry — mirrors CC/clustercontrast/models/__init__.py:6-59 for the ResNet family (the IBN / two-branch /
multi-part variants registered there are outside the hot path)."""
from __future__ import absolute_import

from rg_hip.overlay import extend as _rg_extend  # noqa: E402
_rg_extend(globals(), run_init=False)       # see rg_hip/overlay.py: the reference tree may sit behind this one on sys.path


from .resnet import *  # noqa: F401,F403
from .resnet import resnet18, resnet34, resnet50, resnet101, resnet152

__factory = {
    'resnet18': resnet18,
    'resnet34': resnet34,
    'resnet50': resnet50,
    'resnet101': resnet101,
    'resnet152': resnet152,
}


def names():
    return sorted(__factory.keys())


def create(name, *args, **kwargs):
    """create(name, pretrained=True, cut_at_pooling=False, num_features=0, norm=False, dropout=0, num_classes=0,
    pooling_type='avg') — CC/clustercontrast/models/__init__.py:26-59."""
    if name not in __factory:
        raise KeyError("Unknown model:", name)
    return __factory[name](*args, **kwargs)
