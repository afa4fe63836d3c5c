"""Model registry — mirrors FD-GAN-master/reid/models/__init__.py:6-52 (names(), create())."""
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
    """create(name, pretrained=True, cut_at_pooling=False, num_features=0, norm=False, dropout=0, num_classes=0)
    — same arguments and error as the reference (FD/reid/models/__init__.py:19-52)."""
    if name not in __factory:
        raise KeyError("Unknown model:", name)
    return __factory[name](*args, **kwargs)
