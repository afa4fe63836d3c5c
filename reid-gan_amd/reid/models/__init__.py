"""Model registry — mirrors FD-GAN-master/reid/models/__init__.py:6-52 (names(), create())."""
from __future__ import absolute_import

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
