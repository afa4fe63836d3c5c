from __future__ import absolute_import

from .device_pipeline import PoseMapGenerator, flip_images  # noqa: F401
