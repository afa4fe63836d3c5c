"""reid-gan_amd: MI355X-native implementation of the ReID-GAN training step.

The directory name is not a Python identifier; add it to sys.path (tests/conftest.py, bench.py and
__graft_entry__.py do) so that the drop-in top-level packages resolve exactly as in the reference:

    import fdgan, reid, clustercontrast      # reference-compatible module trees
    import rg_hip                            # the HIP kernel library binding + tape runtime
"""
import os as _os
import sys as _sys

_here = _os.path.dirname(_os.path.abspath(__file__))
if _here not in _sys.path:
    _sys.path.insert(0, _here)
