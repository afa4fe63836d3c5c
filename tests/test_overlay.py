"""this tree first on PYTHONPATH, the reference tree behind it.  A synthetic stand-in for the reference
tree (same package / module names, trivial bodies) checks that reference-only modules and the names defined in the
reference's package __init__ files stay importable, that this build's modules win where both exist, and that a module
implementing only the hot-path part inherits the rest and overrides what it implements.  Runs in a subprocess (clean
sys.modules, no GPU)."""
import os
import subprocess
import sys
import textwrap

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _write(root, rel, body):
    path = os.path.join(root, rel)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write(textwrap.dedent(body))


def test_reference_tree_behind_this_one(tmp_path):
    ref = str(tmp_path / "ref")
    _write(ref, "clustercontrast/__init__.py", """
        from . import datasets
        from . import models
        from . import evaluators
        __version__ = 'ref-0.1'
        """)
    _write(ref, "clustercontrast/datasets.py", "NAMES = ['market1501']\n")
    _write(ref, "clustercontrast/models/__init__.py", "raise RuntimeError('the reference models package must not be imported')\n")
    _write(ref, "clustercontrast/evaluators.py", """
        def extract_cnn_feature(model, inputs):
            return 'reference extract_cnn_feature'
        def extract_features(model, loader):
            return extract_cnn_feature(model, None)
        def pairwise_distance(features, query=None, gallery=None):
            return 'reference pairwise_distance'
        class Evaluator(object):
            def evaluate(self):
                return extract_features(None, None), pairwise_distance(None)
        """)
    _write(ref, "clustercontrast/utils/__init__.py", "def to_numpy(t):\n    return 'ref to_numpy'\n")
    _write(ref, "clustercontrast/utils/logging.py", "class Logger(object):\n    pass\n")
    _write(ref, "clustercontrast/utils/data/__init__.py", """
        from .sampler import RandomMultipleGallerySampler
        class IterLoader(object):
            pass
        """)
    _write(ref, "clustercontrast/utils/data/sampler.py", "class RandomMultipleGallerySampler(object):\n    pass\n")
    _write(ref, "clustercontrast/utils/data/pose_utils.py", "def draw_pose_from_cords():\n    return 'ref drawing helper'\n")
    _write(ref, "reid/__init__.py", "from . import models\nfrom . import datasets\n")
    _write(ref, "reid/datasets.py", "X = 1\n")
    _write(ref, "reid/models/__init__.py", "raise RuntimeError('the reference models package must not be imported')\n")
    _write(ref, "reid/utils/__init__.py", "def to_torch(x):\n    return 'ref to_torch'\n")
    _write(ref, "reid/utils/data/__init__.py", "from .preprocessor import Preprocessor\n")
    _write(ref, "reid/utils/data/preprocessor.py", "class Preprocessor(object):\n    pass\n")
    code = textwrap.dedent("""
        import sys
        import clustercontrast, reid
        from clustercontrast import datasets, models
        from clustercontrast.utils.logging import Logger
        from clustercontrast.utils.data import IterLoader, RandomMultipleGallerySampler
        from clustercontrast.utils.data.pose_utils import draw_pose_from_cords
        from clustercontrast.utils.data.device_pose import cords_to_map
        from clustercontrast.utils import to_numpy
        from clustercontrast.models.cm import ClusterMemory
        from clustercontrast.evaluators import Evaluator, extract_features, extract_cnn_feature, pairwise_distance
        from reid.utils.data import Preprocessor, PoseMapGenerator
        from reid.utils import to_torch
        import reid.datasets
        REPO = %r
        assert clustercontrast.__version__ == 'ref-0.1' and datasets.NAMES == ['market1501']
        assert models.__file__.startswith(REPO) and sys.modules['clustercontrast.models.cm'].__file__.startswith(REPO)
        assert extract_cnn_feature.__module__ == 'clustercontrast.evaluators' and sys.modules['clustercontrast.evaluators'].__file__.startswith(REPO)
        assert extract_features.__module__ == 'clustercontrast._ref_evaluators'       # the host loop is the reference's own
        assert Evaluator.__module__ == 'clustercontrast._ref_evaluators'
        # the reference's Evaluator now runs this build's functions
        import clustercontrast._ref_evaluators as R
        assert R.extract_cnn_feature is extract_cnn_feature and R.pairwise_distance is pairwise_distance
        assert draw_pose_from_cords() == 'ref drawing helper' and to_numpy(0) == 'ref to_numpy' and to_torch(0) == 'ref to_torch'
        assert reid.models.__file__.startswith(REPO) and PoseMapGenerator.__module__ == 'reid.utils.data.device_pipeline'
        print('OVERLAY-OK')
        """ % os.path.join(REPO, "reid-gan_amd"))
    env = dict(os.environ)
    env["PYTHONPATH"] = os.pathsep.join([os.path.join(REPO, "reid-gan_amd"), ref])
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OVERLAY-OK" in out.stdout, out.stdout + out.stderr


def test_without_a_reference_tree_nothing_changes():
    code = "import clustercontrast, reid, fdgan, dual_gan; from clustercontrast.evaluators import extract_cnn_feature, pairwise_distance; print('OK')"
    env = dict(os.environ)
    env["PYTHONPATH"] = os.path.join(REPO, "reid-gan_amd")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout + out.stderr
