"""dual_gan model registry: `find_model_using_name('AE')` resolves `dual_gan/models/AE_model.py::AEModel` the way the
reference's CC/dual_gan/models/__init__.py does (module `<name>_model`, class `<name>Model` compared case-insensitively,
must derive from BaseModel), `get_option_setter` returns its `modify_options`."""
from __future__ import absolute_import

from rg_hip.overlay import extend as _rg_extend
_rg_extend(globals(), run_init=False)       # see rg_hip/overlay.py: the reference tree may sit behind this one on sys.path

from importlib import import_module  # noqa: E402

from .base_model import BaseModel  # noqa: E402


def find_model_using_name(model_name):
    module_name = "dual_gan.models.%s_model" % model_name
    wanted = (model_name + "model").lower()
    candidates = [obj for key, obj in vars(import_module(module_name)).items()
                  if key.lower() == wanted and isinstance(obj, type) and issubclass(obj, BaseModel)]
    if not candidates:
        raise ImportError("In %s.py, there should be a subclass of BaseModel with class name that matches %s in "
                          "lowercase." % (module_name, model_name))
    return candidates[-1]


def get_option_setter(model_name):
    return find_model_using_name(model_name).modify_options
