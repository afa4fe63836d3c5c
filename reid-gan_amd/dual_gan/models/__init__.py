"""This package contains modules related to function, network architectures, and models
(mirror of CC/dual_gan/models/__init__.py)."""

from rg_hip.overlay import extend as _rg_extend  # noqa: E402
_rg_extend(globals(), run_init=False)       # see rg_hip/overlay.py: the reference tree may sit behind this one on sys.path

import importlib

from .base_model import BaseModel


def find_model_using_name(model_name):
    """Import the module "model/[model_name]_model.py"."""
    model_file_name = "dual_gan.models." + model_name + "_model"
    modellib = importlib.import_module(model_file_name)
    model = None
    for name, cls in modellib.__dict__.items():
        if name.lower() == (model_name + 'model').lower() and isinstance(cls, type) and issubclass(cls, BaseModel):
            model = cls
    if model is None:
        raise ImportError("In %s.py, there should be a subclass of BaseModel with class name that matches %s in "
                          "lowercase." % (model_file_name, model_name))
    return model


def get_option_setter(model_name):
    """Return the static method <modify_commandline_options> of the model class."""
    model = find_model_using_name(model_name)
    return model.modify_options
