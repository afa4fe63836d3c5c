"""Package overlay: lets this tree sit FIRST on PYTHONPATH in front of the reference tree.

A regular package found first on sys.path hides every same-named package further down, so `from clustercontrast import
datasets` (reference-only code: data sets, samplers, loggers, visualisers ...) would fail once `clustercontrast` resolves
to this build.  Each package `__init__` here therefore calls `extend(globals())`:

  * `__path__` is extended with the same-named package directories found later on sys.path (`pkgutil.extend_path`), this
    build's directory staying first — sub-modules that exist here win, all others resolve to the reference's files;
  * `run_init=True` (packages whose `__init__` is an empty shell here) additionally executes the reference's
    `__init__.py` inside this package's namespace, so names the reference defines there (`IterLoader`, `to_torch`,
    `from . import datasets` ...) exist as its scripts expect.  A failure in that foreign `__init__` (a dependency missing
    on this machine) is reported as a warning and does not take the hot path down.

With no reference tree on sys.path both steps are no-ops.  RG_OVERLAY=0 disables the mechanism.
"""
from __future__ import absolute_import

import os
import pkgutil
import warnings


def extend(pkg_globals, run_init=False):
    if os.environ.get("RG_OVERLAY", "1") == "0":
        return
    name, own = pkg_globals["__name__"], list(pkg_globals["__path__"])
    full = pkgutil.extend_path(list(own), name)
    extra = [p for p in full if p not in own]
    if not extra:
        return
    pkg_globals["__path__"][:] = own + extra          # in place: the import system holds a reference to this list
    if not run_init:
        return
    for p in extra:
        init = os.path.join(p, "__init__.py")
        if not os.path.isfile(init):
            continue
        try:
            with open(init, "r") as f:
                code = compile(f.read(), init, "exec")
            exec(code, pkg_globals)
        except Exception as e:      # noqa: BLE001 - a foreign __init__ may need packages this machine lacks
            warnings.warn("overlay: %s of the reference tree could not be initialised (%s: %s); "
                          "its names are unavailable, this build's modules are unaffected" % (init, type(e).__name__, e))


def inherit(mod_globals):
    """For a MODULE of this build that implements only the hot-path part of the reference's module of the same name
    (e.g. clustercontrast/evaluators.py: the feature extraction and distance functions, not `Evaluator` / CMC / mAP):
    load the reference's file from the extended package path under a private name, copy the names this module does not
    define, and point the reference module's own references to the overridden functions at this build's versions — so
    `Evaluator.evaluate` (reference code) runs `extract_features` / `pairwise_distance` from here."""
    if os.environ.get("RG_OVERLAY", "1") == "0":
        return
    import importlib.util
    import sys
    name, own_file = mod_globals["__name__"], os.path.abspath(mod_globals["__file__"])
    pkg, _, leaf = name.rpartition(".")
    pkgmod = sys.modules.get(pkg)
    if pkgmod is None:
        return
    for d in list(getattr(pkgmod, "__path__", [])):
        cand = os.path.join(d, leaf + ".py")
        if not os.path.isfile(cand) or os.path.abspath(cand) == own_file:
            continue
        ref_name = "%s._ref_%s" % (pkg, leaf)
        try:
            spec = importlib.util.spec_from_file_location(ref_name, cand)
            ref = importlib.util.module_from_spec(spec)
            ref.__package__ = pkg
            sys.modules[ref_name] = ref
            spec.loader.exec_module(ref)
        except Exception as e:      # noqa: BLE001
            sys.modules.pop(ref_name, None)
            warnings.warn("overlay: %s of the reference tree could not be loaded (%s: %s); only this build's part of "
                          "%s is available" % (cand, type(e).__name__, e, name))
            return
        own_public = {k: v for k, v in mod_globals.items() if not k.startswith("_") and callable(v)
                      and getattr(v, "__module__", None) == name}
        for k, v in vars(ref).items():
            if not k.startswith("__") and k not in mod_globals:
                mod_globals[k] = v
        for k, v in own_public.items():
            if hasattr(ref, k):
                setattr(ref, k, v)
        return
