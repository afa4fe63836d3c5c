"""Drop-in `clustercontrast` package (model + trainer side of cluster-contrast-reid-main/clustercontrast) on the
MI355X HIP kernels."""

from rg_hip.overlay import extend as _rg_extend  # noqa: E402
_rg_extend(globals(), run_init=True)       # see rg_hip/overlay.py: the reference tree may sit behind this one on sys.path
