"""ctypes binding of libreidgan_hip.so (the C ABI declared in include/reidgan_hip.h).

The prototypes are parsed from the header itself, so the header is the single source of truth for
the boundary.  There is NO fallback: if the shared library is missing or a call fails, this module
raises — the product path never silently runs anything else.
"""
from __future__ import absolute_import

import ctypes
import os
import re
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
PKG_ROOT = os.path.dirname(_HERE)                      # reid-gan_amd/
REPO_ROOT = os.path.dirname(PKG_ROOT)
CSRC_DIR = os.path.join(PKG_ROOT, "csrc")
LIB_PATH = os.environ.get("RG_LIB_PATH") or os.path.join(PKG_ROOT, "lib", "libreidgan_hip.so")     # RG_LIB_PATH: A/B builds of the library
HEADER_PATH = os.path.join(REPO_ROOT, "include", "reidgan_hip.h")

_CTYPES = {
    "int": ctypes.c_int,
    "float": ctypes.c_float,
    "int64_t": ctypes.c_int64,
    "size_t": ctypes.c_size_t,
    "unsigned long long": ctypes.c_ulonglong,
    "rg_stream_t": ctypes.c_void_p,
    "void": None,
    "const char*": ctypes.c_char_p,
}


def _ctype_of(decl):
    decl = decl.strip()
    if decl.endswith("*"):
        if decl == "const char*":
            return ctypes.c_char_p
        return ctypes.c_void_p          # every device/host buffer crosses as an opaque address
    return _CTYPES[decl]


def parse_header(path=HEADER_PATH):
    """Returns {name: (restype_decl, [(type_decl, arg_name), ...])} for every prototype."""
    with open(path) as f:
        text = f.read()
    text = re.sub(r"/\*.*?\*/", " ", text, flags=re.S)
    text = re.sub(r"//[^\n]*", " ", text)
    text = re.sub(r"^\s*#.*$", " ", text, flags=re.M)
    protos = {}
    for m in re.finditer(r"([A-Za-z_][\w\s\*]*?)\b(rg_\w+)\s*\(([^;{}]*?)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if "typedef" in ret:
            continue
        ret = re.sub(r"\s*\*", "*", ret)
        parsed = []
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                mm = re.match(r"(.*?)(\w+)$", a)
                tdecl = re.sub(r"\s*\*\s*", "*", mm.group(1).strip())
                parsed.append((tdecl, mm.group(2)))
        protos[name] = (ret, parsed)
    return protos


def proto_hash(protos):
    """sha1 over the canonical prototype list; must match csrc/gen_pymod.py:proto_hash (the generated module embeds it)"""
    import hashlib
    h = hashlib.sha1()
    for name, (ret, args) in protos.items():
        h.update(("%s|%s|%s\n" % (name, ret, ",".join(t for t, _ in args))).encode())
    return h.hexdigest()


def build(verbose=False):
    """Compile every HIP source for gfx950 into reid-gan_amd/lib/libreidgan_hip.so (in-tree)."""
    cmd = ["make", "-C", CSRC_DIR, "-j", str(min(8, os.cpu_count() or 1))]
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, universal_newlines=True)
    if verbose or res.returncode != 0:
        print(res.stdout)
    if res.returncode != 0:
        raise RuntimeError("building libreidgan_hip.so failed")
    return LIB_PATH


_PLAIN_INT = {"rg_version", "rg_family_count", "rg_fold_chunk", "rg_bn_slices", "rg_conv2d_dgrad_rowsum_cols", "rg_f8_grad_tiles", "rg_krsc_chunk", "rg_conv_set_planes", "rg_conv_tune_stats", "rg_conv_splitk_inkernel_count"}   # int-returning queries that are not status codes


class _Lib(object):
    def __init__(self):
        self._dll = None
        self._native = None
        self.protos = parse_header()

    def load(self):
        if self._dll is not None:
            return self
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libreidgan_hip.so not found at %s — build it with `make -C %s` or __graft_entry__.build(); "
                "there is no CPU / PyTorch fallback for the HIP path" % (LIB_PATH, CSRC_DIR))
        # torch first: it brings its own HIP runtime (libamdhip64); loading this library before torch would bind the process
        # to the system copy instead and torch's device context would not be the one the kernels launch into
        import torch  # noqa: F401
        dll = ctypes.CDLL(LIB_PATH)
        for name, (ret, args) in self.protos.items():
            fn = getattr(dll, name)            # AttributeError if the header and library disagree
            fn.restype = _ctype_of(ret) if ret != "void" else None
            fn.argtypes = [_ctype_of(t) for t, _ in args]
        self._dll = dll
        self._native = self._load_native(dll)
        return self

    def _load_native(self, dll):
        """the generated CPython binding (csrc/gen_pymod.py): same symbols, direct argument conversion, GIL released around every
        launching entry point.  Default on (RG_NATIVE_BIND=0 selects ctypes): measured at step level on the same box with the
        4-crop batches where step time = host cost (profiles/r03_host_cost.txt): config 5 15.7 vs 17.0 ms, config 2 19.6 vs
        20.9 ms per step (round 2's version held the GIL across the calls and was SLOWER than ctypes).  Either way it is this
        library that runs.  A module generated from a different header (bind() matches names only) is refused: its PROTO_HASH
        must equal the hash of the header parsed here."""
        if os.environ.get("RG_NATIVE_BIND", "1") == "0":
            return None
        import glob
        import importlib.machinery
        import importlib.util
        cands = glob.glob(os.path.join(PKG_ROOT, "lib", "_rg_native*.so"))
        if not cands:
            return None
        try:
            loader = importlib.machinery.ExtensionFileLoader("_rg_native", cands[0])
            spec = importlib.util.spec_from_file_location("_rg_native", cands[0], loader=loader)
            mod = importlib.util.module_from_spec(spec)
            loader.exec_module(mod)
            want = proto_hash(self.protos)
            if getattr(mod, "PROTO_HASH", None) != want:
                raise ImportError("stale module: generated from other prototypes (%s != %s)" % (getattr(mod, "PROTO_HASH", None), want))
            mod.bind({name: ctypes.cast(getattr(dll, name), ctypes.c_void_p).value for name in self.protos})
        except Exception as e:                                   # stale build (header changed): say so, keep working
            import warnings
            warnings.warn("rg_hip: _rg_native not usable (%s: %s); using ctypes — rebuild with make -C %s"
                          % (type(e).__name__, e, CSRC_DIR))
            return None
        if any(not hasattr(mod, name) for name in self.protos):
            return None
        return mod

    def __getattr__(self, name):
        if name.startswith("rg_"):
            self.load()
            if self._native is not None:
                fn = getattr(self._native, name)                 # status check and error text inside the wrapper
                setattr(self, name, fn)
                return fn
            raw = getattr(self._dll, name)
            ret = self.protos[name][0]
            if ret != "int" or name in _PLAIN_INT:
                return raw
            dll = self._dll

            def checked(*args):
                st = raw(*args)
                if st != 0:
                    raise RuntimeError("%s failed (%d): %s" % (name, st, dll.rg_last_error().decode()))
                return st
            checked.__name__ = name
            setattr(self, name, checked)      # cache
            return checked
        raise AttributeError(name)


lib = _Lib()

FAMILIES = ["conv_fwd", "conv_dgrad", "conv_wgrad", "norm", "eltwise", "pool", "loss", "cm", "optim", "misc", "conv_f8"]
