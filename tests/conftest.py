import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "reid-gan_amd")
for p in (PKG, REPO):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: a second / third parametrisation of a >= 5 s step test whose path another test of the "
                                       "default run already compares with the oracle; run with -m 'gpu and slow' or RG_RUN_SLOW=1 "
                                       "(tools/gpu_tests_all.sh)")


def pytest_collection_modifyitems(config, items):
    """no test may hang the suite: a stuck multi-process rendezvous (or a GPU run-away) fails after a bound instead
    (pytest-timeout is part of the image; without it the markers are inert)"""
    # the default `-m gpu` run keeps one representative of every step test; the duplicates sit behind the `slow` marker
    expr = config.getoption("-m", default="") or ""
    if os.environ.get("RG_RUN_SLOW") != "1" and "slow" not in expr:
        kept, dropped = [], []
        for item in items:
            (dropped if item.get_closest_marker("slow") is not None else kept).append(item)
        if dropped:
            config.hook.pytest_deselected(items=dropped)
            items[:] = kept
    if not config.pluginmanager.hasplugin("timeout"):
        return
    for item in items:
        if item.get_closest_marker("timeout") is None:
            item.add_marker(pytest.mark.timeout(900 if item.get_closest_marker("gpu") else 300))


@pytest.fixture(scope="session")
def dev():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The HIP library is built in-tree by __graft_entry__.build() / make; a fresh checkout that runs the tests first
    builds it here (hipcc cross-compiles gfx950 without a GPU) instead of failing every import of rg_hip."""
    lib = os.path.join(PKG, "lib", "libreidgan_hip.so")
    if not os.path.exists(lib) and os.path.exists("/opt/rocm/bin/hipcc"):
        import subprocess
        subprocess.run(["make", "-C", os.path.join(PKG, "csrc")], check=True, stdout=subprocess.DEVNULL)
    yield
