import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")
    # the tests check the PRODUCT build: a GMR1_HIP_LIBRARY left in the environment by a profiling script (it points at
    # libgmr1_hip_prof.so, whose debug switches change results) must not redirect them
    os.environ.pop("GMR1_HIP_LIBRARY", None)
    # A fresh checkout has no built artefacts (they are git-ignored): compile the HIP library (hipcc cross-compiles
    # without a GPU) and the CPU oracle once, before collection -- what __graft_entry__.build() does.  The product API
    # itself never builds or falls back: without the .so it raises.
    from __graft_entry__ import load_package
    pkg = load_package()
    if not os.path.exists(pkg.build.LIB):
        pkg.build.build()
    import oracle_lib
    oracle_lib.build()


@pytest.fixture(scope="session")
def pkg():
    from __graft_entry__ import load_package
    return load_package()


@pytest.fixture(scope="session")
def orc():
    import oracle_lib
    oracle_lib.lib()
    return oracle_lib


@pytest.fixture(scope="session")
def gpu_api(pkg):
    """The product API, initialised on device 0 (GPU tests only)."""
    # torch carries its own ROCm runtime: when a test also uses torch for HBM tensors / streams,
    # torch has to bring the runtime up first (the order bench.py uses), then the library binds to it
    import torch
    torch.cuda.init()
    pkg.api.load()
    pkg.api.init(0)
    return pkg.api


@pytest.fixture(params=["generic", "acc"])
def decoder(request, gpu_api, orc):
    """Runs a GPU parity test once per libosmocore Viterbi decoder (decisions D1 / D1b): the product in
    the mode gmr1_hip_set_conv_decoder selects against the oracle restating the same decoder."""
    acc = request.param == "acc"
    with gpu_api.conv_decoder(gpu_api.CONV_ACC if acc else gpu_api.CONV_GENERIC), orc.conv_mode(1 if acc else 0):
        yield request.param
