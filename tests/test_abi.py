"""The C-ABI library loads (without a GPU) and exports exactly the symbols
include/safeopt_hip.h declares; the ctypes table covers every one of them."""
import os
import re

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(REPO, "include", "safeopt_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sgp_[A-Za-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built():
    from safeopt_amd.build import build
    return build()


def test_header_declares_the_path():
    syms = declared_symbols()
    for must in ["sgp_gp_set_data", "sgp_gp_predict", "sgp_kern_K", "sgp_grid_confidence",
                 "sgp_grid_maximizers", "sgp_grid_candidates", "sgp_grid_topk",
                 "sgp_grid_expander_check", "sgp_grid_lipschitz_check", "sgp_grid_argmax",
                 "sgp_swarm_fitness", "sgp_comm_init", "sgp_comm_allreduce_max"]:
        assert must in syms


def test_library_exports_every_declared_symbol(built):
    import ctypes
    lib = ctypes.CDLL(built)
    for name in declared_symbols():
        assert hasattr(lib, name), "libsafeopt_hip.so lacks " + name


def test_ctypes_table_matches_header(built):
    from safeopt_amd import _hip
    assert sorted(_hip.PROTOTYPES) == declared_symbols()
    _hip.lib()                               # binds restype/argtypes of all of them
    assert isinstance(_hip.device_count(), int)


def test_no_device_fails_loudly(built):
    from safeopt_amd import _hip
    if _hip.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(_hip.HipError):
        _hip.Context(0)
    import safeopt_amd.gpy as gpy
    import numpy as np
    with pytest.raises(_hip.HipError):      # no silent CPU fallback anywhere
        gpy.models.GPRegression(np.zeros((1, 1)), np.ones((1, 1)))
    with pytest.raises(_hip.HipError):
        gpy.kern.RBF(1).K(np.zeros((2, 1)))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "safeopt_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_inline_asm_mfma_hazards():
    """The sweep kernels issue v_mfma_f64_4x4x4_4b_f64 from inline asm (tied
    accumulators), which the compiler's hazard recogniser cannot pad around.  The
    scanner walks the ISA of every kernel instance for the data hazards that can
    then occur (VALU copy -> MFMA read, MFMA write -> spill store, ...): none may
    be present in the shipped build (needs hipcc, no GPU)."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                        "scripts", "dev", "check_mfma_hazards.py")
    spec = importlib.util.spec_from_file_location("check_mfma_hazards", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main([]) == 0
