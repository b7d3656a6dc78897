import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


def pytest_collection_modifyitems(config, items):
    """`pytest tests` on a machine without a gfx950 device: GPU tests are reported as skipped, not failed
    (there is no CPU fallback to run them on).  PCLEAN_REQUIRE_GPU=1 keeps them loud."""
    if os.environ.get("PCLEAN_REQUIRE_GPU") == "1":
        return
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no gfx950 device (pclean_amd has no CPU fallback)")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (test infrastructure) — built on demand with g++."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle as orc
    orc.build()
    return orc


@pytest.fixture(scope="session")
def hip():
    """One HIP context on cuda:0; fails loudly when the extension or GPU is missing."""
    from pclean_amd import HipContext, PCleanHipError
    try:
        ctx = HipContext(0)
    except PCleanHipError:
        # CPU-only CI: a missing device is not a regression.  On a GPU box (PCLEAN_REQUIRE_GPU=1, or torch sees a
        # device) the error stays loud.
        import torch
        if os.environ.get("PCLEAN_REQUIRE_GPU") == "1" or torch.cuda.is_available():
            raise
        pytest.skip("no gfx950 device")
    yield ctx
    ctx.close()


@pytest.fixture(scope="session")
def hospital_columns():
    """Unique dirty values per queried hospital column (load_data.jl:17-18)."""
    import pandas as pd
    d = pd.read_csv(os.path.join(ROOT, "datasets", "hospital_dirty.csv"), dtype=str, keep_default_na=False)
    cols = ["ProviderNumber", "HospitalName", "Address1", "City", "State", "ZipCode", "CountyName", "PhoneNumber",
            "HospitalType", "HospitalOwner", "EmergencyService", "Condition", "MeasureCode", "MeasureName", "Stateavg"]
    return {c: list(dict.fromkeys(d[c].tolist())) for c in cols}
