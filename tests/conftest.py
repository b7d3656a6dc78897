import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box via gpurun)")


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
    from pclean_amd import HipContext
    ctx = HipContext(0)
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
