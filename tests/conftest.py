import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle_mod():
    """The CPU oracle (test infrastructure): built on demand from oracle/ch_oracle.c."""
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def golden():
    import json
    here = os.path.join(REPO, "tests", "golden")
    with open(os.path.join(here, "sql_reference_rows.json")) as f:
        rows = json.load(f)
    with open(os.path.join(here, "hash_kat.json")) as f:
        kat = json.load(f)
    return dict(rows=rows, kat=kat["kat"])
