import os
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """CPU oracle (test infrastructure).  Built on demand with gcc."""
    import oracle as O

    O.build()
    return O


@pytest.fixture(scope="session")
def golden():
    import oracle as O

    return {
        "pedigree": O.load_pedigree(GOLDEN / "pedigree.txt"),                # data/pedigree.txt (351 rows)
        "divergence": np.loadtxt(GOLDEN / "divergence.txt"),                 # data/divergence.txt
        "generated": O.load_pedigree(GOLDEN / "pedigree_generated.txt"),     # data/pedigree_generated.txt
        "sparse": O.load_pedigree(GOLDEN / "pedigree_sparse.txt"),           # data/pedigree_sparse.txt
        "r_p0uu": float(open(GOLDEN / "r_p0uu.txt").read().strip()),
        "p0uu_generated": 0.6554051647850447,
    }


@pytest.fixture(scope="session")
def abn():
    """The product binding.  Importing never falls back to a CPU path."""
    import alphabeta_rs_amd as A

    A.load_library(build_if_missing=True)
    return A


@pytest.fixture(scope="session")
def gpu_ctx(abn):
    if abn.device_count() <= 0:
        pytest.fail("no HIP device visible: -m gpu tests need an MI355X (there is no CPU fallback)")
    ctx = abn.Context(0)
    yield ctx
    ctx.close()


MODEL_DEFAULT = np.array([0.0001179555, 0.0001180614, 0.03693534, 0.003023981])  # src/structs.rs:66-75
COST_KNOWN_ANSWER = 0.0006700888539608879                                          # src/structs.rs:233
