import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")

# The library's default BGK accumulate mode is the order-free one (double accumulators, |dp| <= ~4e-7 from the reference's
# fp32 summation order).  The parity suites below demand BIT identity with the CPU restatement, which is what the ordered
# mode delivers: they run with it unless a test selects a mode itself (tests/test_bgk_sum_gpu.py covers the default mode).
os.environ.setdefault("LA3DM_BGK_SUM", "0")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def built():
    """make sure the product libraries and the oracle exist (compiles on CPU)."""
    import __graft_entry__ as g
    if not (os.path.exists(os.path.join(ROOT, "la3dm_amd", "csrc", "libla3dm_hip.so"))
            and os.path.exists(os.path.join(ROOT, "la3dm_amd", "csrc", "libla3dm_map.so"))
            and os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so"))):
        g.build()
    return True


def pcd_path(dataset, i):
    return os.path.join(GOLDEN, "data", dataset, f"{dataset}_{i}.pcd")
