import sys
from pathlib import Path

import pytest
import torch  # noqa: F401  (before anything loads libhip_raytracer.so: torch ships its own ROCm runtime, and initialising it after the system one has been loaded into the process fails with "No HIP GPUs are available")

ROOT = Path(__file__).resolve().parent.parent
for p in (str(ROOT), str(ROOT / "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


@pytest.fixture(scope="session")
def restatement():
    """fused/unfused CPU restatements (the oracle). Built on demand."""
    from oracle import oracle
    return {True: oracle.Restatement(True), False: oracle.Restatement(False)}
