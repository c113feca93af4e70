import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the CPU oracle's thread pool: a 1-GPU box grants 16 host cores but reports the whole machine's count, and an oversubscribed
    # pool runs the oracle ~3x slower (50 DDIM steps at 512 px: 596 s against ~220 s); same rule as bench.py's cpu_baseline leg
    try:
        import torch
        torch.set_num_threads(int(os.environ.get("AGD_CPU_THREADS", min(os.cpu_count() or 1, 16))))
    except Exception:
        pass


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped (not failed) when no device is visible, e.g. a bare `pytest tests/`
    in the build container; on the GPU box the HIP library must load or the tests FAIL."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    """measured parity figures of this run -> gpurun_out/parity_report.json (tests/_report.py)"""
    try:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _report
        _report.dump()
    except Exception:
        pass


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
