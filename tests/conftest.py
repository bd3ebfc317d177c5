import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _native_diagnostics(request):
    """GPU runs only: a SIGABRT (HIP/HSA abort on a GPU memory fault, an escaped C++ exception) first prints the NATIVE
    stack of the aborting thread to fd 2.  Together with ``--capture=sys`` in pytest.ini (fd 2 stays the real stderr: pytest's
    default fd-level capture swallows what the HIP runtime prints before it aborts, e.g. "Memory access fault by GPU node",
    because the process dies before the captured text is ever reported -- round 1 lost exactly that text twice,
    profiles/README.md) an abort now leaves its cause in the log."""
    import torch
    if torch.cuda.is_available():
        try:
            import nsgp_repre_amd
            nsgp_repre_amd.load_library().nsgp_debug_install_abort_backtrace()
        except Exception as exc:    # diagnostics must never fail a run
            print(f"[conftest] abort backtrace not installed: {exc}")
    yield


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
