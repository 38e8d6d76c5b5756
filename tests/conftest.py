import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
for p in (ROOT, ROOT / "petit-kernel_amd"):
    if str(p) not in sys.path:
        sys.path.insert(0, str(p))

GOLDEN = ROOT / "tests" / "golden"


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    # Safety net for a checkout without built artefacts (they are git-ignored): build the product library and the
    # oracle once, exactly as __graft_entry__.build() does.  Nothing is substituted: if hipcc fails, the suite fails.
    lib = ROOT / "petit-kernel_amd" / "lib" / "libpetit_amd.so"
    ora = ROOT / "oracle" / "_build" / "libpetit_oracle.so"
    if not lib.exists() or not ora.exists():
        import __graft_entry__
        __graft_entry__.build()


def pytest_collection_modifyitems(config, items):
    # `-m gpu` selects them explicitly; without a GPU they are skipped, never faked.
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason="no GPU in this container (gpu-marked tests run on the MI355X box)")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN
