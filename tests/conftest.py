import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The package refuses to import without its HIP library (no fallback path).  Build it when it is missing or older
    # than its sources (hipcc cross-compiles gfx950 without a GPU); a prebuilt, current .so is left alone.
    import __graft_entry__ as ge
    ge._build_native_module().build(force=False, verbose=False)


def pytest_collection_modifyitems(config, items):
    import torch
    # device_count() does not initialise the GPU (is_available() does): tests/test_00_ranks_on_one_card.py starts its
    # worker processes before this process has touched the card
    if torch.cuda.device_count() > 0:
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
