import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """-m gpu tests are skipped (not failed) when no GPU is visible, so a plain `pytest tests`
    on the CPU container stays green; on the GPU box they run for real."""
    try:
        import torch
        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        return
    skip = pytest.mark.skip(reason='no GPU visible')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


def pytest_sessionfinish(session, exitstatus):
    """Leave the device idle and the garbage collected before the interpreter shuts down: captured graphs, side streams
    and events of the last tests are then destroyed while the HIP runtime is still whole, not during its teardown."""
    try:
        import gc

        import torch
        if torch.cuda.is_available() and torch.cuda.is_initialized():
            torch.cuda.synchronize()
            gc.collect()
            torch.cuda.synchronize()
    except Exception:  # pragma: no cover
        pass
