import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# ---------------------------------------------------------------------------------------------------------------------
# TTSMI_GUARD_ALLOC=1: the memory-safety gate (tests/guard_alloc.cpp).  Every device tensor of this process comes from its
# own mapping, flush against an unmapped page (out-of-bounds accesses past a tensor's end fault), with canaries in front
# and in the alignment slack behind (checked at free; a violation fails the test that was running).  Must be installed
# before the first device allocation, hence at conftest import.  hipGraph capture needs the caching allocator's pools:
# tests that capture are skipped in this mode.
# ---------------------------------------------------------------------------------------------------------------------
GUARD = os.environ.get('TTSMI_GUARD_ALLOC', '0') == '1'
_GUARD_LIB = None
GUARD_SKIP = ('graph', 'hipgraph', 'two_ranks', 'two_rank', 'dp_overlap', 'positional_table')      # captures, or child processes / collectives


def _install_guard():
    global _GUARD_LIB
    import ctypes

    import torch
    path = os.path.join(ROOT, 'tests', '_guard', 'libttsmi_guard_alloc.so')
    if not os.path.exists(path):
        raise RuntimeError(f'{path} missing: run `python -c "import __graft_entry__ as g; g.build()"` first')
    alloc = torch.cuda.memory.CUDAPluggableAllocator(path, 'ttsmi_guard_alloc', 'ttsmi_guard_free')
    torch.cuda.memory.change_current_allocator(alloc)
    _GUARD_LIB = ctypes.CDLL(path)
    for fn in ('ttsmi_guard_violations', 'ttsmi_guard_allocations', 'ttsmi_guard_live_bytes'):
        getattr(_GUARD_LIB, fn).restype = ctypes.c_long


if GUARD:
    _install_guard()


@pytest.fixture(autouse=True)
def _guard_canaries(request):
    """Guard mode: a canary violation recorded while this test ran (its tensors are collected first) fails it."""
    if not GUARD or 'gpu' not in request.keywords:
        yield
        return
    before = _GUARD_LIB.ttsmi_guard_violations()
    yield
    import gc

    import torch
    torch.cuda.synchronize()
    gc.collect()
    after = _GUARD_LIB.ttsmi_guard_violations()
    assert after == before, f'{after - before} canary violation(s): a kernel wrote outside a tensor (see stderr / TTSMI_GUARD_LOG)'


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
        if GUARD:
            skip = pytest.mark.skip(reason='TTSMI_GUARD_ALLOC=1: hipGraph capture / child processes are outside the guard allocator')
            for item in items:
                if any(k in item.name.lower() for k in GUARD_SKIP):
                    item.add_marker(skip)
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
