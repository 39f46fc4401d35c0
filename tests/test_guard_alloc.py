"""The memory-safety gate of the GPU suite (tests/guard_alloc.cpp, tests/conftest.py: TTSMI_GUARD_ALLOC=1).
CPU: the library is built by __graft_entry__.build() and exports the allocator's entry points.  GPU: in a child process
under the guard allocator, (a) a plain computation is unchanged and trips nothing, (b) a write into the alignment slack
behind a tensor is counted as a canary violation.  (The fault on a read past the end is demonstrated by
tools/guard_selftest.py, outside the suite: it kills its process by design.)"""
import ctypes
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'tests', '_guard', 'libttsmi_guard_alloc.so')


def test_guard_allocator_is_built_and_exports_its_entry_points():
    from transformertts_amd import build as b
    b.build_guard_allocator(verbose=False)
    lib = ctypes.CDLL(LIB)
    for fn in ('ttsmi_guard_alloc', 'ttsmi_guard_free', 'ttsmi_guard_violations', 'ttsmi_guard_allocations',
               'ttsmi_guard_live_bytes'):
        assert hasattr(lib, fn), fn
    lib.ttsmi_guard_violations.restype = ctypes.c_long
    assert lib.ttsmi_guard_violations() == 0            # (no device call: counters only)


CHILD = r'''
import ctypes, gc, sys, torch
path = sys.argv[1]
torch.cuda.memory.change_current_allocator(torch.cuda.memory.CUDAPluggableAllocator(path, 'ttsmi_guard_alloc', 'ttsmi_guard_free'))
lib = ctypes.CDLL(path)
lib.ttsmi_guard_violations.restype = ctypes.c_long
a = torch.arange(1001, dtype=torch.float32, device='cuda')
b = (a * 2).sum().item()
assert b == 1001 * 1000, b
assert a.data_ptr() % 16 == 0
del a
gc.collect()
clean = lib.ttsmi_guard_violations()
x = torch.zeros(1001, dtype=torch.uint8, device='cuda')            # 7 bytes of slack behind it, still mapped
import ctypes as C
hip = C.CDLL('libamdhip64.so')
assert hip.hipMemset(C.c_void_p(x.data_ptr() + 1001), 3, C.c_size_t(3)) == 0    # three bytes past the end
torch.cuda.synchronize()
del x
gc.collect()
print('RESULT', clean, lib.ttsmi_guard_violations())
'''


@pytest.mark.gpu
def test_guard_allocator_catches_a_write_behind_a_tensor():
    r = subprocess.run([sys.executable, '-c', CHILD, LIB], capture_output=True, text=True, timeout=300,
                       env={k: v for k, v in os.environ.items() if k != 'TTSMI_GUARD_ALLOC'})
    assert r.returncode == 0, r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith('RESULT')][-1].split()
    assert (int(line[1]), int(line[2])) == (0, 1), (r.stdout, r.stderr[-1000:])
    assert 'CANARY VIOLATION' in r.stderr and 'tail slack' in r.stderr
