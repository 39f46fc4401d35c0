#!/usr/bin/env python
"""Which CUs does a hipExtStreamCreateWithCUMask bit select on this part?  Launches a census (one-wave workgroups that
report HW_REG_XCC_ID) on streams masked two ways - bits i with i % 8 == x (ops._new_wgrad_stream's assumption: bit i = CU
i / 8 of XCC i % 8) and a contiguous run of bits - and prints the per-XCC counts."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from transformertts_amd import _lib  # noqa: E402
from transformertts_amd._lib import check  # noqa: E402


def census(bits, label):
    l = _lib.lib()
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    words = (ctypes.c_uint32 * ((ncu + 31) // 32))()
    for i in bits:
        words[i // 32] |= 1 << (i % 32)
    h = ctypes.c_void_p()
    check(l.ttsmi_debug_stream_create_cu_mask(ctypes.addressof(words), len(words), ctypes.addressof(h)), 'mask')
    counts = torch.zeros(8, dtype=torch.int32, device='cuda:0')
    torch.cuda.synchronize()
    check(l.ttsmi_debug_xcc_census(counts.data_ptr(), 4096, h.value), 'census')
    torch.cuda.synchronize()
    print(f'{label:32s} bits {len(list(bits)):3d}  per-XCC workgroups {counts.cpu().tolist()}')


if __name__ == '__main__':
    ncu = torch.cuda.get_device_properties(0).multi_processor_count
    print('CUs', ncu)
    census(range(ncu), 'all bits')
    for x in (0, 3):
        census([i for i in range(ncu) if i % 8 == x], f'bits i % 8 == {x}')
    census([i for i in range(ncu) if i % 8 < 3], 'bits i % 8 < 3')
    census(range(0, 32), 'bits 0..31')
    census(range(32, 64), 'bits 32..63')
