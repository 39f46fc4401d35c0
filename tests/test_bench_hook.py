"""bench.py's instrumented step must know every C-ABI entry point the product calls WITHOUT a stream as its last
argument: the hook treats a trailing integer as the launch stream (round 4: the new stack launchers ended in a device
pointer and the default bench line crashed in its roofline leg - caught by the profile session, not by a test)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_every_streamless_entry_point_is_passed_through_by_the_hook():
    src = open(os.path.join(ROOT, 'include', 'ttsmi.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    bench = open(os.path.join(ROOT, 'bench.py')).read()
    hook = bench[bench.index('def hook(name, args, fn):'):bench.index('# the launch stream is the entry point')]
    for m in re.finditer(r'\b(?:int|size_t|const char\*)\s+(ttsmi_\w+)\s*\(([^;]*?)\)\s*;', src, flags=re.S):
        name, params = m.group(1), m.group(2)
        if 'ttsmi_stream_t' in params.split(',')[-1]:
            continue                                      # launches on an explicit stream: bracketed with events
        passed = (name.endswith('_bytes') or name.endswith('_nparts') or name.endswith('_supported') or 'comm' in name
                  or f"'{name}'" in hook)
        assert passed, f'{name} has no stream argument and bench.py:instrumented_step does not pass it through'
