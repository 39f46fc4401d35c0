"""CPU-side checks of the drop-in boundary: libttsmi.so loads, exports every symbol include/ttsmi.h
declares, the ctypes table binds each with the declared parameter count, and argument validation
returns error codes (no compute launches - there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, 'include', 'ttsmi.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    out = {}
    for m in re.finditer(r'\b(?:int|size_t|const char\*)\s+(ttsmi_\w+)\s*\(([^;]*?)\)\s*;', src, flags=re.S):
        name, params = m.group(1), m.group(2).strip()
        n = 0 if params in ('', 'void') else len([p for p in params.split(',') if p.strip()])
        out[name] = n
    return out


@pytest.fixture(scope='module')
def built():
    import __graft_entry__ as g
    g.build()
    from transformertts_amd import _lib
    return _lib


def test_header_declares_expected_surface():
    d = _declared()
    for name in ('ttsmi_linear_fwd', 'ttsmi_attention_fwd', 'ttsmi_attention_bwd',
                 'ttsmi_add_layernorm_fwd', 'ttsmi_conv1d_fwd', 'ttsmi_lenreg_index',
                 'ttsmi_lenreg_fwd', 'ttsmi_l1_loss', 'ttsmi_adam_tf', 'ttsmi_stft_logmel'):
        assert name in d
    assert len(d) >= 35


def test_library_exports_and_binds_every_declared_symbol(built):
    d = _declared()
    l = built.lib()
    assert set(d) == set(built.SIGNATURES), set(d) ^ set(built.SIGNATURES)
    for name, nparams in d.items():
        assert hasattr(l, name), f'{name} not exported'
        assert len(built.SIGNATURES[name][1]) == nparams, name
    header_version = int(re.search(r'#define\s+TTSMI_VERSION\s+(\d+)', open(os.path.join(ROOT, 'include', 'ttsmi.h')).read()).group(1))
    assert l.ttsmi_version() == built.EXPECTED_VERSION == header_version    # include/ttsmi.h TTSMI_VERSION, checked at load time


def test_invalid_arguments_return_error_codes_not_crashes(built):
    l = built.lib()
    rc = l.ttsmi_linear_fwd(None, 0, None, 0, 0, None, 0, None, None, 0, 4, 4, 4, 0, 0, None)
    assert rc == -1
    assert b'null' in l.ttsmi_last_error()
    with pytest.raises(built.TtsmiError):
        built.check(rc, 'linear_fwd')
    # unsupported dtype / n_fft are reported, not ignored
    one = ctypes.c_void_p(16)
    rc = l.ttsmi_linear_fwd(one, 4, None, 0, 0, one, 4, None, one, 4, 4, 4, 4, 0, 7, None)
    assert rc == -1 and b'dtype' in l.ttsmi_last_error()
    rc = l.ttsmi_stft_logmel(one, one, one, 1, 1, 512, 256, one, 80, one, one, one, one, 0, 1e-5, one, None)
    assert rc == -3 and b'n_fft' in l.ttsmi_last_error()
    assert l.ttsmi_linear_wgrad_ws_bytes(28800, 1024, 256) > 0
    # the fused weighted-loss entry refuses term counts outside 1..8 and null tables before touching anything
    arr = (ctypes.c_void_p * 9)(*[16] * 9)
    for n_terms in (0, 9):
        rc = l.ttsmi_l1_losses_weighted(n_terms, arr, arr, arr, arr, arr, arr, arr, None, arr, arr, one, one, one, 1 << 20, None)
        assert rc == -1 and b'terms' in l.ttsmi_last_error()
    rc = l.ttsmi_l1_losses_weighted(3, None, arr, arr, arr, arr, arr, arr, None, arr, arr, one, one, one, 1 << 20, None)
    assert rc == -1 and b'null' in l.ttsmi_last_error()
    assert l.ttsmi_l1_losses_weighted_ws_bytes(3) >= 3 * 512 * 4
    assert l.ttsmi_griffinlim_ws_bytes(2) == 0 and l.ttsmi_griffinlim_ws_bytes(900) > 900 * 513 * 8


def test_missing_library_fails_loudly(built, monkeypatch):
    monkeypatch.setattr(built, '_lib', None)
    monkeypatch.setattr(built, 'LIB_PATH', '/nonexistent/libttsmi.so')
    with pytest.raises(built.TtsmiError):
        built.lib()


def test_dense_block_descriptor_layout_matches_the_header(tmp_path):
    """_lib.DenseBlockDesc (ctypes) must mirror `ttsmi_dense_block` of include/ttsmi.h field for field: compile the
    header with the host compiler and compare the size and every field offset."""
    import ctypes
    import subprocess
    from transformertts_amd._lib import DenseBlockDesc
    names = [n for n, _ in DenseBlockDesc._fields_]
    src = tmp_path / 'layout.c'
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(ROOT, "include", "ttsmi.h")}"',
             'int main(void) {', '  printf("%zu\\n", sizeof(ttsmi_dense_block));']
    lines += [f'  printf("%zu\\n", offsetof(ttsmi_dense_block, {n}));' for n in names]
    lines += ['  return 0;', '}']
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.run(['gcc', str(src), '-o', str(exe)], check=True)
    out = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert out[0] == ctypes.sizeof(DenseBlockDesc)
    assert out[1:] == [getattr(DenseBlockDesc, n).offset for n in names]


def test_train_step_descriptor_layout_matches_the_header(tmp_path):
    """_lib.FtStep / FtPredictor / FtPredLayer (ctypes) mirror `ttsmi_ft_step` and its members field for field."""
    import ctypes
    import subprocess
    from transformertts_amd import _lib
    structs = (('ttsmi_ft_pred_layer', _lib.FtPredLayer), ('ttsmi_ft_predictor', _lib.FtPredictor), ('ttsmi_ft_step', _lib.FtStep))
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(ROOT, "include", "ttsmi.h")}"', 'int main(void) {']
    want = []
    for cname, cls in structs:
        lines.append(f'  printf("%zu\\n", sizeof({cname}));')
        want.append(ctypes.sizeof(cls))
        for n, _ in cls._fields_:
            lines.append(f'  printf("%zu\\n", offsetof({cname}, {n}));')
            want.append(getattr(cls, n).offset)
    lines += ['  return 0;', '}']
    src = tmp_path / 'layout_ft.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout_ft'
    subprocess.run(['gcc', str(src), '-o', str(exe)], check=True)
    out = [int(x) for x in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert out == want
    assert _lib.FT_MAX_PRED_LAYERS == 8 and _lib.FT_MAX_BLOCKS == 32      # TTSMI_FT_MAX_* of the header
    hdr = open(os.path.join(ROOT, 'include', 'ttsmi.h')).read()
    assert '#define TTSMI_FT_MAX_PRED_LAYERS 8' in hdr and '#define TTSMI_FT_MAX_BLOCKS 32' in hdr


def test_a_stale_library_is_refused_and_the_override_needs_an_opt_in(tmp_path):
    """A build with another ABI version must not be called (its argument lists differ): _lib.lib() raises.  TTSMI_LIB is a
    measurement knob: ignored unless TTSMI_ALLOW_LIB_OVERRIDE=1 is set with it (advisor finding, round 3)."""
    import subprocess
    import sys
    src = tmp_path / 'stale.c'
    src.write_text('int ttsmi_version(void) { return 100; }\n')
    so = tmp_path / 'libttsmi_stale.so'
    subprocess.run(['gcc', '-shared', '-fPIC', str(src), '-o', str(so)], check=True)
    code = ('import sys; sys.path.insert(0, %r)\n'
            'from transformertts_amd import _lib\n'
            'print(_lib.LIB_PATH)\n'
            'try:\n'
            '    _lib.lib(); print("LOADED")\n'
            'except _lib.TtsmiError as e:\n'
            '    print("REFUSED", "ABI version 100" in str(e))\n') % ROOT
    env = dict(os.environ, TTSMI_LIB=str(so))
    env.pop('TTSMI_ALLOW_LIB_OVERRIDE', None)
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, check=True).stdout.split('\n')
    assert out[0].endswith(os.path.join('transformertts_amd', 'lib', 'libttsmi.so')) and out[1] == 'LOADED', out
    env['TTSMI_ALLOW_LIB_OVERRIDE'] = '1'
    out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True, check=True).stdout.split('\n')
    assert out[0] == str(so) and out[1] == 'REFUSED True', out


def test_every_included_header_is_a_build_dependency_and_part_of_the_digest():
    """A header that a source includes but the build does not list neither triggers a rebuild nor moves the library digest
    (round 5: csrc/chain16b.h was edited, the library stayed the old one and a GPU session measured it).  Every quoted
    #include of csrc/ must resolve to a file in build._headers(), and every source must be in build.SOURCES."""
    import re
    from transformertts_amd import build
    csrc = build.CSRC
    listed = {os.path.realpath(h) for h in build._headers()}
    files = [f for f in os.listdir(csrc) if f.endswith(('.hip', '.cpp', '.h'))]
    assert {f for f in files if f.endswith(('.hip', '.cpp'))} == set(build.SOURCES)
    for f in files:
        for inc in re.findall(r'^\s*#\s*include\s+"([^"]+)"', open(os.path.join(csrc, f)).read(), re.M):
            cands = [os.path.join(csrc, inc), os.path.join(csrc, '..', '..', 'include', inc)]
            hit = [os.path.realpath(c) for c in cands if os.path.exists(c)]
            assert hit, f'{f}: #include "{inc}" not found'
            assert hit[0] in listed, f'{f} includes {inc}, which is not a build dependency'
    # the digest moves with a header: same inputs plus one byte
    base = build.library_digest()
    extra = build._digest([os.path.join(csrc, s) for s in build.SOURCES] + build._headers() + [__file__])[:16]
    assert base != extra and len(base) == 16
