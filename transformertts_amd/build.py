"""Build libttsmi.so (hipcc, gfx950 only) in-tree: transformertts_amd/lib/libttsmi.so.

hipcc cross-compiles without a GPU, so this runs in the CPU build container; the resulting .so is
git-ignored but travels to the GPU box with the repo snapshot."""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIBDIR = os.path.join(HERE, 'lib')
OBJDIR = os.path.join(HERE, 'build')
LIB = os.path.join(LIBDIR, 'libttsmi.so')
SOURCES = ['api.cpp', 'gemm.hip', 'gemm_bf16.hip', 'attention.hip', 'attention_bf16.hip', 'layernorm.hip',
           'elementwise.hip', 'lenreg.hip', 'stft_mel.hip', 'dense_block.hip', 'rowgemm.hip', 'gemm_k256.hip', 'chain.hip', 'train_step.hip', 'griffinlim.hip', 'nnls.hip', 'collective.cpp']
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-result']
# measurement builds only, e.g. TTSMI_EXTRA_HIPCC_FLAGS=-DTTSMI_ABLATION_BUILD (stage-ablation knobs, csrc/common.h)
FLAGS += os.environ.get('TTSMI_EXTRA_HIPCC_FLAGS', '').split()


def _hipcc() -> str:
    for cand in (os.environ.get('HIPCC'), '/opt/rocm/bin/hipcc', 'hipcc'):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    raise RuntimeError('hipcc not found')


def _digest(paths) -> str:
    h = hashlib.sha256()
    for p in paths:
        with open(p, 'rb') as f:
            h.update(f.read())
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


def _headers():
    """Every header a source may include: all of csrc/*.h (a header missing from a hand-kept list once left a kernel edit
    out of the build and out of the digest) and the public include/ttsmi.h."""
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')) + [os.path.join(HERE, '..', 'include', 'ttsmi.h')]


def library_digest() -> str:
    """One digest of everything libttsmi.so is compiled from (every source, the shared headers, the flags): measurement
    files that describe the kernels of a particular build (profiles/*_pmc_hbm_traffic_*.json) are stamped with it, and
    bench.py reports `traffic: null` when the stamp is not the digest of the tree it runs from."""
    headers = _headers()
    return _digest([os.path.join(CSRC, s) for s in SOURCES] + headers)[:16]


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    headers = _headers()
    hipcc = _hipcc()
    jobs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(OBJDIR, src + '.o')
        stamp = obj + '.sha'
        dig = _digest([sp] + headers)
        if not force and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
            continue
        jobs.append((sp, obj, stamp, dig))

    def compile_one(job):
        sp, obj, stamp, dig = job
        cmd = [hipcc] + FLAGS + ['-x', 'hip', '-c', sp, '-o', obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'hipcc failed for {sp}:\n{r.stdout}\n{r.stderr}')
        with open(stamp, 'w') as f:
            f.write(dig)
        return sp

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            for done in ex.map(compile_one, jobs):
                if verbose:
                    print(f'[ttsmi build] compiled {os.path.basename(done)}', file=sys.stderr)
    objs = [os.path.join(OBJDIR, s + '.o') for s in SOURCES]
    if jobs or force or not os.path.exists(LIB):
        cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs + ['-ldl']
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f'link failed:\n{r.stdout}\n{r.stderr}')
        if verbose:
            print(f'[ttsmi build] linked {LIB}', file=sys.stderr)
    return LIB


GUARD_SRC = os.path.join(HERE, '..', 'tests', 'guard_alloc.cpp')
GUARD_LIB = os.path.join(HERE, '..', 'tests', '_guard', 'libttsmi_guard_alloc.so')


def build_guard_allocator(force: bool = False, verbose: bool = True) -> str:
    """tests/guard_alloc.cpp -> tests/_guard/libttsmi_guard_alloc.so: the memory-safety gate of the GPU suite (a torch
    pluggable allocator; TEST infrastructure, host code only - the product path never loads it)."""
    os.makedirs(os.path.dirname(GUARD_LIB), exist_ok=True)
    stamp = GUARD_LIB + '.sha'
    dig = _digest([GUARD_SRC])
    if not force and os.path.exists(GUARD_LIB) and os.path.exists(stamp) and open(stamp).read() == dig:
        return GUARD_LIB
    cmd = [_hipcc(), '-O2', '-std=c++17', '-fPIC', '-shared', '-x', 'hip', '--offload-arch=gfx950', GUARD_SRC, '-o', GUARD_LIB]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f'hipcc failed for {GUARD_SRC}:\n{r.stdout}\n{r.stderr}')
    with open(stamp, 'w') as f:
        f.write(dig)
    if verbose:
        print(f'[ttsmi build] built {GUARD_LIB}', file=sys.stderr)
    return GUARD_LIB


if __name__ == '__main__':
    print(build(force='--force' in sys.argv))
    print(build_guard_allocator(force='--force' in sys.argv))
