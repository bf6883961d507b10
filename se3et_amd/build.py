"""Builds se3et_amd/csrc/*.hip into the in-tree shared library libse3et_hip.so for gfx950 (hipcc cross-compiles
without a GPU).  `python -m se3et_amd.build [--force]`."""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'csrc')
LIB = os.path.join(CSRC, 'libse3et_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wall', '-Wno-unused-function']


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def extra_flags(src):
    """Per-source flags, selected by markers in the source text."""
    text = open(src).read()
    # the bit-exact geometry kernels must not fuse a*b+c (they mirror unfused x86 float arithmetic)
    extra = ['-ffp-contract=off'] if 'SE3_EXACT_FP' in text else []
    # (csrc/geo_records.hip: no packed-f32 arithmetic from the SLP vectoriser)
    extra += ['-fno-slp-vectorize'] if 'SE3_NO_SLP_VECTORIZE' in text else []
    return extra


def build(force=False, verbose=True):
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    hdrs = sorted(glob.glob(os.path.join(CSRC, '*.h'))) + [
        os.path.join(os.path.dirname(os.path.dirname(CSRC)), 'include', 'se3et_hip.h')]
    objs, jobs = [], []
    os.makedirs(os.path.join(CSRC, 'build'), exist_ok=True)
    # objects whose source is gone (renamed / deleted kernels) are removed, never linked
    names = {os.path.basename(s)[:-4] for s in srcs}
    for o in glob.glob(os.path.join(CSRC, 'build', '*.o')):
        if os.path.basename(o)[:-2] not in names:
            os.remove(o)
    for s in srcs:
        o = os.path.join(CSRC, 'build', os.path.basename(s)[:-4] + '.o')
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([HIPCC] + FLAGS + extra_flags(s) + ['-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
