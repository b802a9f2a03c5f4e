"""Build libfv_hotpath.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m face_vijnana_yolov3_amd.build [--force]

hipcc cross-compiles without a GPU.  The built .so is git-ignored but travels to the GPU box.
"""
import concurrent.futures
import glob
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, 'csrc')
OBJ = os.path.join(CSRC, '_obj')
LIB = os.path.join(PKG, 'libfv_hotpath.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
FLAGS = ['--offload-arch=gfx950', '-O3', '-fPIC', '-std=c++17', '-ffp-contract=off', '-fno-slp-vectorize', '-Wall',
         '-Wno-unused-function', '-Wno-unused-variable']


def source_fingerprint():
    """sha1 over the kernel sources: profiles/*_pmc_traffic.json records it, and bench.py reports the
    measured HBM traffic only while the kernels are still the ones that were profiled."""
    import hashlib
    h = hashlib.sha1()
    for f in sorted(glob.glob(os.path.join(CSRC, '*.hip')) + glob.glob(os.path.join(CSRC, '*.h'))):
        h.update(os.path.basename(f).encode()); h.update(open(f, 'rb').read())
    return h.hexdigest()[:16]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, '*.hip')))
    hdrs = sorted(glob.glob(os.path.join(CSRC, '*.h'))) + sorted(glob.glob(os.path.join(PKG, '..', 'include', '*.h')))
    jobs = []
    objs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + '.o')
        objs.append(o)
        if force or _stale(o, [s] + hdrs):
            jobs.append([HIPCC] + FLAGS + ['-c', s, '-o', o])

    def run(cmd):
        if verbose:
            print(' '.join(cmd), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError('hipcc failed:\n%s\n%s' % (' '.join(cmd), r.stderr[-8000:]))
        return r.stderr

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        for warn in ex.map(run, jobs):
            if verbose and warn.strip():
                print(warn)
    if force or jobs or _stale(LIB, objs):
        run([HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs)
    return LIB


if __name__ == '__main__':
    print(build_library(force='--force' in sys.argv, verbose=True))
